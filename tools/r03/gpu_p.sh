#!/bin/bash
# round 3, call P: taper parameters of the work items (big factor, small divisor, fraction of the samples in big items): C2 full frame + 1/8 shard
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03p
for l in base tb35 ts4 tf90 tf75 tb2s2 tb4s5; do
  echo "== $l"; if [ $l = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$PWD/luisarender_amd/lib/variants/liblrhip_$l.so; fi
  timeout 300 python tools/shard_probe.py 1024 quick 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03p/taper_sweep.txt

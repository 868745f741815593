#!/bin/bash
# round 6, call 4: the stall probe, level 1 (the shipped flow with four marks) and level 2 (the split flow, every section)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
for cfg in "c2 256" "c4 64" "c5 512" "c3 256"; do set -- $cfg
  LIB=probe timeout 300 python tools/stall_probe.py $1 $2 $O/stalls1_$1.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls1_$1.txt
done
SAMPLER=PaddedSobol LIB=probe timeout 300 python tools/stall_probe.py c2 256 $O/stalls1_c2_sobol.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls1_c2_sobol.txt
for cfg in "c2 256" "c4 64" "c5 512"; do set -- $cfg
  LIB=probe2 timeout 300 python tools/stall_probe.py $1 $2 $O/stalls2_$1.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls2_$1.txt
done

#!/bin/bash
# round 5, GPU call 2: what bounds the traversal loop now?  VALU sensitivity probes (extra dependent v_fma per node / leaf step), three waves per SIMD, fused fetch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05b; mkdir -p $O
SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 base fused pn32 pn64 pl32 w3 lt fl 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_pool.txt
SCHED=legacy timeout 900 python tools/ab_libs.py 256 c2 base fused pn32 pn64 w3 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_legacy.txt
SCHED=both timeout 300 python tools/ab_libs.py 64 c1 base fused w3 2>&1 | grep -v amdgpu.ids | tee $O/ab_c1.txt

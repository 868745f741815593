#!/bin/bash
# round 3, call D: wavefront mode (lean megakernel + heavy-closure kernel + continuation pass): GPU suite, C5 A/B against the
# all-in-one <124>, slice-size sweep; v_bfi plane select A/B on C2; PaddedSobol with closed-form dimensions 0 / 1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests -m gpu -q -s -x > gpurun_out/r03d/gpu_tests.log 2>&1; grep -E "passed|failed|C1 full|C2-class|wavefront vs|FAILED|Error|^E  " gpurun_out/r03d/gpu_tests.log | tail -14
{
echo "== C5 64 spp: wavefront (default slice)"; timeout 300 python tools/c5_ablation.py 64 full
echo "== C5 64 spp: all-in-one <124>"; WAVEFRONT=0 timeout 300 python tools/c5_ablation.py 64 full
for sp in 4194304 16777216 67108864; do echo "== C5 256 spp, slice paths $sp"; WF_SLICE_PATHS=$sp timeout 300 python tools/c5_ablation.py 256 full; done
echo "== C5 256 spp all-in-one"; WAVEFRONT=0 timeout 300 python tools/c5_ablation.py 256 full
echo "== C2 A/B v_bfi"; for i in 1 2; do tools/ab.sh 256 base bfi0; done
} > gpurun_out/r03d/ab.txt 2>&1
cat gpurun_out/r03d/ab.txt | grep -v amdgpu.ids
python bench.py --workload c2 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 independent 256', round(d['value'],1))"
python - <<'PY' 2>/dev/null
import sys, tempfile
sys.path.insert(0, '.')
import bench
class A: pass
with tempfile.TemporaryDirectory() as tmp:
    v = bench.run_workload("c2", A(), 0, 1, 0, tmp, 3, 1, 256, "PaddedSobol")
    print("c2 PaddedSobol 256 spp:", round(v[0], 1), "Msamples/s, kernel", v[3])
PY

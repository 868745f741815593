cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak > gpurun_out/r02_valu_peak.json
cat gpurun_out/r02_valu_peak.json | python -c "
import sys, json
d = json.load(sys.stdin)
for r in d['results']: print(r['op'], r['waves_per_simd'], round(r['cycles_per_wave_instr'], 3))
print(d['clock_mhz'], d['cus'])"
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cornell or tile_shards or determinism or edge" 2>&1 | tail -3
bash tools/ab.sh 256 base noxcd w3 w5 base noxcd 2>&1 | tee gpurun_out/r02b_ab.txt

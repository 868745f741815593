#!/bin/bash
# byte texels behind the 192 MB rule: the new parity test, then C4 / C5 under storage modes 0 (float), 1 (automatic), 2 (8-bit wherever possible)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zd; O=gpurun_out/r05zd
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "byte_texels or jpeg_texture or textured_disney" 2>&1 | tail -3 | tee $O/test.log
for m in 0 1 2; do
  echo "== texture storage $m" | tee -a $O/byte_textures.txt
  TEXTURE_STORAGE=$m timeout 200 python tools/c4_ablation.py 128 full 2>&1 | grep "^full" | sed 's/^/C4 /' | tee -a $O/byte_textures.txt
  TEXTURE_STORAGE=$m timeout 200 python tools/c5_ablation.py 512 full 2>&1 | grep "^full" | sed 's/^/C5 /' | tee -a $O/byte_textures.txt
done

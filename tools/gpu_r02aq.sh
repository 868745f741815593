#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WL=c3 bash tools/profile_c2.sh r02i_c3 512 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_r02i_c3 gpurun_out/r02i_c3_512spp.json 471859200 2>&1 | tail -1
WL=c4 bash tools/profile_c2.sh r02i_c4 64 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_r02i_c4 gpurun_out/r02i_c4_64spp.json 530841600 2>&1 | tail -1
rm -rf gpurun_out/prof_r02i_c3/*/*.db gpurun_out/prof_r02i_c4/*/*.db 2>/dev/null

#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
python tools/occupancy.py base 4096 4097 4116 4117 5128 5129 7176 h0 h4 h8 2>&1 | grep -v amdgpu.ids | tee $O/occupancy.txt
python tools/occupancy.py probe 4097 4117 5129 7177 2>&1 | grep -v amdgpu.ids | tee -a $O/occupancy.txt
python tools/occupancy.py probe2 4097 2>&1 | grep -v amdgpu.ids | tee -a $O/occupancy.txt

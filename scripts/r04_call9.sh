#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_front_phases.py 2>&1 | grep -v amdgpu.ids > $O/c9_front_phases.log
cat $O/c9_front_phases.log

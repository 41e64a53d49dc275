#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_phases.py full nosave bwd > $O/c4_phases.log 2>&1
cat $O/c4_phases.log

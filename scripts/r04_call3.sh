#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_phases.py full nosave bwd > $O/c3_phases.log 2>&1
cat $O/c3_phases.log
GEOM=4 SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chain_big_timing.py full nosave bwd > $O/c3_phases_g4.log 2>&1
cat $O/c3_phases_g4.log
timeout 200 python scripts/chainq_timing.py 4 7 > $O/c3_timing.log 2>&1
cat $O/c3_timing.log

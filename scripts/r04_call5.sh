#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for st in 0 3 6 12; do
echo "== stagger $st"
SWN_CHAINQ_STAGGER=$st SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_phases.py nosave full 2>&1 | grep -v amdgpu.ids
SWN_CHAINQ_STAGGER=$st timeout 200 python scripts/chainq_timing.py 7 2>&1 | grep "segments 16"
done > $O/c5.log 2>&1
timeout 200 python scripts/chainq_timing.py 4 2>&1 | grep "segments 16" >> $O/c5.log
cat $O/c5.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_front_phases.py 2>&1 | grep -v amdgpu.ids > $O/c10_front_phases.log
cat $O/c10_front_phases.log
SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_phases.py nosave 2>&1 | grep -v amdgpu.ids > $O/c10_phases.log
cat $O/c10_phases.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "front_chains_on_the_persistent or (geometry_bit_exact and (6 or 7))" 2>&1 | tail -3
timeout 200 python scripts/chainq_timing.py 4 7 2>&1 | grep "segments 16"

#!/bin/bash
# round 6, GPU call AF: the one-launch routing with a segment's tiles on ONE XCD (swn_route_top1x mode 3, experiment build): the sync
# primitive's price (xcd_sync_probe), the twin test on the experiment build, time per call against modes 0 / 1 / 2, a stress
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 120 scripts/experiments/xcd_sync_probe 2000 > $O/af_xcd_sync_probe.txt 2>&1; cat $O/af_xcd_sync_probe.txt
export SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_routeone.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "route" 2>&1 | tail -6
timeout 600 python scripts/experiments/route_mode_bench.py 60 2>&1 | grep -v amdgpu.ids | tee $O/af_route_modes.txt

#!/bin/bash
# round 6, GPU call B: which RCCL calls survive hipGraph capture (tests/test_rccl_gpu.py died with SIGSEGV in the captured padded step)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python scripts/experiments/rccl_capture_probe.py > $O/b_probe.log 2>&1
cat $O/b_probe.log
timeout 600 python tests/rccl_ep_worker.py 5 > $O/b_worker.log 2>&1; echo "worker rc $?"; tail -30 $O/b_worker.log

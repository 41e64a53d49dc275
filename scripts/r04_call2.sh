#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "geometry_bit_exact and (6 or 7)" 2>&1 | tail -30 > $O/c2_pytest.log
echo "pytest rc=$?" >> $O/c2_pytest.log
tail -25 $O/c2_pytest.log
timeout 300 python scripts/chainq_timing.py 4 7 > $O/c2_timing.log 2>&1
cat $O/c2_timing.log | tail -20

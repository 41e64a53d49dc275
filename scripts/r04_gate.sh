#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -x -m gpu -k "gate or mission" 2>&1 | tail -3
bash scripts/r04_prof_recipe.sh mb2 --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 | head -12
timeout 300 python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced 2>/dev/null | tail -1 | cut -c1-170

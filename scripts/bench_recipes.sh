#!/bin/bash
# bench.py lines of every recipe on the current build -> gpurun_out/${R:-r05}_bench_recipes.md   (bash scripts/bench_recipes.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/${R:-r05}_bench_recipes.md
echo '# Round ${R:-r05} - bench.py lines of every recipe, final build, one MI355X (`python bench.py <flags> --no-cpu-baseline --no-balanced --steps 10 --warmup 3`)' > $O
echo >> $O; echo 'Only the first line (no flags) is the headline metric (BASELINE configs[1]); the others are informational recipes of the same build.' >> $O
echo >> $O; echo '```' >> $O
run() { echo "### bench.py $*" >> $O; timeout 280 python bench.py "$@" --no-cpu-baseline --no-balanced --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-2400 >> $O; }
run
run --bg
run --bg --fine 512
run --dense
run --hash --capacity-factor 1.25
run --hash --capacity-factor 1.25 --dtype fp16
run --eval
run --fine 512
run --mip --samples 257
run --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16
run --dtype fp16
echo '```' >> $O

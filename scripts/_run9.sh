set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 280 python bench.py --hash --capacity-factor 1.25 --no-cpu-baseline --no-balanced --steps 10 --warmup 3 2>&1 | tail -3 | cut -c1-400

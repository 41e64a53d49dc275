cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 python scripts/graph_flake_probe.py graph 40 2>&1 | grep -v amdgpu.ids | tail -12
timeout 500 python scripts/graph_flake_probe.py eager 40 2>&1 | grep -v amdgpu.ids | tail -12

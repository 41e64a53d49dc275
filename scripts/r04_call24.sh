#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 python scripts/bw_probe2.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo default; timeout 200 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | tail -4
echo nt; SWN_LIB=switch_nerf_amd/libswn_hip_wgnt.so timeout 200 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | tail -4
done

#!/bin/bash
# bash scripts/isa_check.sh <file.hip> <kernel-name pattern> [extra flags]: gfx950 assembly of one source under /tmp/isa, register /
# spill counts of the kernels matching the pattern and where their scratch accesses sit between barriers / MFMAs / stores.
set -e
F=$1; PAT=$2; shift 2
mkdir -p /tmp/isa
B=$(basename $F .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=fast-honor-pragmas "$@" --cuda-device-only -S /root/repo/switch_nerf_amd/csrc/$B.hip -o /tmp/isa/$B.s 2>&1 | grep -v "hip-link" | head -30
grep -n "\.name:\|\.vgpr_count\|vgpr_spill_count\|sgpr_spill_count" /tmp/isa/$B.s | paste - - - - | sed 's/ \+/ /g' | grep "$PAT" | sed 's/^[0-9]*: //'
N=$(grep "\.name:" /tmp/isa/$B.s | grep "$PAT" | head -1 | awk '{print $2}')
python3 /root/repo/scripts/isa_phases.py /tmp/isa/$B.s $N

#!/bin/bash
# copies what scripts/collect_profiles_r06.sh left under gpurun_out/r06/ into profiles/ (the tracked copies) and rebuilds profiles/traffic.json
# + profiles/r06_hbm_traffic.md from the PMC passes (scripts/make_traffic.py ties them to the kernel sources' hash)
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r06
cp $O/r06_pmc_FETCH_SIZE.txt $O/r06_pmc_WRITE_SIZE.txt $O/r06_pmc_l2_chains.txt $O/r06_pmc_sq_experts.txt profiles/
cp $O/r06_kernel_stats_step_no_overlap.md $O/r06_kernel_stats_1024rays.md $O/r06_kernel_stats_hash.md $O/r06_kernel_stats_mission_bay.md profiles/
cp $O/r06_bench_default.json $O/r06_bench_20_5.json $O/r06_bench_1024rays_graph.json $O/r06_bench_1024rays_eager.json $O/r06_bench_eval_graph.json \
   $O/r06_bench_loopback_dp_1024rays.json $O/r06_bench_loopback_ep.json $O/r06_bench_loopback_ep_owner_tail.json profiles/r06_bench/
cp $O/r06_bench_recipes.txt profiles/r06_bench_recipes.md
cp $O/r06_gate_check_512x16.txt $O/r06_gate_check_512x16_fp16.txt profiles/
python scripts/make_traffic.py r06 > /dev/null
python - <<PY
import json, re
O = "$O/"
for f in ["r06_bench_default", "r06_bench_20_5", "r06_bench_1024rays_graph", "r06_bench_1024rays_eager", "r06_bench_eval_graph", "r06_bench_loopback_dp_1024rays",
          "r06_bench_loopback_ep", "r06_bench_loopback_ep_owner_tail"]:
    j = json.loads([l for l in open(O + f + ".json").read().splitlines() if l.startswith("{")][-1])
    r = j.get("roofline") or {}
    print(f, j["ms_per_step"], j["value"], j["config"].get("csrc_sha256", "")[:10], "frac", r.get("frac"), "traffic", r.get("traffic"))
name = None
for l in open(O + "r06_bench_recipes.txt"):
    if l.startswith("###"): name = l.strip()
    else:
        m = re.search(r'"ms_per_step": ([0-9.]+)', l); print(name, m.group(1) if m else "ERR")
import sys; sys.path.insert(0, ".")
from switch_nerf_amd import _lib
print("sources", _lib.source_hash()[:10], "traffic.json", json.load(open("profiles/traffic.json"))["csrc_sha256"][:10])
PY

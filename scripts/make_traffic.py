#!/usr/bin/env python
"""profiles/traffic.json from the round's PMC text files - by script, not by hand.

    python scripts/make_traffic.py r05          # reads profiles/r05_pmc_FETCH_SIZE.txt + profiles/r05_pmc_WRITE_SIZE.txt

Each input is the output of scripts/pmc_summary.py over ONE counter-only rocprofv3 pass of
    SWN_NO_OVERLAP=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --no-events --graph off
(the TIMED workload, the router's own routing; ONE step, so that every dispatch of the pass has the kept rows its bench line reports) followed by that pass's bench JSON line, which carries the kept rows of the pass and
`config.csrc_sha256` = the hash of the kernel sources the pass ran (bench.py refuses the table - `roofline.traffic: null` - when the
sources it runs differ).  Corrected HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950: FETCH_SIZE counts 128-byte read
requests at 64 bytes; /opt/skills/guides/MI355X_MICROARCH.md, calibrated on gate_fwd_mfma = one [P, 256] bf16 read)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launch of bench.py's `kernels` table -> the kernels whose counters add up to it (substring match on the demangled name)
LAUNCHES = {
    "expert_fwd": ["chainq_kernel<swn_big::Bf16, 7, true>"],
    "expert_bwd": ["chainq_kernel<swn_big::Bf16, 8, true>", "dwsig_runs_kernel"],
    "expert_wgrad": ["wgrad_stream_kernel<unsigned short, 1>"],
    "front_fwd": ["chainq_kernel<swn_big::Bf16, 3, true>"],
    "front_bwd": ["chainq_kernel<swn_big::Bf16, 6, true>"],
    "dense_wgrad": ["wgrad_stream_kernel<unsigned short, 0>"],
    "gate_fwd": ["gate_fwd_mfma_kernel"],
    "gate_bwd": ["gate_bwd_mfma_kernel"],
    "heads_bwd": ["heads_bwd_kernel"],
    "sample_pe": ["sample_pe_kernel"],
}


def parse(path, counter):
    """{kernel name: (mean counter value per dispatch, dispatches)}, bench JSON line or None"""
    tab, line_json, cur = {}, None, None
    for ln in open(path):
        if ln.startswith("{"):
            try:
                line_json = json.loads(ln)
            except Exception:
                pass
            continue
        if not ln.startswith(" "):
            cur = ln.strip()
            continue
        m = re.match(r"\s+(\S+)\s+([0-9.eE+-]+)\s+\(n=(\d+)\)", ln)
        if m and m.group(1) == counter and cur:
            tab[cur] = (float(m.group(2)), int(m.group(3)))
    return tab, line_json


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    f_path, w_path = (os.path.join(ROOT, "profiles", f"{rnd}_pmc_{c}.txt") for c in ("FETCH_SIZE", "WRITE_SIZE"))
    fetch, jf = parse(f_path, "FETCH_SIZE")
    write, jw = parse(w_path, "WRITE_SIZE")
    if jf is None or jw is None:
        raise SystemExit("make_traffic: a PMC file lacks its bench JSON line (kept rows / csrc_sha256 of the pass)")
    cf, cw = jf["config"], jw["config"]
    if cf.get("csrc_sha256") != cw.get("csrc_sha256"):
        raise SystemExit("make_traffic: the two passes ran different kernel sources")
    P = cf["rays_per_gpu"] * cf["samples"]
    kept = {"FETCH_SIZE": cf["kept_token_fraction"] * P, "WRITE_SIZE": cw["kept_token_fraction"] * P}
    out = {"_how": f"scripts/make_traffic.py {rnd}: 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch, mean over the pass) of "
                   f"profiles/{rnd}_pmc_FETCH_SIZE.txt / _WRITE_SIZE.txt; the passes ran ONE step of the TIMED workload (router's routing: the first step after the reset keeps ~52 % of the tokens); "
                   "bench.py scales by the ratio of the algorithmic bytes to its own run's kept rows",
           "csrc_sha256": cf.get("csrc_sha256"), "points": P, "kept_rows": round((kept["FETCH_SIZE"] + kept["WRITE_SIZE"]) / 2),
           "kept_token_fraction": {k: round(v / P, 4) for k, v in kept.items()}, "kernel_set": cf.get("kernel_set"), "launches": {}}
    for name, pats in LAUNCHES.items():
        fk = sum(v[0] for k, v in fetch.items() if any(p in k for p in pats))
        wk = sum(v[0] for k, v in write.items() if any(p in k for p in pats))
        if fk == 0 and wk == 0:
            continue
        out["launches"][name] = {"fetch_kib": fk, "write_kib": wk, "hbm_bytes": int((2 * fk + wk) * 1024),
                                 "read_bytes": int(2 * fk * 1024), "write_bytes": int(wk * 1024)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    # the same numbers as a table (profiles/<round>_hbm_traffic.md)
    kept_rows = out["kept_rows"]
    md = [f"# Round {rnd[1:].lstrip('0')} - HBM traffic per launch from the PMC counters (made by scripts/make_traffic.py {rnd}; do not edit)", "",
          f"Two counter-only passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; `rocprofv3 --kernel-trace --pmc <counter>`, `SWN_NO_OVERLAP=1`) of ONE",
          f"step of the timed workload: {P} points, {kept_rows} kept rows ({kept_rows / P:.4f}); kernel sources `{str(out['csrc_sha256'])[:16]}`.",
          "Corrected traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch; gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes).", "",
          "| launch (bench.py name) | kernels | FETCH_SIZE KiB | WRITE_SIZE KiB | read GB | written GB | HBM GB | bytes per point | bytes per kept row |",
          "|---|---|---|---|---|---|---|---|---|"]
    for name, t in out["launches"].items():
        md.append(f"| {name} | {' + '.join('`' + q + '`' for q in LAUNCHES[name])} | {t['fetch_kib']:.4e} | {t['write_kib']:.4e} | {t['read_bytes'] / 1e9:.3f} | "
                  f"{t['write_bytes'] / 1e9:.3f} | {t['hbm_bytes'] / 1e9:.3f} | {t['hbm_bytes'] / P:.0f} | {t['hbm_bytes'] / kept_rows:.0f} |")
    md += ["", "bench.py reports `hbm_measured_bytes` = these bytes x (algorithmic bytes at its own run's kept rows / algorithmic bytes at the pass's kept",
           "rows) and `hbm_measured_over_alg`; it reports nothing (`roofline.traffic: null`) when the kernel sources or the kernel set differ.", ""]
    open(os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.md"), "w").write("\n".join(md))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

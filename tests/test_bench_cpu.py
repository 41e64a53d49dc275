"""bench.py on a machine without a GPU: it must fail loudly (no CPU fallback of the product path), for one rank and for a self-launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_bench_refuses_without_a_gpu():
    one = _run("--steps", "1", "--warmup", "0")
    assert one.returncode != 0 and "visible" in one.stderr and not any(l.startswith("{") for l in one.stdout.splitlines())
    many = _run("--gpus", "4", "--steps", "1", "--warmup", "0")
    assert many.returncode != 0 and "GPU(s) are visible" in many.stderr and not any(l.startswith("{") for l in many.stdout.splitlines())


def test_traffic_table_is_made_by_script_and_tied_to_the_kernel_sources(tmp_path):
    """scripts/make_traffic.py turns the two PMC summaries (+ the bench line of each pass) into profiles/traffic.json: corrected bytes
    = 2 x FETCH + WRITE KiB per launch, the kept rows and the source hash of the pass; _lib.source_hash() is stable and changes with
    the sources (bench.py reports traffic only when the hashes agree)."""
    import importlib.util
    import json
    import shutil
    sys.path.insert(0, ROOT)
    from switch_nerf_amd import _lib
    h = _lib.source_hash()
    assert len(h) == 64 and h == _lib.source_hash()
    spec = importlib.util.spec_from_file_location("make_traffic", os.path.join(ROOT, "scripts", "make_traffic.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    line = json.dumps({"config": {"rays_per_gpu": 8192, "samples": 256, "kept_token_fraction": 0.78, "csrc_sha256": h, "kernel_set": {"geom": 7}}})
    (tmp_path / "profiles").mkdir()
    for c, v7, v8 in (("FETCH_SIZE", 1.0e6, 2.0e6), ("WRITE_SIZE", 9.0e6, 9.5e6)):
        (tmp_path / "profiles" / f"rXX_pmc_{c}.txt").write_text(
            f"void swn_big::chainq_kernel<swn_big::Bf16, 7, true>(swn_big::ArgsQ)\n   {c}   {v7:.4e}  (n=3)\n"
            f"void swn_big::chainq_kernel<swn_big::Bf16, 8, true>(swn_big::ArgsQ)\n   {c}   {v8:.4e}  (n=3)\n"
            f"swn::dwsig_runs_kernel(float const*, long, int, float*)\n   {c}   1.0000e+02  (n=3)\n" + line + "\n")
    mt.ROOT = str(tmp_path)
    argv = sys.argv
    try:
        sys.argv = ["make_traffic.py", "rXX"]
        mt.main()
    finally:
        sys.argv = argv
    t = json.load(open(tmp_path / "profiles" / "traffic.json"))
    assert t["csrc_sha256"] == h and t["points"] == 8192 * 256 and t["kept_rows"] == round(0.78 * 8192 * 256)
    assert t["launches"]["expert_fwd"]["hbm_bytes"] == int((2 * 1.0e6 + 9.0e6) * 1024)
    assert t["launches"]["expert_bwd"]["hbm_bytes"] == int((2 * (2.0e6 + 100) + 9.5e6 + 100) * 1024)


def test_kernel_switches_are_resolved_once_from_the_environment():
    """model.resolve_kernel_switches: the SWN_* knobs parse into the switch set a model keeps (no forward / backward reads os.environ:
    VERDICT round 5 item 9); an empty environment = the shipped kernel set."""
    from switch_nerf_amd.model import resolve_kernel_switches, SwitchNeRF
    d = resolve_kernel_switches({})
    assert d == dict(front_geom=7, chain_geom=7, tail_geom=1, fused_tail=True, fused_tail_bwd=True, fused_dwsig=True, fused_heads=True, overlap=True)
    e = resolve_kernel_switches({"SWN_CHAIN_GEOM": "5", "SWN_FUSED_TAIL": "0", "SWN_NO_OVERLAP": "1", "SWN_NO_FUSED_HEADS": "1", "SWN_FRONT_GEOM": "1"})
    assert (e["chain_geom"], e["fused_tail"], e["overlap"], e["fused_heads"], e["front_geom"]) == (5, False, False, False, 1)
    import inspect
    src = inspect.getsource(SwitchNeRF)
    body = src[src.index("def forward_rays"):]
    assert "os.environ" not in body, "forward / backward must not read the environment"

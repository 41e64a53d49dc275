"""bench.py's multi-rank flow on a real GPU: two ranks share cuda:0 and talk over gloo (RCCL needs one GPU per rank; the
collectives are backend-agnostic torch.distributed calls).  Data parallel and expert parallel must train identically: routing is
rank-local in both modes; under expert parallelism the experts are SHARDED (a rank owns, updates and check-points E / W of them, the
dense parameters stay replicated and all-reduced) and only kept rows travel - after two optimizer steps both modes report the same loss."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(parallelism, port, extra=()):
    env = dict(os.environ, SWN_DIST_BACKEND="gloo", SWN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--rays", "1024",
           "--parallelism", parallelism, "--no-events"] + list(extra)
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    return json.loads(lines[0])


def test_two_ranks_dp_and_ep_agree():
    dp = _run("dp", 29561)
    ep = _run("ep", 29562)
    assert dp["n_gpus"] == ep["n_gpus"] == 2 and dp["config"]["parallelism"] == "dp2" and ep["config"]["parallelism"] == "ep2"
    assert dp["scaling"] == "strong" and dp["cpu_baseline"] is None        # N > 1 default: the batch is split over the ranks (runner.py:575)
    assert dp["config"]["global_batch_rays"] == 1024 and dp["config"]["rays_per_gpu"] == 512
    # Every gradient sum of a rank has a fixed order (no atomics since round 3), so each mode is bit-reproducible; the two modes add
    # the SAME terms in different orders (dp: per-rank partial sums met by the all-reduce; ep: the owner sums the rows of both ranks
    # in one pass), i.e. they differ by fp32 rounding of the gradients (1e-7 relative), which one Adam step carries into the second
    # step's loss at the 1e-6 level (bf16 activations: an occasional rounding flip).  Bound: 5e-5 relative (round 3: 5e-4).
    assert abs(dp["config"]["loss"] - ep["config"]["loss"]) <= 5e-5 * abs(dp["config"]["loss"])
    assert abs(dp["config"]["kept_token_fraction"] - ep["config"]["kept_token_fraction"]) < 1e-3
    assert dp["value"] > 0 and ep["value"] > 0
    # the data-parallel line carries the expert-parallel measurement of the same batch (the driver's scaling run gets one invocation per
    # N: north_star's all-to-all must not need a second one) - rays/s, bytes per GPU, hidden fraction, the rows every rank's experts ran
    xd = dp["config"]["expert_parallel"]
    assert "error" not in xd, xd
    assert xd["value"] > 0 and xd["ms_per_step"] > 0 and xd["collectives_per_step"] == 4 * xd["segments"] and xd["hidden_fraction"] is not None
    assert len(xd["expert_rows_per_rank"]) == 2 and min(xd["expert_rows_per_rank"]) > 0 and xd["load_imbalance_max_over_mean"] >= 1.0
    assert 0 < xd["bytes_leaving_this_gpu_per_step"] < xd["capacity_padded_bytes_per_step"]
    assert dp["config"]["kernel_set"]["expert_parallel"] in (0, 2) and len(dp["config"]["csrc_sha256"]) == 64
    x = ep["config"]["expert_parallel"]
    # kept rows only: what leaves a GPU is (W - 1) / W of 4 exchanges of the kept rows, not of the capacity-padded payload
    assert x["segments"] >= 1 and 0 < x["bytes_leaving_this_gpu_per_step"] < x["capacity_padded_bytes_per_step"]
    full = 4 * x["kept_rows_per_step"] * 256 * 2          # every kept row, four exchanges, bf16 rows of 256 features
    assert 0.1 * full <= x["bytes_leaving_this_gpu_per_step"] <= 0.9 * full      # ~ (W - 1) / W of it, depending on where the experts sit
    assert x["collectives_per_step"] == 4 * x["segments"] and x["hidden_fraction"] is not None
    # the padded (host-free, capturable) mode of the exchange: capacity-padded equal splits like the reference - same training, the
    # capacity-padded payload on the wire
    pad = _run("ep", 29563, ("--ep-padded", "on"))
    assert abs(dp["config"]["loss"] - pad["config"]["loss"]) <= 5e-5 * abs(dp["config"]["loss"])
    xp = pad["config"]["expert_parallel"]
    assert xp["bytes_leaving_this_gpu_per_step"] == xp["capacity_padded_bytes_per_step"] and xp["collectives_per_step"] == 4 * xp["segments"]


def test_two_ranks_owner_tail_expert_parallel_trains_like_data_parallel_with_half_the_bytes():
    """ExpertParallel(owner_tail=True) on two ranks (one GPU, gloo): the experts' owner runs the fused launches on the tokens it receives
    from both ranks; after two optimizer steps the loss equals the data-parallel run's, and what leaves a GPU is <= 0.55 x the kept-rows
    mode's bytes (kept rows + 16 B per token out, 32 B per token back; 16 B out, dx + 4 B back)."""
    dp = _run("dp", 29564, ("--no-ep-probe",))
    ep = _run("ep", 29565)
    ot = _run("ep", 29566, ("--ep-owner-tail",))
    assert abs(dp["config"]["loss"] - ot["config"]["loss"]) <= 5e-5 * abs(dp["config"]["loss"])
    assert abs(dp["config"]["kept_token_fraction"] - ot["config"]["kept_token_fraction"]) < 1e-3
    x, y = ep["config"]["expert_parallel"], ot["config"]["expert_parallel"]
    assert y["owner_tail"] and not x["owner_tail"] and ot["config"]["kernel_set"].get("ep_owner_tail")
    assert 0 < y["bytes_leaving_this_gpu_per_step"] <= 0.55 * x["bytes_leaving_this_gpu_per_step"], (x["bytes_leaving_this_gpu_per_step"], y["bytes_leaving_this_gpu_per_step"])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without torchrun's environment must start N ranks itself (the driver calls it that way) - and
    must refuse, not silently run one rank, when the node has fewer devices."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SWN_DIST_BACKEND="gloo", SWN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "1024", "--no-events",
           "--scaling", "weak"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch_rays"] == 2048 and j["value"] > 0
    env.pop("SWN_FORCE_DEVICE")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "GPU(s) are visible" in bad.stderr


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_expert_parallel_evaluation_unequal_splits(dtype):
    """Evaluation without token dropping, experts sharded over two ranks: the packed rows travel with unequal splits (the reference's
    list_all_to_all, tutel_communicate_nobatch.py:18-51) and the result equals the single-rank forward bit for bit (its tail as the
    64-row launch the expert-parallel path uses: SWN_FUSED_TAIL=0)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SWN_FUSED_TAIL="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "tests", "ep_eval_worker.py"), dtype]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("EP_EVAL rank") == 2 and "MISMATCH" not in out.stdout


def test_expert_parallel_checkpoint_equals_data_parallel_checkpoint():
    """Two ranks, two optimizer steps with sharded experts, checkpoint.save_checkpoint on every rank (which gathers the expert shards
    from their owners): parameters and Adam moments equal the data-parallel run's on both ranks."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(ROOT, "tests", "ep_ckpt_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("EP_CKPT rank") == 2 and "MISMATCH" not in out.stdout


def test_split_backward_graphs_with_overlapped_allreduce_are_bit_identical():
    """Two ranks on one GPU: the data-parallel step as two backward graphs with the expert block's all-reduce issued between them on
    the side stream == the single-graph step == the eager step, bit for bit, over three optimizer steps (tests/dp_overlap_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29575", os.path.join(ROOT, "tests", "dp_overlap_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("DP_OVERLAP rank") == 2 and "MISMATCH" not in out.stdout

"""RCCL (torch.distributed backend "nccl" on ROCm) carries the multi-GPU step's collectives on the ONE GPU a test box has: a world-1
process group in loopback mode (parallel.init_from_env(loopback=True)) - the collectives are not short-circuited, every
all_to_all_single / all_reduce of the W > 1 code path is issued with its streams, events, split sizes and (padded expert parallelism,
split backward graphs) inside / between hipGraph captures.  What this proves: the API / stream / capture contract with RCCL, which is
where a first 8-GPU run would die; what it cannot: xGMI bandwidth or W > 1 semantics (gloo world-2 tests cover those:
tests/test_parallel_cpu.py, tests/test_parallel_gpu.py).  Reference slots: tutel_moe_layer_nobatch.py:157-185 (all-to-all around the
experts), runner.py:203-207 (DDP's gradient buckets)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SWN_DIST_BACKEND", "SWN_FORCE_DEVICE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    return env


def test_rccl_expert_parallel_collectives_and_captured_padded_step():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_ep_worker.py"), "50"], env=_env(29581), cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2500:], out.stderr[-2500:])
    assert "RCCL_EP done: OK" in out.stdout and "MISMATCH" not in out.stdout and out.stdout.count(": OK") >= 9


def test_rccl_split_backward_graphs_with_overlapped_allreduce_50_steps():
    """graph.GraphedTrainStep(split_backward=True): the expert block's all-reduce issued through RCCL on the side stream BETWEEN the two
    backward graphs, the dense prefix behind the second - 50 optimizer steps bit-identical to the one-graph and the eager step."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_overlap_worker.py"), "nccl", "50"], env=_env(29583), cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2500:], out.stderr[-2500:])
    assert "DP_OVERLAP rank 0 (nccl, 50 steps): OK" in out.stdout


def _bench(port, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--loopback", "--steps", "3", "--warmup", "1", "--rays", "1024",
           "--no-events", "--no-balanced", "--no-cpu-baseline"] + list(extra)
    out = subprocess.run(cmd, env={k: v for k, v in _env(port).items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_loopback_expert_parallel_over_rccl():
    """`bench.py --gpus 1 --loopback --parallelism ep`: the expert-parallel step (kept rows, unequal splits; then the padded, captured
    form) with RCCL underneath; the data-parallel loopback line carries the all-reduce report and the expert-parallel probe."""
    ep = _bench(29585, ["--parallelism", "ep"])
    x = ep["config"]["expert_parallel"]
    assert ep["config"]["parallelism"] == "ep1-loopback" and ep["value"] > 0 and x["collectives_per_step"] == 4 * x["segments"]
    assert x["hidden_fraction"] is not None and x["bytes_leaving_this_gpu_per_step"] == 0        # (one rank: nothing leaves the GPU)
    pad = _bench(29586, ["--parallelism", "ep", "--ep-padded", "on", "--graph", "on"])
    assert pad["value"] > 0 and abs(pad["config"]["loss"] - ep["config"]["loss"]) <= 5e-5 * abs(ep["config"]["loss"])
    dp = _bench(29587, [])
    assert dp["config"]["parallelism"] == "dp1-loopback" and dp["config"]["allreduce"]["collectives_per_step"] == 2
    assert dp["config"]["allreduce"]["allreduce_hidden_fraction"] is not None
    xd = dp["config"]["expert_parallel"]
    assert "error" not in xd and xd["value"] > 0
    assert abs(dp["config"]["loss"] - ep["config"]["loss"]) <= 5e-5 * abs(dp["config"]["loss"])

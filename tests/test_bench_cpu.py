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

"""Which RCCL calls survive hipGraph capture on this ROCm / torch build?  One subprocess per variant (a crash must not take the
others down); world-1 process group (loopback).  python scripts/experiments/rccl_capture_probe.py [variant]"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = ["allreduce_sync", "allgather_sync", "reduce_scatter_sync", "a2a_sync_eager", "a2a_sync", "a2a_sync_delgraph", "a2a_sync_abort", "ep_all_to_all_v_workaround", "a2a_list", "p2p_batch", "a2a_async_same_stream",
            "a2a_async_side_stream", "a2a_sync_side_stream", "ep_all_to_all_v", "ep_exchange_counts", "ep_all_to_all_v_eager"]


def run(variant):
    faulthandler.enable()
    faulthandler.dump_traceback_later(40, exit=True)          # a hang: the Python stack of every thread, then exit
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from switch_nerf_amd import parallel
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    parallel.init_from_env("nccl", dev, loopback=True)
    x = torch.randn(4096, 256, device=dev).to(torch.bfloat16)
    y = torch.zeros_like(x)
    side = torch.cuda.Stream(device=dev)
    ep = parallel.ExpertParallel(0, 1, 8, padded=True, loopback=True)
    counts = torch.randint(0, 400, (4, 8), device=dev, dtype=torch.int32)
    out = {}

    def body():
        if variant == "allreduce_sync":
            dist.all_reduce(x)
            y.copy_(x)
        elif variant in ("a2a_sync", "a2a_sync_eager", "a2a_sync_delgraph", "a2a_sync_abort"):
            dist.all_to_all_single(y, x)
        elif variant == "allgather_sync":
            dist.all_gather_into_tensor(y, x)
        elif variant == "reduce_scatter_sync":
            dist.reduce_scatter_tensor(y, x)
        elif variant == "a2a_list":
            dist.all_to_all([y], [x])
        elif variant == "p2p_batch":
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0)]):
                w.wait()
        elif variant == "a2a_async_same_stream":
            w = dist.all_to_all_single(y, x, async_op=True)
            w.wait()
        elif variant in ("a2a_async_side_stream", "a2a_sync_side_stream"):
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if variant == "a2a_async_side_stream":
                    w = dist.all_to_all_single(y, x, async_op=True)
                else:
                    dist.all_to_all_single(y, x)
                    done = torch.cuda.Event()
                    done.record()
            if variant == "a2a_async_side_stream":
                w.wait()
            else:
                torch.cuda.current_stream().wait_event(done)
        elif variant in ("ep_all_to_all_v", "ep_all_to_all_v_eager", "ep_all_to_all_v_workaround"):
            ep.all_to_all_v(x, [4096], y, [4096], side)()
        elif variant == "ep_exchange_counts":
            out["rc"] = ep.exchange_counts(counts, 256, side)()
    # warm-up (eager) on a capture-style side stream
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(f"{variant}: eager ok", flush=True)
    if variant.endswith("_eager"):
        print(f"RESULT {variant}: OK", flush=True)
        return
    y.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    print(f"{variant}: captured", flush=True)
    for _ in range(3):
        y.zero_()
        g.replay()
    torch.cuda.synchronize()
    good = torch.equal(y, x) if variant != "ep_exchange_counts" else torch.equal(out["rc"], counts.clamp(max=256))
    print(f"RESULT {variant}: {'OK' if good else 'WRONG'}", flush=True)
    if variant.endswith("_delgraph"):
        import gc
        del g
        gc.collect()
        torch.cuda.synchronize()
    if variant.endswith("_abort"):
        sys.stdout.flush()
        os._exit(0)
    dist.destroy_process_group()
    print(f"{variant}: process group destroyed", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] != "--only":
        run(sys.argv[1])
    else:
        for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else VARIANTS):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=120)
                rc, so, se = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as e:
                rc, so, se = "TIMEOUT", (e.stdout or b"").decode(errors="replace"), (e.stderr or b"").decode(errors="replace")
            res = [l for l in so.splitlines() if l.startswith("RESULT")]
            print(f"== {v}: rc {rc} {res[-1] if res else 'NO RESULT'}", flush=True)
            if rc != 0 or not res:
                print("   stdout:", so[-600:].replace("\n", "\n   "))
                print("   stderr:", se[-3000:].replace("\n", "\n   "), flush=True)

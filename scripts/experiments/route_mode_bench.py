"""EXPERIMENT (profiles/r06_experiments.md 11): swn_route_top1x modes of the experiment build (SWN_LIB=.../libswn_hip_routeone.so) - time per call
(HIP events around N back-to-back calls, the launch rate included: what a graph replay removes is bounded by the graphed column) and a
bit-equality stress against the per-phase launches.   python scripts/experiments/route_mode_bench.py [stress_rounds]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import synth
from switch_nerf_amd import ops as o

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def setup(n_seg, seg, E, seed, qb=0):
    g = torch.from_numpy(synth.make_gates(seed, n_seg * seg, E, 2.0, quantize_bits=qb)).to(dev)
    idx = g.argmax(1).int()
    gmax = g.gather(1, idx.long()[:, None])[:, 0].contiguous()
    return g, idx, gmax


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def graphed(fn, n=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    return timed(gr.replay, n) / 10


for n_seg, seg, E in [(2, 131072, 8), (16, 131072, 8), (4, 212992, 16)]:
    g, idx, gmax = setup(n_seg, seg, E, 500 + n_seg)
    cap = seg // E
    ref = o.route_top1(idx, gmax, g, seg, E, cap, True, want_drops=True, multi=True)
    nd = int(ref[5][-1].item())
    line = [f"{n_seg} x {seg} tokens, {E} experts:"]
    for mode in (0, 1, 2, 3):
        fn = lambda: o.route_top1(idx, gmax, g, seg, E, cap, True, want_drops=True, mode=mode)
        one = fn()
        ok = all(torch.equal(a, b) for a, b in zip(one[:6], ref[:6])) and torch.equal(one[6][:nd], ref[6][:nd])
        line.append(f"mode {mode}: eager {timed(fn):7.1f} us, graphed {graphed(fn):7.1f} us{'' if ok else '  MISMATCH'}")
    print("  ".join(line), flush=True)

# stress of mode 3: changing inputs every launch (tie-heavy ones among them), all outputs against the per-phase kernels
bad = 0
for r in range(rounds):
    n_seg, seg, E = [(2, 131072, 8), (16, 131072, 8), (3, 40000, 16), (9, 5000, 64), (8, 2049, 4), (1, 300, 8)][r % 6]
    g, idx, gmax = setup(n_seg, seg, E, 9000 + r, qb=(3 if r % 5 == 0 else 0))
    cap = max(1, int([1.0, 0.5, 1.25][r % 3] * ((seg + E - 1) // E)))
    ref = o.route_top1(idx, gmax, g, seg, E, cap, True, want_drops=True, multi=True)
    nd = int(ref[5][-1].item())
    for rep in range(4):
        one = o.route_top1(idx, gmax, g, seg, E, cap, True, want_drops=True, mode=3)
        ok = all(torch.equal(a, b) for a, b in zip(one[:6], ref[:6])) and torch.equal(one[6][:nd], ref[6][:nd])
        bad += 0 if ok else 1
print(f"mode 3 stress: {rounds * 4} launches, {bad} with a difference; sync words left: {sum(int(t.abs().sum().item()) for t in o._route_sync.values())}")

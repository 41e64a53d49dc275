"""How often do graph-replayed and eager training diverge within three optimizer steps (tests/test_graph_gpu.py), and where first?"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
from switch_nerf_amd.model import SwitchNeRF
from switch_nerf_amd.graph import GraphedTrainStep

d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
N, S, chunk = 512, 64, 8192
batches = [synth.make_rays(300 + i, N) for i in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
fails = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    ms = []
    for _ in range(2):
        m = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16)
        m.load_state_dict(synth.make_weights(41, synth.BUILDING))
        ms.append(m)
    ma, mb = ms
    step = None
    if mode == "graph":
        r0, i0, g0 = batches[0]
        step = GraphedTrainStep(ma, d(g0), d(r0), d(i0), S, chunk, perturb=0.0, noise_std=0.0)
        ma.load_state_dict(synth.make_weights(41, synth.BUILDING)); ma.m.zero_(); ma.v.zero_(); ma.step_count = 0; ma.refresh_compute_copies()
    for it, (rays, img, rgbs) in enumerate(batches):
        pd = (ma.flat - mb.flat).abs().max().item()
        ra = step(d(rgbs), d(rays), d(img)) if step is not None else ma.train_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0)
        ia = ra["ctx"]["idx"].clone()
        rb = mb.train_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0)
        mis = int((ia != rb["ctx"]["idx"]).sum().item())
        if mis:
            fails += 1
            names = []
            for name, (off, shape) in ma.spec.items():
                n = int(np.prod(shape))
                e = (ma.flat[off:off + n] - mb.flat[off:off + n]).abs().max().item()
                if e > 0:
                    names.append((name, e))
            print(f"trial {trial} step {it}: {mis} routing mismatches; max param diff BEFORE this step {pd:.3e}; after: {sorted(names, key=lambda t: -t[1])[:6]}", flush=True)
            break
print(f"FLAKE mode={mode}: {fails} failing trials")

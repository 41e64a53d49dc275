"""Which parameter gradients differ between two runs of the same step (same weights, same batch), and by how much.
python scripts/determinism_check.py [bf16|fp32] [rays] [graph]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
from switch_nerf_amd.model import SwitchNeRF

dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
S, chunk = 64, 8192
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rays, img, rgbs = synth.make_rays(300, N)
ms = []
for _ in range(2):
    m = SwitchNeRF(synth.BUILDING, dtype=dt)
    m.load_state_dict(synth.make_weights(41, synth.BUILDING))
    ms.append(m)
step = None
if use_graph:
    from switch_nerf_amd.graph import GraphedTrainStep
    step = GraphedTrainStep(ms[0], d(rgbs), d(rays), d(img), S, chunk, perturb=0.0, noise_std=0.0)
worst = {}
for rep in range(6):
    grads, idxs = [], []
    for k, m in enumerate(ms):
        if k == 0 and step is not None:
            st = step(optimizer_step=False)
        else:
            st = m.grad_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0)
        torch.cuda.synchronize()
        grads.append(m.grad.clone()); idxs.append(st["ctx"]["idx"].clone())
    assert torch.equal(idxs[0], idxs[1]), "routing differs with identical weights"
    for name, (off, shape) in ms[0].spec.items():
        n = int(np.prod(shape))
        a, b = grads[0][off:off + n], grads[1][off:off + n]
        e = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
        worst[name] = max(worst.get(name, 0.0), e)
for k, v in sorted(worst.items(), key=lambda kv: -kv[1]):
    print(f"{k:12s} {v:.3e}")

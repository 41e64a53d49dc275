"""Stress: the same training step (same weights, same batch) on two models, many repetitions; every gradient that comes from the
deterministic kernels (chains, weight gradients: xyz / gate0 / gate1 / l1 / l2h / exp*) must be BIT-identical in every repetition -
a hardware hazard in a store / an intermittent race would show as a rare non-zero difference.  Also compares the saved activations
of the two models when a difference is found.   python scripts/determinism_stress.py [reps] [rays] [samples] [chunk]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
from switch_nerf_amd.model import SwitchNeRF

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 131072
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rays, img, rgbs = synth.make_rays(300, N)
ms = []
for _ in range(2):
    m = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16)
    m.load_state_dict(synth.make_weights(41, synth.BUILDING))
    ms.append(m)
det = [k for k in ms[0].spec if k.split(".")[0] in ("xyz", "gate0", "gate1", "l1", "l2h") or k.startswith("exp")]
bad = 0
first = None
for rep in range(reps):
    cs = []
    for m in ms:
        st = m.grad_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0)
        cs.append(st["ctx"])
    torch.cuda.synchronize()
    diffs = []
    for name in det:
        off, shape = ms[0].spec[name]
        n = int(np.prod(shape))
        if not torch.equal(ms[0].grad[off:off + n], ms[1].grad[off:off + n]):
            diffs.append(name)
    if diffs:
        bad += 1
        if first is None:
            first = (rep, diffs)
            for key in ("h0", "a1", "g", "eo", "y", "h1", "h2", "raw"):
                a, b = cs[0][key], cs[1][key]
                ne = (a != b)
                print(f"rep {rep}: {key}: {int(ne.sum())} differing elements", flush=True)
            for l, (a, b) in enumerate(zip(cs[0]["saves"], cs[1]["saves"])):
                print(f"rep {rep}: save{l}: {int((a != b).sum())} differing elements (valid rows and padding)", flush=True)
print(f"STRESS reps={reps} rays={N} samples={S}: repetitions with a difference in a deterministic gradient: {bad}; first: {first}")

"""smoke(): one tiny train step on cuda:0 through the C ABI, checked against the CPU oracle."""
import numpy as np
import torch

import synth
from oracle import switchnerf_oracle as O


def run():
    from switch_nerf_amd.model import SwitchNeRF
    N, S, chunk = 32, 64, 1024
    sd = synth.make_weights(5, synth.BUILDING, gate_scale=0.02)
    rays, img, rgbs = synth.make_rays(6, N)
    m = SwitchNeRF(synth.BUILDING, dtype=torch.float32)
    m.load_state_dict(sd)
    d = lambda a: torch.from_numpy(a).cuda()
    st = m.train_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0, optimizer_step=True)
    p = O.params_from_numpy(sd)
    with torch.no_grad():
        ref = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk)
    err = (st["ctx"]["rgb"].cpu() - ref["results"]["rgb_coarse"]).abs().max().item()
    assert err < 1e-4, f"smoke: rgb differs from the oracle by {err}"
    assert abs(st["loss"].item() - ref["loss"].item()) < 1e-4 * abs(ref["loss"].item()) + 1e-7
    m16 = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16)
    m16.load_state_dict(sd)
    st16 = m16.train_step(d(rgbs), d(rays), d(img), S, chunk, perturb=0.0, optimizer_step=True)
    assert abs(st16["loss"].item() - ref["loss"].item()) < 5e-2 * abs(ref["loss"].item())
    torch.cuda.synchronize()
    print(f"smoke ok: fp32 rgb max err {err:.2e}, loss {st['loss'].item():.6f} (oracle {ref['loss'].item():.6f}), bf16 loss {st16['loss'].item():.6f}")

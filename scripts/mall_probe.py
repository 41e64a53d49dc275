"""Does the 256 MiB Infinity Cache keep the most recently WRITTEN data for the next kernel?  Write a 2 GiB tensor front to back, then time a
read of its last / first 128 MiB.  python scripts/mall_probe.py"""
import torch
dev = torch.device("cuda")
n_total, n_part = 2 << 28, 1 << 25            # fp32 elements: 2 GiB, 128 MiB
x = torch.empty(n_total, device=dev)
src = torch.randn(n_total, device=dev)
out = torch.empty(n_part, device=dev)
def t(fn, prep, reps=20):
    ms = []
    for _ in range(reps):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2]
write = lambda: x.copy_(src)
for name, sl in (("last 128 MiB (written most recently)", slice(n_total - n_part, n_total)), ("first 128 MiB (written 2 GiB ago)", slice(0, n_part)),
                 ("middle", slice(n_total // 2, n_total // 2 + n_part))):
    ms = t(lambda: torch.add(x[sl], 1.0, out=out), write)
    print(f"read {name}: {ms * 1e3:.1f} us = {n_part * 4 * 2 / ms / 1e9:.2f} TB/s (read + write of 128 MiB)")
ms = t(lambda: torch.add(x[:n_part], 1.0, out=out), lambda: torch.add(x[:n_part], 1.0, out=out))
print(f"read the same 128 MiB twice in a row: {ms * 1e3:.1f} us = {n_part * 4 * 2 / ms / 1e9:.2f} TB/s")

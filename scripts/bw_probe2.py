"""HBM read-only / write-only / copy rates on this box with plain torch kernels (4 GiB tensors: far beyond the 256 MiB Infinity Cache)."""
import torch
dev = torch.device("cuda")
n = 1 << 31                      # 2 G bf16 = 4 GiB
x = torch.randn(n // 2, device=dev, dtype=torch.float32).view(torch.bfloat16) if False else torch.ones(n, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


gb = n * 2 / 1e9
print(f"read-only  (sum of 4 GiB as int16)   : {gb / t(lambda: x.view(torch.int16).sum(dtype=torch.int64)):8.1f} GB/s")
print(f"read-only  (amax of 4 GiB as fp32)   : {gb / t(lambda: x.view(torch.float32).amax()):8.1f} GB/s")
print(f"write-only (fill_)                   : {gb / t(lambda: y.fill_(1.0)):8.1f} GB/s")
print(f"copy       (read + write)            : {2 * gb / t(lambda: y.copy_(x)):8.1f} GB/s")
print(f"two reads + one write (add)          : {3 * gb / t(lambda: torch.add(x, x, out=y)):8.1f} GB/s   (x read twice: one read may hit cache)")

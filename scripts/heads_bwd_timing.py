"""swn_heads_bwd at full size (2,097,152 points, 256 / 128 features, rows_per_group 256): time and a checksum of every output.
Usage: [SWN_LIB=...] python scripts/heads_bwd_timing.py [noy]"""
import sys, torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev = torch.device("cuda"); torch.manual_seed(0)
P, M, H2, S = 2097152, 256, 128, 256
y = torch.randn(P, M, device=dev).relu().bfloat16(); h2 = torch.randn(P, H2, device=dev).relu().bfloat16()
wc = torch.randn(3, H2, device=dev) * 0.1
raw = torch.rand(P, 4, device=dev); d_raw = torch.randn(P, 4, device=dev) * 1e-3
g = [torch.zeros(M, device=dev), torch.zeros(1, device=dev), torch.zeros(3, H2, device=dev), torch.zeros(3, device=dev)]
NOY = "noy" in sys.argv          # without y: the sigma weight gradient left to the fused backward chain (swn_chain_desc.comb_dwsig)
def run():
    return o.heads_bwd(None if NOY else y, h2, wc, raw, d_raw, *g, rows_per_group=S)
out = run(); torch.cuda.synchronize()
print("checksums", [float(t.double().sum()) for t in out], [float(t.double().sum()) for t in g])
best = 1e9
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 10)
nbytes = P * ((0 if NOY else M * 2) + H2 * 2 * 2 + 32 + 4)
print(f"heads_bwd {best*1e3:.1f} us  {nbytes/best/1e6:.0f} GB/s")

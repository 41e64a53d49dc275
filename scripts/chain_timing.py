"""Phase timing of the expert chain kernel (build with SWN_DEFS=-DSWN_EXP_TIMING ...): s_memtime deltas of wave 0 of the first
4096 workgroups: K loop, barrier after it, epilogue, second barrier, write-out; s_memtime ticks at 100 MHz (10 ns)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from switch_nerf_amd import ops as o
dev = torch.device('cuda'); dt = torch.bfloat16
E, M, CAP, NSEG = 8, 256, 16384, 16
NG = NSEG * E; ROWS = NG * CAP
mode = sys.argv[1] if len(sys.argv) > 1 else "bare"
h0 = torch.randn(ROWS, M, device=dev).to(dt)
W = [o.pack_weights(torch.randn(E, M, M, device=dev).mul_(1 / 16), dt, True) for _ in range(7)]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(7)]
y = torch.empty(ROWS, M, dtype=dt, device=dev)
saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(7)]
nw = o.chain_mask_words(dt, NG, CAP)
masks = [torch.empty(nw, dtype=torch.int32, device=dev) for _ in range(7)]
if mode == "bare":
    layers = [o.Layer(W[l], None) for l in range(7)]
else:
    layers = [o.Layer(W[l], B[l], relu=1 if l < 6 else 0, save=saves[l] if l < 6 else None, mask=masks[l] if l < 6 else None) for l in range(7)]
dbg = torch.zeros(4096 * 16, dtype=torch.int32, device=dev)       # 4096 x 8 int64
counts = torch.full((NG,), CAP, dtype=torch.int32, device=dev)
for _ in range(3):
    o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, y_add_gather=dbg, tag=1)
torch.cuda.synchronize()
t = dbg.view(torch.int64).view(4096, 8).cpu().double()
names = ["k_loop", "barrier1", "epilogue", "barrier2", "writeout", "total", "start"]
tot = t[:, 5].mean().item()
print(f"mode {mode}: per workgroup (7 layers), s_memtime ticks; mean over 4096 workgroups")
for i, n in enumerate(names[:6]):
    print(f"  {n:10s} {t[:, i].mean().item():10.1f}  ({100 * t[:, i].mean().item() / tot:5.1f} %)   per layer {t[:, i].mean().item() / 7:8.1f}")
print("  start spread (ticks):", (t[:, 6].max() - t[:, 6].min()).item())

"""Phase timing of the 8-wave 512-feature chain (chain.hip -DSWN_WIDE=2).  Build:
  SWN_VARIANT=timingw SWN_DEFS=-DSWN_EXP_TIMING SWN_ONLY=bf16 bash switch_nerf_amd/build.sh
  SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_timingw.so python scripts/experiments/chain_wide_timing.py [fwd|bwd|bare]
s_memtime deltas of wave 0 of the first 4096 workgroups: K loop, barrier behind it, epilogue, second barrier, write-out (100 MHz ticks:
10 ns), Mission Bay's per-GPU share: 16 experts x 512 features, groups of 13312 rows (one 212992-point segment), every group full."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from switch_nerf_amd import ops as o
dev = torch.device('cuda'); dt = torch.bfloat16
E, M, CAP, NSEG = 16, 512, 13312, 4
NG = NSEG * E; ROWS = NG * CAP
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
h0 = torch.randn(ROWS, M, device=dev).to(dt)
L = 7
W = [o.pack_weights(torch.randn(E, M, M, device=dev).mul_(1 / 22), dt, mode != "bwd") for _ in range(L)]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]
y = torch.empty(ROWS, M, dtype=dt, device=dev)
saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L)]
nw = o.chain_mask_words(dt, NG, CAP, M)
masks = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(L)]
if mode == "bare":
    layers = [o.Layer(W[l], None) for l in range(L)]
elif mode == "fwd":
    layers = [o.Layer(W[l], B[l], relu=1 if l < L - 1 else 0, save=saves[l] if l < L - 1 else None, mask=masks[l] if l < L - 1 else None) for l in range(L)]
else:
    layers = [o.Layer(W[l], None, relu=2 if l < L - 1 else 0, mask=masks[l] if l < L - 1 else None, save=saves[l] if l < L - 1 else None) for l in range(L)]
dbg = torch.zeros(4096 * 16, dtype=torch.int32, device=dev)       # 4096 x 8 int64
counts = torch.full((NG,), CAP, dtype=torch.int32, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    if it == 3:
        ev[0].record()
    o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, y_add_gather=dbg, tag=1)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1])
flop = 2.0 * ROWS * M * M * L
print(f"mode {mode}: {ROWS} rows x {L} layers of {M}: {ms:.3f} ms = {flop / ms / 1e9:.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f} of the MFMA peak (timing build)")
t = dbg.view(torch.int64).view(4096, 8).cpu().double()
names = ["k_loop", "barrier1", "epilogue", "barrier2", "writeout", "total"]
tot = t[:, 5].mean().item()
print(f"per workgroup ({L} layers, 128-row tile), s_memtime ticks of 10 ns; mean over 4096 workgroups; total {tot:.0f} ticks")
for i, n in enumerate(names):
    m_ = t[:, i].mean().item()
    print(f"  {n:10s} {m_:10.1f}  ({100 * m_ / tot:5.1f} %)   per layer {m_ / L:8.1f} ticks = {m_ / L * 10:7.0f} ns")
print(f"  unaccounted (prologue: ring preload + staging; end) {tot - sum(t[:, i].mean().item() for i in range(5)):.0f} ticks")
print("  start spread (ticks):", (t[:, 6].max() - t[:, 6].min()).item())

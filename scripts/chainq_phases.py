"""Phase timers of the persistent expert chain (geometry 7; library variant built with SWN_DEFS=-DSWN_BIG_TIMING):
   SWN_VARIANT=timing SWN_DEFS=-DSWN_BIG_TIMING bash switch_nerf_amd/build.sh
   SWN_LIB=switch_nerf_amd/libswn_hip_timing.so python scripts/chainq_phases.py [full|nosave|bwd]
mean shader clocks of each of the 8 waves per tile and per layer."""
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
M, E, L, CAP, NSEG = 256, 8, 7, 16384, 16
NG = NSEG * E
ROWS = NG * CAP
torch.manual_seed(0)
h0 = torch.randn(ROWS, M, device=dev).to(dt)
perm = torch.randperm(ROWS, device=dev).int()
Wm = [torch.randn(E, M, M, device=dev).mul_(1 / 16) for _ in range(L)]
Wf = [o.pack_weights(w, dt, True) for w in Wm]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]
saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
masks = [torch.zeros(o.chain_mask_words(dt, NG, CAP, M), dtype=torch.int32, device=dev) for _ in range(L - 1)]
y = torch.empty(ROWS, M, dtype=dt, device=dev)
counts = torch.full((NG,), CAP, dtype=torch.int32, device=dev)
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
for mode in sys.argv[1:] or ["full", "nosave", "bwd"]:
    save = mode in ("full", "bwd")
    if mode == "bwd":
        layers = [o.Layer(Wf[l], None, relu=2 if l < L - 1 else 0, mask=masks[l] if l < L - 1 else None, save=saves[l] if l < L - 1 else None) for l in range(L)]
    else:
        layers = [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if (save and l < L - 1) else None,
                          mask=masks[l] if (save and l < L - 1) else None) for l in range(L)]
    def f():
        o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm,
                    y_add_gather=dbg.view(torch.int32), tag=1, geometry=7)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    for nm, sl in [(f"wave {w_} (row group {w_ >> 2})", slice(512 * w_, 512 * w_ + 256)) for w_ in range(8)]:
        raw = dbg.view(4096, 8)[sl]
        t = raw.double().mean(0).tolist()
        tiles = (raw[:, 6] & 0xFFFF).double().mean().item()
        t_si = (raw[:, 6] >> 16).double().mean().item()
        clk = t[7] / (ms * 1e-3) / 1e9
        print(f"{mode} {nm}: {ms:.3f} ms, {tiles:.1f} tiles per workgroup, implied clock {clk:.2f} GHz; per tile: S {t[0] / tiles:.0f} (write-out issued at {t[1] / tiles:.0f}, staging + claim issued at {t_si / tiles:.0f}); "
              f"per layer: K {t[2] / tiles / L:.0f} + barrier {t[3] / tiles / L:.0f}, E {t[4] / tiles / L:.0f} + barriers (E and S) {t[5] / tiles / L:.0f}; "
              f"total per tile {t[7] / tiles:.0f}")

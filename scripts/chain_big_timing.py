"""Phase timers of the 256-row chain (build with SWN_DEFS=-DSWN_BIG_TIMING): mean shader clocks per workgroup of wave 0.
python scripts/chain_big_timing.py [full|nosave|bare|bwd]"""
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
GEOM = int(__import__("os").environ.get("GEOM", "2"))
for mode in sys.argv[1:] or ["full", "nosave", "bare", "bwd"]:
    save, bare = mode in ("full", "bwd"), mode == "bare"
    if mode == "bwd":
        layers = [o.Layer(Wf[l], None, relu=2 if l < L - 1 else 0, mask=masks[l] if l < L - 1 else None, save=saves[l] if l < L - 1 else None) for l in range(L)]
    else:
        layers = [o.Layer(Wf[l], None if bare else B[l], relu=0 if bare else (1 if l < L - 1 else 0), skip=(l == 3 and not bare),
                          save=saves[l] if (save and l < L - 1) else None, mask=masks[l] if (save and l < L - 1) else None) for l in range(L)]
    def f():
        o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm,
                    y_add_gather=dbg.view(torch.int32), tag=1, geometry=GEOM)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    if GEOM == 4:
        wgs_per_cu = (ROWS // 256) / 256.0
        for nm, sl in (("wave 0 (row group 0)", slice(0, 2048)), ("wave 4 (row group 1)", slice(2048, 4096))):
            t = dbg.view(4096, 8)[sl].double().mean(0).tolist()
            clk = t[7] * wgs_per_cu / (ms * 1e-3) / 1e9
            print(f"{mode} {nm}: {ms:.3f} ms; implied clock {clk:.2f} GHz; per layer: K phase {t[0] / L:.0f}, wait+barrier after K {t[1] / L:.0f}, "
                  f"E phase {t[2] / L:.0f} (of it bias/mask/write-out issue {t[4] / L:.0f}), barrier after E {t[3] / L:.0f}; prologue {t[5]:.0f}, tail {t[6]:.0f}, total {t[7]:.0f}")
        continue
    t = dbg.view(4096, 8).double().mean(0).tolist()
    names = ["K wait", "K barrier", "K loop", "epilogue", "post-K barrier/restage", "prologue", "write-out", "total"]
    wgs_per_cu = (ROWS // 256) / 256.0
    clk = t[7] * wgs_per_cu / (ms * 1e-3) / 1e9
    print(f"{mode}: {ms:.3f} ms; implied shader clock {clk:.2f} GHz (total ticks per workgroup x {wgs_per_cu:.0f} workgroups per CU / wall)")
    print("   " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, t)))
    print(f"   per layer: K loop {t[2] / L:.0f} (wait {t[0] / L:.0f}, barrier {t[1] / L:.0f}), epilogue {t[3] / L:.0f}, mid {t[4] / L:.0f}")

"""256-row chain geometry (chain_big.hip) vs the 64-row kernels (chain.hip): bit-exact comparison on ragged groups, then timing at
BASELINE sizes.  python scripts/chain_big_check.py [check|time|all]"""
import sys, json
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o

dev = torch.device('cuda')
dt = torch.bfloat16
M, E, L = 256, 8, 7
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
GEOMS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 3, 4)      # the first one is the reference


def build(ng, cap, counts, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    rows = ng * cap
    P = rows + 1000
    h0 = torch.randn(P, M, generator=g).to(dev).to(dt)
    perm = torch.full((rows,), -1, dtype=torch.int32)
    src = torch.randperm(P, generator=g)[:rows].int()
    for gi in range(ng):
        c = int(counts[gi])
        perm[gi * cap: gi * cap + c] = src[gi * cap: gi * cap + c]
    Wm = [torch.randn(E, M, M, generator=g).mul_(1 / 16).to(dev) for _ in range(L)]
    B = [torch.randn(E, M, generator=g).mul_(0.1).to(dev) for _ in range(L)]
    return h0, perm.to(dev), Wm, B


def run_fwd(geom, h0, perm, counts_t, Wf, B, ng, cap, save=True):
    rows = ng * cap
    saves = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    nw = o.chain_mask_words(dt, ng, cap, M)
    masks = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(L - 1)]
    y = torch.zeros(rows, M, dtype=dt, device=dev)
    layers = [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if (save and l < L - 1) else None,
                      mask=masks[l] if (save and l < L - 1) else None) for l in range(L)]
    o.mlp_chain(h0, layers, y, n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts_t, group_rows_clamp=cap,
                x_gather=perm, tag=1, geometry=geom)
    return y, saves, masks


def run_bwd(geom, dout, perm, counts_t, Wb, masks, skip_add, ng, cap):
    rows = ng * cap
    dz = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    dx = torch.zeros(rows, M, dtype=dt, device=dev)
    bl = []
    for i in range(L):
        l = L - 1 - i
        bl.append(o.Layer(Wb[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None))
    o.mlp_chain(dout, bl, dx, n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts_t, group_rows_clamp=cap, x_gather=perm,
                y_add=skip_add, tag=2, geometry=geom)
    return dx, dz


def valid_rows(ng, cap, counts):
    m = torch.zeros(ng * cap, dtype=torch.bool)
    for gi in range(ng):
        m[gi * cap: gi * cap + int(counts[gi])] = True
    return m.to(dev)


def check():
    ok = True
    for (ng, cap, seed) in ((16, 1000, 1), (8, 256, 2), (24, 700, 3), (8, 16384, 4)):
        g = torch.Generator().manual_seed(seed)
        counts = torch.randint(0, cap + 1, (ng,), generator=g)
        counts[0] = cap
        counts[1] = 0
        if ng > 2:
            counts[2] = 1
        if ng > 3:
            counts[3] = min(cap, 257)
        counts_t = counts.int().to(dev)
        h0, perm, Wm, B = build(ng, cap, counts, seed)
        Wf = [o.pack_weights(w, dt, True) for w in Wm]
        Wb = [o.pack_weights(w, dt, False) for w in Wm]
        vm = valid_rows(ng, cap, counts)
        res = {}
        for geom in GEOMS:
            y, saves, masks = run_fwd(geom, h0, perm, counts_t, Wf, B, ng, cap)
            dout = (torch.randn(h0.shape[0], M, generator=torch.Generator().manual_seed(seed + 7)).to(dev) * 0.1).to(dt)
            skip_add = torch.randn(ng * cap, M, generator=torch.Generator().manual_seed(seed + 9)).to(dev).to(dt)
            dx, dz = run_bwd(geom, dout, perm, counts_t, Wb, masks, skip_add, ng, cap)
            torch.cuda.synchronize()
            res[geom] = dict(y=y, saves=saves, dx=dx, dz=dz)
        def cmp(name, a, b):
            nonlocal ok
            a, b = a[vm], b[vm]
            same = torch.equal(a, b)
            if not same:
                d = (a.float() - b.float()).abs()
                bad = (a != b).float().mean().item()
                print(f"  MISMATCH {name}: max abs {d.max().item():.4g}, frac differing {bad:.3g}, nan {torch.isnan(a.float()).any().item()} {torch.isnan(b.float()).any().item()}")
                ok = False
            return same
        print(f"groups {ng} cap {cap} counts {counts.tolist()[:8]}...")
        allsame = True
        for gg in GEOMS[1:]:
            allsame &= cmp(f"g{gg} y", res[1]["y"], res[gg]["y"])
            for l in range(L - 1):
                allsame &= cmp(f"g{gg} save{l}", res[1]["saves"][l], res[gg]["saves"][l])
            allsame &= cmp(f"g{gg} dx", res[1]["dx"], res[gg]["dx"])
            for l in range(L - 1):
                allsame &= cmp(f"g{gg} dz{l}", res[1]["dz"][l], res[gg]["dz"][l])
            # rows past the valid ones must be untouched (zeros)
            for nm, t in (("y", res[gg]["y"]), ("save0", res[gg]["saves"][0]), ("dx", res[gg]["dx"]), ("dz0", res[gg]["dz"][0])):
                if t[~vm].abs().sum().item() != 0:
                    print(f"  WROTE PAST THE VALID ROWS: g{gg} {nm}")
                    ok = False
        # inference variant (no saves / masks)
        y1, _, _ = run_fwd(1, h0, perm, counts_t, Wf, B, ng, cap, save=False)
        for gg in GEOMS[1:]:
            y2, _, _ = run_fwd(gg, h0, perm, counts_t, Wf, B, ng, cap, save=False)
            allsame &= cmp(f"g{gg} y (no saves)", y1, y2)
        print("  bit-exact" if allsame else "  DIFFERENT")
    print("CHECK", "OK" if ok else "FAILED")
    return ok


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def time_all():
    CAP, NSEG = 16384, 16
    NG = NSEG * E
    ROWS = NG * CAP
    torch.manual_seed(0)
    h0 = torch.randn(ROWS, M, device=dev).to(dt)
    perm = torch.randperm(ROWS, device=dev).int()
    Wm = [torch.randn(E, M, M, device=dev).mul_(1 / 16) for _ in range(L)]
    Wf = [o.pack_weights(w, dt, True) for w in Wm]
    Wb = [o.pack_weights(w, dt, False) for w in Wm]
    B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]
    saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
    nw = o.chain_mask_words(dt, NG, CAP, M)
    masks = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(L - 1)]
    y = torch.empty(ROWS, M, dtype=dt, device=dev)
    out = {}
    for frac in (1.0, 0.8):
        counts = torch.full((NG,), CAP, dtype=torch.int32, device=dev)
        if frac < 1.0:      # unbalanced like the random-init router: half of the experts full, the rest share what is left
            c = torch.full((NG,), CAP, dtype=torch.int32)
            c[1::2] = int(CAP * (2 * frac - 1))
            counts = c.to(dev)
        kept = int(counts.sum().item())
        for geom in GEOMS:
            def fwd(save=True, bare=False):
                layers = [o.Layer(Wf[l], None if bare else B[l], relu=0 if bare else (1 if l < L - 1 else 0), skip=(l == 3 and not bare),
                                  save=saves[l] if (save and l < L - 1) else None, mask=masks[l] if (save and l < L - 1 and not bare) else None)
                          for l in range(L)]
                o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm,
                            tag=1, geometry=geom)
            def bwd():
                bl = []
                for i in range(L):
                    l = L - 1 - i
                    bl.append(o.Layer(Wb[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=saves[l - 1] if l > 0 else None))
                o.mlp_chain(h0, bl, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm,
                            y_add=saves[3], tag=2, geometry=geom)
            for name, f in (("fwd full", lambda: fwd()), ("fwd no saves", lambda: fwd(save=False)), ("fwd bare", lambda: fwd(False, True)),
                            ("bwd full", bwd)):
                ms = timeit(f)
                tf = 2.0 * L * M * M * kept / (ms * 1e-3) / 1e12
                print(f"kept {frac:.1f} geometry {geom} {name:14s} {ms:7.3f} ms  {tf:7.1f} TF  mfma_frac {tf / 2500:.3f}", flush=True)
                out[f"{frac}/{geom}/{name}"] = dict(ms=round(ms, 3), tflops=round(tf, 1))
    json.dump(out, open('gpurun_out/chain_big_time.json', 'w'), indent=1)


if mode in ("check", "all"):
    good = check()
if mode in ("time", "all"):
    time_all()

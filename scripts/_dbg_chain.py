import sys, torch
sys.path.insert(0, '.')
sys.argv = ['x', 'none']
exec(open('scripts/chain_big_check.py').read().split("if mode in")[0])
ng, cap, seed = 8, 16384, 4
g = torch.Generator().manual_seed(seed)
counts = torch.randint(0, cap + 1, (ng,), generator=g); counts[0] = cap; counts[1] = 0; counts[2] = 1; counts[3] = 257
counts_t = counts.int().to(dev)
h0, perm, Wm, B = build(ng, cap, counts, seed)
Wf = [o.pack_weights(w, dt, True) for w in Wm]; Wb = [o.pack_weights(w, dt, False) for w in Wm]
vm = valid_rows(ng, cap, counts)
dout = (torch.randn(h0.shape[0], M, generator=torch.Generator().manual_seed(seed + 7)).to(dev) * 0.1).to(dt)
skip_add = torch.randn(ng * cap, M, generator=torch.Generator().manual_seed(seed + 9)).to(dev).to(dt)
R = {}
for geom in (1, 2, 2):
    y, saves, masks = run_fwd(geom, h0, perm, counts_t, Wf, B, ng, cap)
    dx, dz = run_bwd(geom, dout, perm, counts_t, Wb, masks, skip_add, ng, cap)
    torch.cuda.synchronize()
    R.setdefault(geom, []).append((y, saves, dx, dz))
a, b, c = R[1][0], R[2][0], R[2][1]
print("geom2 run-to-run identical:", all(torch.equal(x, y) for x, y in zip(b[3], c[3])), torch.equal(b[2], c[2]))
for l in range(L - 2, -1, -1):
    d1, d2 = a[3][l], b[3][l]
    neq = ((d1 != d2) & vm[:, None]).nonzero()
    print("dz", l, "mismatches", neq.shape[0])
    for (r, cc) in neq[:5].tolist():
        print("   row", r, "col", cc, "old", d1[r, cc].item(), "new", d2[r, cc].item(), "saved act (fwd) old/new", a[1][l][r, cc].item(), b[1][l][r, cc].item())

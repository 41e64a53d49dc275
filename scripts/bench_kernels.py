"""Micro-benchmarks of the MFMA kernels (chain / wgrad) at BASELINE sizes; prints ms and TFLOP/s per variant."""
import sys, os, time, json
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o

dev = torch.device('cuda')
dt = torch.bfloat16
E, M, CAP, NSEG = 8, 256, 16384, 16
NG = NSEG * E
ROWS = NG * CAP
torch.manual_seed(0)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

h0 = torch.randn(ROWS, M, device=dev).to(dt)
perm = torch.randperm(ROWS, device=dev).int()
counts = torch.full((NG,), CAP, dtype=torch.int32, device=dev)
W = [o.pack_weights(torch.randn(E, M, M, device=dev).mul_(1 / 16), dt, True) for _ in range(8)]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(8)]
saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(8)]
nw = o.chain_mask_words(dt, NG, CAP)
masks = [torch.empty(nw, dtype=torch.int32, device=dev) for _ in range(8)]
xs = torch.empty(ROWS, M, dtype=dt, device=dev)
y = torch.empty(ROWS, M, dtype=dt, device=dev)
res = {}

def run(name, L, save=True, mask=True, gather=True, skip=True, bias=True, relu=True, xsave=True, tag=0):
    layers = [o.Layer(W[l], B[l] if bias else None, relu=(1 if (relu and l < L - 1) else 0), skip=(skip and l == 3 and L > 3),
                      save=saves[l] if (save and l < L - 1) else None, mask=masks[l] if (mask and relu and l < L - 1) else None)
              for l in range(L)]
    f = lambda: o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP,
                            x_gather=perm if gather else None, x_save=xs if xsave else None, tag=tag)
    ms = timeit(f)
    tf = 2.0 * L * M * M * ROWS / (ms * 1e-3) / 1e12
    res[name] = dict(ms=round(ms, 3), tflops=round(tf, 1))
    print(f"{name:40s} {ms:8.3f} ms  {tf:7.1f} TF", flush=True)

run("expert_fwd full L=7", 7)
run("L=7 no saves", 7, save=False, xsave=False)
run("L=7 no saves no masks", 7, save=False, mask=False, xsave=False)
run("L=7 no saves/masks/gather", 7, save=False, mask=False, gather=False, xsave=False)
run("L=7 bare (no bias/relu/skip)", 7, save=False, mask=False, gather=False, skip=False, bias=False, relu=False, xsave=False)
run("L=1 bare", 1, save=False, mask=False, gather=False, skip=False, bias=False, relu=False, xsave=False)
run("L=2 bare", 2, save=False, mask=False, gather=False, skip=False, bias=False, relu=False, xsave=False)
run("L=4 bare", 4, save=False, mask=False, gather=False, skip=False, bias=False, relu=False, xsave=False)
run("L=8 bare", 8, save=False, mask=False, gather=False, skip=False, bias=False, relu=False, xsave=False)
run("L=1 gather+xsave", 1, save=False, mask=False, gather=True, skip=False, bias=False, relu=False, xsave=True)

# wgrad
a = torch.randn(ROWS, M, device=dev).to(dt); b = torch.randn(ROWS, M, device=dev).to(dt)
dw = torch.zeros(E, M, M, device=dev); db = torch.zeros(E, M, device=dev)
for ns, ws in ((4, True), (4, False), (8, True), (2, True)):
    ms = timeit(lambda: o.wgrad(a, b, dw, db, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, n_splits=ns, tag=1, use_workspace=ws))
    gbs = ROWS * M * 2 * 2 / (ms * 1e-3) / 1e9
    print(f"wgrad expert splits={ns} ws={ws}: {ms:.3f} ms  {2.0*M*M*ROWS/(ms*1e-3)/1e12:.1f} TF  {gbs:.0f} GB/s", flush=True)
dw1 = torch.zeros(1, M, M, device=dev); db1 = torch.zeros(1, M, device=dev)
for ns, ws in ((256, True), (256, False), (512, True), (1024, True)):
    ms = timeit(lambda: o.wgrad(a, b, dw1, db1, n_splits=ns, use_workspace=ws))
    gbs = ROWS * M * 2 * 2 / (ms * 1e-3) / 1e9
    print(f"wgrad dense splits={ns} ws={ws}: {ms:.3f} ms  {2.0*M*M*ROWS/(ms*1e-3)/1e12:.1f} TF  {gbs:.0f} GB/s", flush=True)
json.dump(res, open('gpurun_out/bench_kernels.json', 'w'), indent=1)

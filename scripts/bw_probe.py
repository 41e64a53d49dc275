import torch, time
dev='cuda'
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for gb in (1,4,8):
    n=gb*(1<<30)//2
    x=torch.empty(n,dtype=torch.bfloat16,device=dev); y=torch.empty_like(x)
    ms=t(lambda: x.fill_(1.0)); print(f"fill {gb} GB: {ms:.3f} ms  {gb*1.0737/ms*1e3:.0f} GB/s write")
    ms=t(lambda: y.copy_(x)); print(f"copy {gb} GB: {ms:.3f} ms  {2*gb*1.0737/ms*1e3:.0f} GB/s r+w")
    ms=t(lambda: x.sum()); print(f"sum  {gb} GB: {ms:.3f} ms  {gb*1.0737/ms*1e3:.0f} GB/s read")

import sys, time, torch
sys.path.insert(0, '.')
from switch_nerf_amd.model import SwitchNeRF, BUILDING
from switch_nerf_amd.graph import GraphedTrainStep
from bench import synth_batch
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, chunk = 256, 131072
model = SwitchNeRF(BUILDING, dtype=torch.bfloat16, device=dev, seed=0)
rays, idx, rgbs = synth_batch(N, 1000, dev)
P = N * S
def step():
    pr = torch.rand(N, S, device=dev)
    noise = torch.randn(P, device=dev)
    return model.train_step(rgbs, rays, idx, S, min(chunk, P), perturb=1.0, perturb_rand=pr, sigma_noise=noise)
for _ in range(3):
    st = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    st = step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 30 * 1e3, "loss", st["loss"].item())
gs = GraphedTrainStep(model, rgbs, rays, idx, S, chunk)
for _ in range(3):
    st = gs()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    st = gs()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 30 * 1e3, "loss", st["loss"].item(), "steps", model.step_count)

"""Does dropping a GraphedTrainStep whose pool holds tensors with a recorded stream crash?  (the intermittent SIGSEGV of `python bench.py`
in round 6's evidence collection.)  python scripts/experiments/graph_drop_stress.py [cycles]; SWN_EXP_RECORD_STREAM=1 puts
Tensor.record_stream back into SwitchNeRF._join_side_outputs."""
import faulthandler, gc, os, sys
faulthandler.enable()
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from switch_nerf_amd.graph import GraphedTrainStep
from switch_nerf_amd.model import SwitchNeRF
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
m = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16, device=dev)
rays, img, rgbs = (torch.from_numpy(v).to(dev) for v in synth.make_rays(5, 256))
for it in range(cycles):
    step = GraphedTrainStep(m, rgbs, rays, img, 64, 4096)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    step = None
    gc.collect()
    x = [torch.empty(1 << 20, device=dev) for _ in range(8)]      # allocator traffic behind the drop (deferred events are processed on malloc)
    del x
    torch.cuda.synchronize()
print(f"GRAPH_DROP_STRESS ok: {cycles} cycles, record_stream {'ON' if os.environ.get('SWN_EXP_RECORD_STREAM') == '1' else 'off'}", flush=True)

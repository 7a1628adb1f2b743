"""Which torch operators (not the engine's kernels) a training step launches: torch.profiler over one eager step (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cnmnet_amd.trainer import TrainStepWoNormal, synthetic_training_sample
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
step = TrainStepWoNormal(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev))
s = {k: v.to(dev) for k, v in synthetic_training_sample(4, 192, 256, seed=1).items()}
a = (s["rgbs"], s["cameras"], s["disparities"], s["depths"])
for _ in range(3):
    step(*a)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(*a)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = 0.0
for e in rows[:40]:
    tot += e.device_time_total
    print("%8.1f us  x%-4d %-28s %s" % (e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
print("aten total (top 40): %.2f ms" % (tot / 1e3))

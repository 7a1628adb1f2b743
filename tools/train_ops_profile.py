"""Which aten ops does one training step launch, and from where?  (torch.profiler over two steps, GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.trainer import TrainStepWoNormal, synthetic_training_sample
dev = torch.device("cuda:0"); torch.manual_seed(0)
step = TrainStepWoNormal(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev))
s = {k: v.to(dev) for k, v in synthetic_training_sample(4, 192, 256, seed=1).items()}
args = (s["rgbs"], s["cameras"], s["disparities"], s["depths"])
for _ in range(6): step(*args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    for _ in range(2): step(*args)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=4)
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.stack) for e in ka if e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
for k, c, t, st in rows[:40]:
    src = [l for l in st if "cnmnet_amd" in l or "trainer" in l][:2]
    print("%-28s x%4d  %8.1f us/step  %s" % (k, c // 2, t / 2, " <- ".join(x.split("/")[-1][:70] for x in src)))

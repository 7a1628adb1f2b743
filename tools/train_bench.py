"""Throughput of the training step (BASELINE config 3 shape per GPU: batch 4, 192x256, 64 planes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.trainer import TrainStepWoNormal, synthetic_training_sample
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
GRAPH = len(sys.argv) > 2 and sys.argv[2] == "graph"
step = TrainStepWoNormal(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev), graph=GRAPH)
s = {k: v.to(dev) for k, v in synthetic_training_sample(B, 192, 256, seed=1).items()}
for _ in range(2):
    log = step(s["rgbs"], s["cameras"], s["disparities"], s["depths"])
torch.cuda.synchronize(); t = time.perf_counter()
n = 5
for _ in range(n):
    log = step(s["rgbs"], s["cameras"], s["disparities"], s["depths"])
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("train step%s B=%d: %.1f ms -> %.1f samples/s  (loss %.4f, peak mem %.2f GB)" % (" (HIP graph)" if GRAPH else "", B, dt * 1e3, B / dt, log["loss"], torch.cuda.max_memory_allocated() / 2**30))

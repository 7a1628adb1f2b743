"""One process, full-size training step: the two-graph (segmented) replay of TrainStep against the one-graph replay and the eager step.
The exchange is replaced by a no-op reducer, so what is timed is the cut itself.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd.trainer import TrainStepWoNormal, synthetic_training_sample
from cnmnet_amd.depthnet import depthNet, DepthRefineNet

class NoExchange:
    def launch_ready(self, ready): pass
    def reduce_all(self): pass
    def finish(self): pass

dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = {k: v.to(dev) for k, v in synthetic_training_sample(B, 192, 256, seed=3).items()}
args = (sd["rgbs"], sd["cameras"], sd["disparities"], sd["depths"])
for name, graph, seg in (("eager", False, False), ("one graph", True, False), ("two graphs", True, True)):
    torch.manual_seed(0)
    step = TrainStepWoNormal(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev), lr=1e-4, graph=graph)
    if seg:
        step.reducer = NoExchange()
    for _ in range(3): log = step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): log = step(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-10s B=%d: %.1f ms per step (loss %.4f, peak mem %.2f GB)" % (name, B, dt * 1e3, log["loss"], torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del step; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()

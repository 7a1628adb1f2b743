"""Throughput of the training step (BASELINE config 3 shape per GPU: batch 4, 192x256, 64 planes).
python tools/train_bench.py [B] [graph] [normals]    graph: replay the step as one HIP graph; normals: the `train` step
(Depth2normal k = 9 losses + warped-depth losses) instead of `train_wo_normal`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.trainer import TrainStep, TrainStepWoNormal, synthetic_training_sample
dev = torch.device("cuda:0")
if os.environ.get("CNM_WGRAD_STREAMK"):                   # A/B: 0 = the split form of the weight-gradient GEMMs
    from cnmnet_amd import _lib
    _lib.load().cnm_tune_wgrad_streamk(int(os.environ["CNM_WGRAD_STREAMK"]))
if os.environ.get("CNM_WGRAD_LINEAR"):
    from cnmnet_amd import _lib
    _lib.load().cnm_tune_wgrad_linear(int(os.environ["CNM_WGRAD_LINEAR"]))
if os.environ.get("CNM_WGRAD_SK_SHARE"):
    from cnmnet_amd import _lib
    _lib.load().cnm_tune_wgrad_streamk_share(int(os.environ["CNM_WGRAD_SK_SHARE"]))
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
GRAPH, NORMALS = "graph" in sys.argv[2:], "normals" in sys.argv[2:]
step = (TrainStep if NORMALS else TrainStepWoNormal)(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev), graph=GRAPH)
s = {k: v.to(dev) for k, v in synthetic_training_sample(B, 192, 256, seed=1).items()}
args = (s["rgbs"], s["cameras"], s["disparities"], s["depths"]) + ((s["normals"],) if NORMALS else ())
for _ in range(8):                                       # the first iterations grow the caching allocator's pools: 70 ms instead of 56 ms per step
    log = step(*args)
torch.cuda.synchronize(); t = time.perf_counter()
n = 10
for _ in range(n):
    log = step(*args)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("%s step%s B=%d: %.1f ms -> %.1f samples/s  (loss %.4f, peak mem %.2f GB)" % ("train (normals, k=9)" if NORMALS else "train_wo_normal", " (HIP graph)" if GRAPH else "", B, dt * 1e3, B / dt, log["loss"], torch.cuda.max_memory_allocated() / 2**30))

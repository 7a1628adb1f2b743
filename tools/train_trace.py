"""Loss of the first training steps, eager against HIP graph, one pass over both sources against two depthNet calls (GPU box).
python tools/train_trace.py [B] [normals]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import trainer
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.trainer import TrainStep, TrainStepWoNormal, synthetic_training_sample
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NORMALS = "normals" in sys.argv[2:]
s = {k: v.to(dev) for k, v in synthetic_training_sample(B, 192, 256, seed=1).items()}
args = (s["rgbs"], s["cameras"], s["disparities"], s["depths"]) + ((s["normals"],) if NORMALS else ())
for one_pass in (True, False):
    for graph in (False, True):
        trainer.SOURCES_IN_ONE_PASS = one_pass
        torch.manual_seed(0)
        step = (TrainStep if NORMALS else TrainStepWoNormal)(depthNet(3.0).to(dev), DepthRefineNet(32, 3.0).to(dev), graph=graph)
        losses = [float(step(*args)["loss"]) for _ in range(14)]
        print("one pass %-5s graph %-5s:" % (one_pass, graph), " ".join("%.4g" % l for l in losses), flush=True)

import os, time, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
import torch
import numpy as np
from oracle import ref_arrangement as ra
from cnmnet_amd import synthetic as syn
print("default threads", torch.get_num_threads())
img, cams = syn.frames(1, 2, 192, 256, seed=1)
T = torch.from_numpy
dn, rn = ra.DepthNetCPU(3.0).eval(), ra.DepthRefineNetCPU(32, 3.0).eval()
args = (T(img[:, 0]), T(img[:, 1]), T(img[:, 2]), T(cams[:, 0]), T(cams[:, 1]), T(cams[:, 2]))
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    ra.frame_forward(dn, rn, *args)
    t = time.perf_counter(); ra.frame_forward(dn, rn, *args); dt = time.perf_counter() - t
    print(n, "threads: %.2f s/frame" % dt, flush=True)
    if dt > 20: break

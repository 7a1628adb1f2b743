import sys, torch
sys.path.insert(0, "/root/repo")
from cnmnet_amd import ops
torch.manual_seed(0)
for (co, ci, k) in ((512, 512, 3), (512, 1024, 3), (128, 67, 3), (256, 128, 5), (64, 65, 3)):
    w = torch.randn(co, ci, k, k, device="cuda")
    fn = (lambda: ops.pack_winograd4(w, None, 3 if ci == 67 else 0))
    u = fn()
    # reference: G w G^T in float64 on the host for a few entries
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%dx%dx%d: %.1f us per pack, %.2f TB/s written, checksum %.9e" % (co, ci, k, e0.elapsed_time(e1) / 20 * 1e3, u.numel() * 4 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12, u.double().abs().sum().item()))

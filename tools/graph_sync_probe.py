"""HIP-graph replay of a torch full reduction whose scratch block is dirtied later in the same graph (GPU box only).
torch's split reductions clear their semaphores with cudaMemsetAsync (a memset NODE in a captured graph); the block is
recycled inside the graph's pool, so every replay depends on that node running.  Prints the replayed values before and
after a device-wide synchronise, for the one-launch reduction and for the two-stage form (rows of 256, then the row sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(4, 1, 192, 256, device=dev)
cases = {
    "one launch (split reduction + memset node)": lambda t: t.sum(),
    "two stages (no scratch)": lambda t: t.reshape(-1, 256).sum(1).sum(),
}
for name, fn in cases.items():
    def body():
        x.add_(1.0)                                                                       # every replay sums a different tensor: a stale result shows
        out = fn(x) + 0
        junk = [torch.full((128,), 7, dtype=torch.int32, device=dev) for _ in range(8)]   # recycles the freed scratch blocks
        return out, junk
    eager = float(body()[0]); base = float(x.double().sum())
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, junk = body()
    del junk
    vals = []
    for i in range(6):
        if i == 3:
            torch.cuda.synchronize()
        g.replay()
        vals.append(float(out) - float(x.double().sum()))
    print("%-44s replayed sum - fp64 sum of the same tensor: %s   (device synchronise before the 4th)" % (name, ["%.6g" % v for v in vals]), flush=True)

# a bare memset node: hipMemsetAsync(b, 0) ; b += 1 ; out = b  -- every replay must return 1
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (4, 512, 4096, 1 << 20):
    b = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(b.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        b.add_(1)
        out = b.clone()
    vals = []
    for i in range(6):
        if i == 3:
            torch.cuda.synchronize()
        g.replay()
        vals.append((int(out.min()), int(out.max())))
    print("memset node of %7d bytes (rc %d), then += 1: replays (min, max) %s   (device synchronise before the 4th)" % (nbytes, rc, vals), flush=True)

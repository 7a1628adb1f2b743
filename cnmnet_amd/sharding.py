"""Multi-GPU sharding of the hot path: frames (and (ref,src) pairs) are independent
(SURVEY.md section 8e), so N GPUs = N processes each owning a contiguous shard, with NO
data-path collective.  The process group (RCCL on GPUs, gloo in CPU tests) is used only to
line ranks up and to agree on the job's wall time (max over ranks)."""
import torch


def shard_range(total, rank, world):
    """Contiguous, balanced split of `total` frames: the first total % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def job_elapsed(local_elapsed_s, dist=None, device="cpu"):
    """Wall time of the whole job = slowest rank (all-reduce MAX of one double)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_elapsed_s)
    t = torch.tensor([local_elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank, local_elapsed_s, dist=None, device="cpu"):
    """Whole-job units/s: sum over ranks of the units each processed / max-over-ranks time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_per_rank / local_elapsed_s
    n = torch.tensor([float(units_per_rank)], dtype=torch.float64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()) / job_elapsed(local_elapsed_s, dist, device)

"""Frame-batch scale-out: independent camera streams are sharded across ranks, one process
per GPU.  The time axis of a stream is strictly sequential (frame t needs frame t-1's image, tracks and map:
reference src/tracker.cpp:61-69,331), so nothing on the data path crosses ranks.  The only collective is one
broadcast of the intrinsics {K[9], d[5]} from rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests) plus
the max-reduce of the elapsed time that bench.py's contract asks for."""
from __future__ import annotations

import numpy as np


def shard_streams(n_streams: int, rank: int, world: int) -> list[int]:
    """Stream s runs on rank s % world (round-robin); returns this rank's global stream ids."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_streams, world))


def broadcast_intrinsics(K, d, dist=None, device="cpu", src: int = 0):
    """Rank `src` supplies K (3x3) and d (5); every rank returns identical float64 copies."""
    import torch
    buf = torch.zeros(14, dtype=torch.float64, device=device)
    if dist is None or dist.get_rank() == src:
        buf[:9] = torch.as_tensor(np.asarray(K, np.float64).reshape(9))
        if d is not None:
            buf[9:] = torch.as_tensor(np.asarray(d, np.float64).reshape(5))
    if dist is not None:
        dist.broadcast(buf, src=src)
    out = buf.cpu().numpy()
    return out[:9].reshape(3, 3).copy(), out[9:].copy()


def max_over_ranks(value: float, dist=None, device="cpu") -> float:
    """bench.py contract: the job's time is the slowest rank's."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_poses(rvec_tvec: np.ndarray, dist=None, device="cpu"):
    """Optional reporting path: all-gather of per-stream {rvec, tvec} (6 doubles each) so rank 0 can emit every
    trajectory.  rvec_tvec: (n_local, 6).  Returns a list of per-rank arrays (equal n_local on every rank)."""
    if dist is None:
        return [np.asarray(rvec_tvec, np.float64)]
    import torch
    t = torch.as_tensor(np.asarray(rvec_tvec, np.float64), device=device).contiguous()
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]

"""N > 1 path on CPU: world_size-2 gloo processes exercise the stream sharding, the intrinsics broadcast (the
path's only collective), the pose all-gather used for reporting and the max-over-ranks timing bench.py reports."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from ros2_mono_vo_amd import parallel


def test_shard_streams_partitions_everything_once():
    for n, w in ((8, 1), (8, 2), (8, 4), (8, 8), (5, 3), (2, 4)):
        got = sorted(sum((parallel.shard_streams(n, r, w) for r in range(w)), []))
        assert got == list(range(n))
        sizes = [len(parallel.shard_streams(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K0 = np.array([[1152.0, 0, 640], [0, 1152, 360], [0, 0, 1]])
        d0 = np.array([0.1, -0.2, 0.001, 0.002, 0.05])
        K, d = parallel.broadcast_intrinsics(K0 if rank == 0 else np.full((3, 3), -1.0), d0 if rank == 0 else None, dist)
        mine = parallel.shard_streams(8, rank, world)
        poses = np.array([[s, 0, 0, 0, 0, s * 10.0] for s in mine], np.float64)
        allp = parallel.gather_poses(poses, dist)
        tmax = parallel.max_over_ranks(1.0 + rank, dist)
        dist.barrier()
        q.put((rank, K.tolist(), d.tolist(), mine, [a.tolist() for a in allp], tmax))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_gather_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    K0 = [[1152.0, 0, 640], [0, 1152, 360], [0, 0, 1]]
    for rank, K, d, mine, allp, tmax in res:
        assert K == K0 and d == [0.1, -0.2, 0.001, 0.002, 0.05]      # every rank holds rank 0's intrinsics
        assert mine == list(range(rank, 8, world))
        assert tmax == 2.0                                             # slowest rank
        assert [row[0] for row in allp[0]] == [0, 2, 4, 6] and [row[0] for row in allp[1]] == [1, 3, 5, 7]

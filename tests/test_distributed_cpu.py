"""CPU, gloo, world sizes 2 and 8 (the node the driver scales to): pair sharding + the single pose gather of the multi-GPU path."""
import os
import socket

import numpy as np
import pytest

from relativepose_amd import distributed as D


def test_shard_range_partitions_exactly():
    for total in (1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = D.shard_range(total, r, world)
                got += list(range(lo, hi))
            assert got == list(range(total))


def _worker(rank, world, port, total, q):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    lo, hi = D.shard_range(total, r, w)
    pose = torch.arange(lo, hi, dtype=torch.float64)[:, None, None] + torch.eye(4, dtype=torch.float64)[None]
    status = torch.arange(lo, hi, dtype=torch.int32) % 5
    D.barrier(w)
    P, S = D.gather_poses(pose, status, total, w)
    t = D.max_over_ranks(float(rank + 1), w, torch.device("cpu"))
    q.put((rank, P.numpy(), S.numpy(), t))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_gather_poses_gloo_world2(total):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    want = np.arange(total)[:, None, None] + np.eye(4)[None]
    for rank, P, S, t in res:
        assert np.array_equal(P, want)
        assert np.array_equal(S, np.arange(total) % 5)
        assert t == 2.0


@pytest.mark.parametrize("total", [250, 256])
def test_gather_poses_gloo_world8_ragged_and_even(total):
    """Eight ranks like the 8-GPU node: 250 pairs = ragged shards of 32 / 31 (configs[3] with --total-pairs 250), 256 = even; every
    rank ends up with every pose in pair order, one all_gather per rank."""
    import torch.multiprocessing as mp
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    want = np.arange(total)[:, None, None] + np.eye(4)[None]
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, P, S, t in res:
        assert np.array_equal(P, want)
        assert np.array_equal(S, np.arange(total) % 5)
        assert t == float(world)

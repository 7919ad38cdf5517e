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


# ---- evaluation.evaluate_pairs_sharded: the sharded evaluation harness with a stub pipeline (no GPU) ----------------------------------
class _StubPipe:
    """prepare / run_pipelined of RelativePosePipeline for CPU tests: the 'pose' of a pair is the identity with the pair's tag (carried in
    rgb[b, 0, 0, 0, 0]) as translation x, status = tag % 3."""
    dataset = "suncg"

    def prepare(self, rgb, norm, depth, pts, ptw, device):
        return {"tags": rgb[:, 0, 0, 0, 0].copy()}

    alive = 0          # prepared states currently held by the pipeline
    max_alive = 0

    def run_pipelined(self, states, steps, on_result=None, depth=None, before_batch=None, provider=None):
        """The protocol of RelativePosePipeline.run_pipelined(provider=...): batch k is prepared right before it starts, `depth` batches are
        in flight, a batch's state is released when its result has been handed to on_result."""
        import torch
        cls = type(self)
        live, nxt, out = [], 0, [None] * steps
        while nxt < steps or live:
            while len(live) < max(1, depth or 1) and nxt < steps:
                live.append((nxt, provider(nxt) if provider is not None else states[nxt]))
                nxt += 1
                cls.alive += 1
                cls.max_alive = max(cls.max_alive, cls.alive)
            k, st = live.pop(0)
            t = torch.from_numpy(st["tags"]).to(torch.float64)
            pose = torch.eye(4, dtype=torch.float64).repeat(len(t), 1, 1)
            pose[:, 0, 3] = t
            status = t.to(torch.int32) % 3
            out[k] = on_result(k, pose, status) if on_result is not None else (pose, status)
            cls.alive -= 1
        return out


def _stub_batches(sizes):
    out, k = [], 0
    for B in sizes:
        rgb = np.zeros((B, 2, 3, 2, 8), np.float32)
        rgb[:, 0, 0, 0, 0] = np.arange(k, k + B)
        out.append({"rgb": rgb, "norm": rgb.copy(), "depth": np.zeros((B, 2, 2, 8), np.float32), "pts": np.zeros((B, 2, 4, 2)),
                    "ptw": np.ones((B, 2, 4)), "R": np.tile(np.eye(4), (B, 2, 1, 1))})
        k += B
    return out


def _stub_record(sub, idx, poses, k0):
    R = []
    for q, b in enumerate(idx):
        Rp = np.eye(4); Rp[:3, :4] = poses[q][:3, :4]
        R.append({"k": k0 + int(b), "tag": float(sub["rgb"][q, 0, 0, 0, 0]), "R_pred_44": Rp})
    return R


def _eval_worker(rank, world, port, sizes, path, q):
    import torch
    from relativepose_amd import evaluation as E
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    n0 = D.COLLECTIVES["all_gather"]
    stats = E.evaluate_pairs_sharded(_StubPipe(), _stub_batches(sizes), torch.device("cpu"), result_path=path, rank=r, world=w,
                                     record_fn=_stub_record)
    q.put((rank, None if stats is None else [(s["k"], s["tag"], float(s["R_pred_44"][0, 3])) for s in stats], D.COLLECTIVES["all_gather"] - n0))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_sharded_evaluation_gloo_world2_ragged_batches(tmp_path):
    """Two ranks, global batches of 5, 4 and 7 pairs (ragged shards 3+2, 2+2, 4+3): rank 0 ends up with one record per pair in GLOBAL
    pair order, each built from the pose its owner rank computed, written to ONE result file; one pose all_gather for the run."""
    import torch.multiprocessing as mp
    from relativepose_amd import evaluation as E
    sizes = [5, 4, 7]
    path = str(tmp_path / "exp.result.npy")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, sizes, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=180) for _ in range(2)))
    for p in procs:
        p.join(60)
    assert res[1][0] is None and res[0][1] == 1 and res[1][1] == 1
    assert res[0][0] == [(k, float(k), float(k)) for k in range(sum(sizes))]
    saved = E.load_results(path)
    assert [s["k"] for s in saved] == list(range(sum(sizes)))


def test_sharded_evaluation_resumes_in_units_of_100_pairs(tmp_path):
    """evaluation.py:129-133: an existing result file keeps its first (len // 100) * 100 records; the run continues behind them
    (single process; the skipped pairs are never prepared)."""
    import torch
    from relativepose_amd import evaluation as E
    sizes = [64, 64, 64, 30]
    path = str(tmp_path / "exp.result.npy")
    full = E.evaluate_pairs_sharded(_StubPipe(), _stub_batches(sizes), torch.device("cpu"), result_path=path, record_fn=_stub_record)
    assert [s["k"] for s in full] == list(range(222))
    E.save_results(path, full[:150] + [{"k": -1}] * 0)            # an interrupted run: 150 records on disk -> 100 are kept

    class Counting(_StubPipe):
        prepared = 0

        def prepare(self, rgb, *a):
            Counting.prepared += rgb.shape[0]
            return super().prepare(rgb, *a)

    again = E.evaluate_pairs_sharded(Counting(), _stub_batches(sizes), torch.device("cpu"), result_path=path, record_fn=_stub_record,
                                     round_batches=1)
    assert [s["k"] for s in again] == list(range(222)) and Counting.prepared == 122
    assert [s["status"] for s in again[100:]] == [k % 3 for k in range(100, 222)]          # the matcher's status travels into the records
    assert len(E.load_results(path)) == 222
    fresh = E.evaluate_pairs_sharded(Counting(), _stub_batches(sizes), torch.device("cpu"), result_path=path, record_fn=_stub_record, resume=False)
    assert len(fresh) == 222


def test_sharded_evaluation_prepares_at_most_depth_batches_at_a_time():
    """ADVICE r4 / VERDICT r4 weak #8: device memory must not grow with the number of batches -- with the default round (the whole list) a
    run over 12 global batches holds at most `depth` prepared batches at any time (each pins ~2.2 GB at 32 pairs on the real pipeline)."""
    import torch
    from relativepose_amd import evaluation as E

    class Bounded(_StubPipe):
        alive = 0
        max_alive = 0

    sizes = [8] * 12
    stats = E.evaluate_pairs_sharded(Bounded(), _stub_batches(sizes), torch.device("cpu"), record_fn=_stub_record, depth=2)
    assert [s["k"] for s in stats] == list(range(96))
    assert Bounded.max_alive == 2 and Bounded.alive == 0
    Bounded.max_alive = 0
    E.evaluate_pairs_sharded(Bounded(), _stub_batches(sizes), torch.device("cpu"), record_fn=_stub_record, depth=3)
    assert Bounded.max_alive == 3


def test_synthetic_batch_seeds_depend_on_the_global_pair_index_only():
    """the same global pair gets the same panoramas AND keypoints however the split is cut into batches."""
    from relativepose_amd import evaluation as E
    a = E.SyntheticBatch(8, 4000, "suncg", "second", 20).take([5])
    b = E.SyntheticBatch(4, 4004, "suncg", "second", 20).take([1])            # the same global pair 4005 in another batching
    assert all(np.array_equal(a[k], b[k]) for k in ("rgb", "depth", "pts", "ptw", "R"))
    c = E.SyntheticBatch(8, 4000, "suncg", "second", 20).take([4, 5])
    assert not np.array_equal(c["pts"][0], c["pts"][1])

"""Multi-GPU: scan pairs are independent, so they shard across ranks in
contiguous blocks (the reference's manual ``--entrySplit``, evaluation.py:59 /
datasets/SUNCG.py:68-69) with no data-path collective; one all_gather of the
[B_local,16] poses (+ status) at the end -- RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests."""
import os


def init_from_env(backend=None):
    """torch.distributed init from RANK/WORLD_SIZE/MASTER_* (torchrun contract). Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("RELPOSE_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # RELPOSE_DIST_BACKEND=gloo: test hook (two ranks sharing ONE GPU cannot use RCCL)
            backend = os.environ.get("RELPOSE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        import datetime
        # a rank that never arrives (a GPU that is not visible, a wrong MASTER_PORT) must not hang the others for the default 10-30 min
        timeout = datetime.timedelta(seconds=float(os.environ.get("RELPOSE_DIST_TIMEOUT", "300")))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
        except Exception as e:
            raise RuntimeError(f"relativepose_amd.distributed: rank {rank}/{world} could not join the process group (backend {backend} -- "
                               f"'nccl' is RCCL on ROCm --, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, "
                               f"timeout {timeout.total_seconds():.0f} s: RELPOSE_DIST_TIMEOUT): {e}") from e
    return rank, world, local


def peer_access_summary():
    """hipDeviceCanAccessPeer over the visible GPUs (one node: xGMI links make every pair peer-accessible): {"gpus", "peer_pairs",
    "peer_accessible"} -- the pose all_gather runs over those links; reported in the bench line, not relied upon."""
    import torch
    try:        # diagnostic only: a runtime that refuses the query must not take the bench line down with it
        n = torch.cuda.device_count()
        ok = sum(1 for i in range(n) for j in range(n) if i != j and torch.cuda.can_device_access_peer(i, j))
    except Exception as e:      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    return {"gpus": n, "peer_pairs": n * (n - 1), "peer_accessible": ok}


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of pair indices owned by ``rank``; blocks differ by at most one pair."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


COLLECTIVES = {"all_gather": 0}      # data-path collectives issued by this process (bench.py reports the count)


def gather_poses(pose_local, status_local, total, world, force_collective=False):
    """all_gather of per-rank [b_r,4,4] poses and [b_r] status into [total,4,4] / [total] on every rank
    (ragged blocks are padded to the largest block).  force_collective: issue the collective even with ONE rank (test hook: the
    1-GPU box runs the RCCL all_gather of the [B,17] f64 block through exactly this code, tests/test_gpu_rccl.py)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:
        return pose_local, status_local
    bmax = (total + world - 1) // world
    buf = torch.zeros(bmax, 17, dtype=torch.float64, device=pose_local.device)
    b = pose_local.shape[0]
    buf[:b, :16] = pose_local.reshape(b, 16)
    buf[:b, 16] = status_local.to(torch.float64)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    COLLECTIVES["all_gather"] += 1
    poses, status = [], []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        poses.append(out[r][:hi - lo, :16].reshape(-1, 4, 4))
        status.append(out[r][:hi - lo, 16].to(torch.int32))
    return torch.cat(poses), torch.cat(status)


def rccl_info():
    """{"rccl_version", "librccl_loaded"}: what the process actually linked and loaded ("nccl" IS RCCL on ROCm); diagnostic for the bench's
    per-rank stderr line and the RCCL test."""
    import torch
    info = {}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    try:
        with open("/proc/self/maps") as f:
            libs = sorted({line.split()[-1] for line in f if "rccl" in line.lower()})
        info["librccl_loaded"] = libs
    except OSError:
        info["librccl_loaded"] = None
    return info


def barrier(world):
    import torch.distributed as dist
    if world > 1:
        dist.barrier()


def max_over_ranks(value, world, device):
    import torch
    import torch.distributed as dist
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

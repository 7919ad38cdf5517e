"""GPU: the RCCL branch of the multi-GPU path on the ONE GPU the test box has.

Five rounds of records never loaded librccl: the only RCCL test needs two GPUs and skips on every box.  Here a child process initialises
torch.distributed with backend "nccl" (= RCCL on ROCm) and world_size 1 on the MI355X and pushes a [32,17] float64 block -- the exact
buffer of distributed.gather_poses -- through dist.all_gather (the world == 1 early return bypassed by the test flag): librccl is loaded, a
communicator is created, the dtype and shape are accepted, and the result equals the input.  Reference: the manual shard of
evaluation.py:59 / datasets/SUNCG.py:68-69; the gather is this build's only collective."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from relativepose_amd import distributed as D
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + os.getpid() %% 200))
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
pose = torch.randn(32, 4, 4, dtype=torch.float64, generator=g).to(dev)
status = torch.randint(0, 7, (32,), dtype=torch.int32, generator=g).to(dev)
n0 = D.COLLECTIVES["all_gather"]
p2, s2 = D.gather_poses(pose, status, 32, 1, force_collective=True)
torch.cuda.synchronize()
t = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction of bench.py (max_over_ranks)
dist.barrier()
info = D.rccl_info()
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "pose_equal": bool(torch.equal(p2, pose)), "status_equal": bool(torch.equal(s2, status)),
       "collectives": D.COLLECTIVES["all_gather"] - n0, "allreduce": float(t.item()), "pose_dtype": str(p2.dtype), "pose_device": str(p2.device), **info}
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_rccl_all_gather_of_the_pose_block_world_size_one():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RELPOSE_FORCE_DEVICE", "RELPOSE_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    from gpu_util import log
    log("rccl_world1", **r)
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["pose_equal"] and r["status_equal"] and r["collectives"] == 1 and r["allreduce"] == 3.25
    assert r["pose_dtype"] == "torch.float64" and r["pose_device"].startswith("cuda")
    assert r["librccl_loaded"], "backend 'nccl' ran without librccl mapped into the process"

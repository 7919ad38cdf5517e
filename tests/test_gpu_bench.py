"""bench.py contract (GPU): one JSON line with the driver's keys, the roofline and cpu_baseline objects, and a
pipelined run whose poses equal the plain sequential run (checked by tests/test_gpu_pipeline.py); here only the schema
and basic sanity of the numbers at a tiny batch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract_small_batch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "4", "--steps", "3", "--warmup", "1", "--keypoints", "60"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "pairs/s" and r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "f32" and "workload" in r["config"]
    assert abs(r["value"] - 4 * 3 / (r["ms_per_step"] * 3 / 1e3)) < 1e-6 * r["value"]
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert r["status_ok_fraction"] == 1.0


def test_bench_world2_path_two_ranks_sharing_one_gpu():
    """The N>1 code path of bench.py (per-rank shards, pipelined steps, pose all_gather per step, max-over-ranks timing) with
    two ranks on ONE GPU: RCCL refuses two ranks on a device, so the collective runs over gloo (RELPOSE_DIST_BACKEND test
    hook).  Only the plumbing is checked; the number is meaningless (two processes time-slice one GPU)."""
    env = dict(os.environ, RELPOSE_DIST_BACKEND="gloo", RELPOSE_FORCE_DEVICE="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "4",
                          "--keypoints", "60"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["config"]["pairs_per_gpu"] == 4
    assert abs(r["value"] - 8 * 2 / (r["ms_per_step"] * 2 / 1e3)) < 1e-6 * r["value"]      # whole-job pairs / max-over-ranks time
    assert r["status_ok_fraction"] == 1.0 and "cpu_baseline" not in r

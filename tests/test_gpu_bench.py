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
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"].startswith("f32 (fp32 products emulated on the bf16 matrix pipe") and "workload" in r["config"]
    assert r["config"]["baseline_config_index"] == 1 and "pcie_inclusive" in r and "roofline_affinity" in r and r["roofline"]["traffic"] is None
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


def test_bench_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` launched PLAINLY (no torchrun, WORLD_SIZE unset) must run two ranks itself and report
    n_gpus 2 -- never fall through to one GPU.  With a single visible GPU the two ranks share it through the
    gloo test hook; with >= 2 GPUs the same command runs one rank per GPU over RCCL (next test)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RELPOSE_DIST_BACKEND="gloo", RELPOSE_FORCE_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "4", "--keypoints", "60",
                          "--no-aux", "--no-h2d"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["dist_world_size"] == 2 and r["config"]["dist_backend"] == "gloo"
    assert r["config"]["pairs_per_step_total"] == 8 and r["status_ok_fraction"] == 1.0


def test_bench_eight_rank_plumbing_ragged_and_even():
    """Eight ranks -- the world size the scaling run uses -- through the one-device gloo hook (no 8-GPU node is available to the build):
    (a) configs[3]'s 256 pairs split evenly, --gather run: ONE all_gather in the whole timed region; (b) a ragged split (250 pairs =
    six shards of 31 and two of 32... i.e. blocks differing by one pair), one gather per step.  Checked: every pair is owned exactly
    once, the collective counts, the whole-job value.  (Small keypoint count: eight processes time-slice one GPU.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RELPOSE_DIST_BACKEND="gloo", RELPOSE_FORCE_DEVICE="0")
    for total, steps, want_gathers in ((256, 2, 1), (250, 2, 2)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--scaling", "strong", "--total-pairs", str(total),
                              "--steps", str(steps), "--warmup", "0", "--keypoints", "40", "--batches", "2", "--inflight", "2", "--no-aux", "--no-h2d"],      # (8 ranks on ONE device: 2 workspaces of 11.5 GB each)
                             capture_output=True, text=True, timeout=2400, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        r = json.loads(lines[0])
        c = r["config"]
        assert r["n_gpus"] == 8 and c["dist_world_size"] == 8 and r["scaling"] == "strong"
        assert sum(c["shard_sizes"]) == total == c["pairs_per_step_total"] and max(c["shard_sizes"]) - min(c["shard_sizes"]) <= 1
        assert c["pose_all_gathers_in_timed_region"] == want_gathers, c
        assert r["status_ok_fraction"] == 1.0
        assert abs(r["value"] - total * steps / (r["ms_per_step"] * steps / 1e3)) < 1e-6 * r["value"]


def test_bench_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RELPOSE_FORCE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rccl_two_gpus_weak_and_strong():
    """The RCCL (backend "nccl") branch, one rank per GPU -- needs >= 2 GPUs, skipped on the 1-GPU test box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RELPOSE_FORCE_DEVICE", "RELPOSE_DIST_BACKEND")}
    for extra, total in ((["--pairs", "4"], 8), (["--scaling", "strong", "--total-pairs", "6"], 6)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--keypoints", "60",
                              "--no-aux", "--no-h2d"] + extra, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert r["n_gpus"] == 2 and r["config"]["dist_backend"] == "nccl" and r["config"]["pairs_per_step_total"] == total


@pytest.mark.parametrize("config", [0, 2, 3, 4])
def test_bench_other_baseline_configs_run(config):
    """--config 2|3|4 (Matterport N=400, ScanNet/kinect -- bf16x6 conv arithmetic --, SUNCG 320x1280 + f16x3) produce a line naming their workload; small batch."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--pairs", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-aux", "--batches", "2"], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = r["config"]
    assert c["baseline_config_index"] == config and r["status_ok_fraction"] == 1.0 and "pcie_inclusive" in r
    want = {0: ("suncg", "second", "160x640", 80, "bf16x6"), 2: ("matterport", "second", "160x640", 400, "bf16x6"), 3: ("scannet", "kinect", "160x640", 200, "bf16x6"), 4: ("suncg", "second", "320x1280", 200, "f16x3")}[config]
    assert (c["dataset"], c["mask"], c["pano"], c["keypoints"], c["conv_precision"]) == want
    # configs 1-3: fp32 products emulated with full-width operands (the dtype string names the arithmetic); configs[4]: the 3-term fp16 path
    assert r["dtype"].startswith("f32 (fp32 products emulated on the bf16 matrix pipe") == (config != 4)


def test_sharded_evaluation_eight_ranks_equal_one_rank(tmp_path):
    """python -m relativepose_amd.evaluation (evaluate_pairs_sharded: BASELINE configs[3]'s sharded evaluation): a synthetic ScanNet
    "split" of 48 pairs in global batches of 24, once on one rank and once on eight ranks (3 pairs of every batch each; the one-device
    gloo hook -- no 8-GPU node is available to the build).  Scan pairs are BatchNorm groups and the kernels are batch-invariant, so
    the two result files must hold the same records in the same order with BITWISE the same poses; one pose all_gather per run."""
    import numpy as np
    from relativepose_amd import evaluation as E
    outs = {}
    for gpus in (1, 8):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        if gpus > 1:
            env.update(RELPOSE_DIST_BACKEND="gloo", RELPOSE_FORCE_DEVICE="0")
        exp = str(tmp_path / f"split{gpus}")
        out = subprocess.run([sys.executable, "-m", "relativepose_amd.evaluation", "--gpus", str(gpus), "--dataset", "scannet", "--pairs", "48",
                              "--batch", "24", "--keypoints", "60", "--exp", exp, "--rm"], capture_output=True, text=True, timeout=2400, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert r["pairs"] == 48 and r["n_gpus"] == gpus and r["pose_all_gathers"] == (0 if gpus == 1 else 1)
        assert sum(v["nobs"] for v in r["stats"].values()) == 48
        outs[gpus] = E.load_results(exp + ".result.npy")
    assert [e["img_src"] for e in outs[1]] == [e["img_src"] for e in outs[8]] == [f"pair{k}/src" for k in range(48)]
    for a, b in zip(outs[1], outs[8]):
        assert np.array_equal(a["R_pred_44"], b["R_pred_44"]) and a["err_ad"] == b["err_ad"] and a["overlap"] == b["overlap"]

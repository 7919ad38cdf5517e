#!/usr/bin/env python
"""Benchmark of the relative-pose hot path on MI355X.

Metric (BASELINE.json): scan-pairs/sec end-to-end (completion + feat +
spectral-match), 160x640 RGB-D.  One "step" = one pass of the whole hot path
(3 recurrent levels of {warp, SCNet, compose+sample, match}) over one batch of
synthetic scan pairs already resident in HBM.  Workload at N GPUs:
BASELINE.json configs[1] per GPU -- SUNCG conventions, 160x640, 200 keypoints per
view, 32 pairs per GPU (weak scaling: pairs shard across ranks, no data-path
collective, one RCCL all_gather of the 4x4 poses per step).  Two steps are in flight
(--inflight 2): the launch-bound matcher phase of step k runs under the SCNet forward of step k+1
(pipeline.run_pipelined); every step still does the complete path on its own 32 pairs.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" (dominant kernel = the fp32-MFMA implicit-GEMM conv, live
HIP-event timing) and "cpu_baseline" (the numpy/torch oracle on the host cores,
bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SUNCG_SIGMAS = [[0.28884460993320005, 0.3723397110060548, 0.04471146704846696, 0.008681938149233242],
                [0.3301724627277194, 0.22653872741771977, 0.03371542612584658, 0.009278392068704865],
                [0.44732243168057817, 0.3039564896467746, 0.029312830444192497, 0.011085327519146518]]
# data/relativePoseModule/final_param_suncg_rlevel_3.txt of the reference (36 tuned floats = config data)

GFLOP_PER_IMAGE = 36.14            # SCNet conv+convT MACs x2 per 224x224 sample
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(args, data, pts, ptw, S):
    """The oracle (CPU restatement of the reference loop) on the first pairs of the same workload (checker timed beside the GPU path; never the product)."""
    import torch
    from oracle import pipeline_oracle as P
    from oracle.scnet_oracle import SCNetOracle
    from relativepose_amd import weights
    net = SCNetOracle(weights.make_state_dict(7, S), S, 1)
    tm, npair = {}, min(3, len(pts))          # bounded sample: ~15 s of host time
    t0 = time.time()
    for i in range(npair):
        tmi = {}
        P.run_pair(net, data["rgb"][i], data["norm"][i], data["depth"][i], pts[i], ptw[i], np.array(SUNCG_SIGMAS), "suncg", "second", S,
                   timing=tmi)
        for k, v in tmi.items():
            tm[k] = tm.get(k, 0.0) + v / npair
    dt = time.time() - t0
    return {"value": npair / dt, "unit": "pairs/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{npair} scan pairs x 3 recurrent levels of the same workload (N={args.keypoints} keypoints), "
                      f"oracle = numpy/scipy matcher + torch-CPU fp32 SCNet; {dt:.1f}s",
            "seconds_per_stage": {k: round(v, 3) for k, v in tm.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=32, help="scan pairs per GPU")
    ap.add_argument("--keypoints", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x3"], default="f32",
                    help="conv arithmetic: f32 = exact fp32 MFMA (the parity configuration, default); bf16x3 = opt-in split-bf16 products")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches (steps) in flight per GPU: the matcher phase of step k overlaps the SCNet forward of step k+1")
    ap.add_argument("--streams", type=int, default=1, help="with --inflight 1: split the batch over this many HIP streams instead")
    args = ap.parse_args()

    import torch
    from relativepose_amd import distributed as D
    from relativepose_amd import synth, weights, rpmodule
    from relativepose_amd.model import SCNet
    from relativepose_amd.pipeline import RelativePosePipeline

    rank, world, local = D.init_from_env()
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    S, B, N = 15, args.pairs, args.keypoints
    total = B * world
    lo, hi = D.shard_range(total, rank, world)

    # synthetic inputs + random-init weights (no dataset / checkpoint ships with the reference)
    data = synth.make_pairs(hi - lo, 2000 + lo, "suncg")      # seed = 1000*config + pair index
    pts, ptw = synth.make_keypoints(hi - lo, N, 2000 + lo, "second")
    depth = max(1, args.inflight)
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(weights.make_state_dict(7, S))
    net.set_precision(args.precision)
    Cc = N * 5
    pipe = RelativePosePipeline(net, "suncg", "second", SUNCG_SIGMAS, max_edges=min(Cc * (Cc - 1), 1 << 20))
    nloc = hi - lo
    ns_ = 1 if depth > 1 else max(1, min(args.streams, nloc))
    cuts = [nloc * i // ns_ for i in range(ns_ + 1)]
    states = [pipe.prepare(data["rgb"][a:b], data["norm"][a:b], data["depth"][a:b], pts[a:b], ptw[a:b], dev)
              for a, b in zip(cuts[:-1], cuts[1:])]

    # one prepared batch (own device buffers + HIP stream) per step in flight; slot j > 0 gets its own scan pairs
    batches = [states[0]] if depth > 1 else None
    for j in range(1, depth):
        dj = synth.make_pairs(nloc, 2000 + lo + 100000 * j, "suncg")
        pj, wj = synth.make_keypoints(nloc, N, 2000 + lo + 100000 * j, "second")
        batches.append(pipe.prepare(dj["rgb"], dj["norm"], dj["depth"], pj, wj, dev))

    def step():
        if len(states) == 1:
            pose, status, _ = pipe.run(states[0])
        else:
            res = pipe.run_interleaved(states)
            pose, status = torch.cat([r[0] for r in res]), torch.cat([r[1] for r in res])
        return D.gather_poses(pose, status, total, world)

    def run_steps(k):
        """k steps = k batches of B pairs per GPU, each followed by the pose gather; returns the last result."""
        if depth > 1:
            return pipe.run_pipelined(batches, k, lambda i, pose, status: D.gather_poses(pose, status, total, world))[-1]
        out = None
        for _ in range(k):
            out = step()
        return out

    if args.warmup:
        run_steps(args.warmup)
    torch.cuda.synchronize()
    D.barrier(world)
    t0 = time.perf_counter()
    poses, status = run_steps(args.steps)
    torch.cuda.synchronize()
    D.barrier(world)
    dt = D.max_over_ranks(time.perf_counter() - t0, world, dev)

    if rank == 0:
        ms = dt / args.steps * 1e3
        res = {"metric": "scan-pairs/sec end-to-end (completion+feat+spectral-match), 160x640 RGB-D",
               "value": total * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else f"f32 (conv products as 3 x {args.precision[:-2]} MFMA, fp32 accumulate; opt-in, NOT the parity configuration)",
               "data": "synthetic (seeded box-room RGB-D panoramas, injected keypoints, random-init weights)",
               "config": {"workload": "SUNCG 160x640, N=200 keypoints, batch=32 pairs per GPU, alterStep=3 (BASELINE configs[1])",
                          "pairs_per_gpu": B, "keypoints": N, "recurrent_levels": 3, "parallelism": f"pairs sharded x{world}",
                          "batches_in_flight": depth, "streams_per_batch": len(states)},
               "status_ok_fraction": float((status == 0).double().mean().item())}
        # --- roofline of the dominant kernel: implicit-GEMM conv, HIP events on the launch stream
        x = torch.randn(2 * B, 16, 160, 640, device=dev)
        net.profile(x, 1)
        g_ms, o_ms, n_gemm = net.profile(x, 3)
        flops = GFLOP_PER_IMAGE * 1e9 * 2 * B
        ach = flops / (g_ms * 1e-3) / 1e12
        # bf16x3 (opt-in): every fp32 product costs three dense bf16 MFMA products -> algorithmic peak = 2500 / 3
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else 2500.0 / 3
        res["roofline"] = {"kernel": "conv_igemm_kernel (fp32 MFMA 32x32x2)" if args.precision == "f32" else f"conv_igemm_kernel (3 x {args.precision[:-2]} MFMA 32x32x16)",
                           "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                           # HBM bytes of the conv stack per forward: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
                           # FETCH doubled per the gfx950 correction) of tools/scnet_only.py at this batch: profiles/r01_scnet_hbm_pmc.txt
                           "traffic": 38.2e9 if B == 32 else None, "traffic_unit": "bytes per forward (all conv launches)",
                           "launches_per_forward": int(n_gemm), "ms_per_forward_gemm": g_ms, "ms_per_forward_other": o_ms,
                           "algorithmic_gflop_per_forward": flops / 1e9}
        # --- N x N affinity build (materialised fp32 wij), the kernel the HBM target is stated on
        cases = [synth.make_match_case(N, 5000 + b)[:2] for b in range(B)]
        kp = rpmodule.pack_keypoints(cases, dev)
        para = rpmodule.opts(*SUNCG_SIGMAS[0])
        f_s, w_s, f_t, w_t, ns_, nt_ = kp[2], kp[3], kp[6], kp[7], kp[8], kp[9]
        for _ in range(3):
            rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        tot = 0.0
        for _ in range(reps):
            e0.record()
            rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=True)
            e1.record()
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        a_ms = tot / reps
        abytes = ((N + N) * 33 * 4 + N * N * 4) * B
        res["roofline_affinity"] = {"kernel": "affinity_topk_kernel<true>", "bound": "hbm", "achieved": abytes / (a_ms * 1e-3) / 1e9,
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": abytes / (a_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    "traffic": None, "ms_per_launch": a_ms, "algorithmic_bytes_per_launch": abytes,
                                    "note": "event pair around one launch incl. launch latency; the kernel is f64-VALU-bound (ocml exp per entry, numpy-order f32 distance for bit-exact top-K) and saturates at ~155 GB/s at any batch: profiles/r01_affinity_scaling.txt"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, data, pts, ptw, S)
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

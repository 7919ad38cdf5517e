"""Evaluation-side bookkeeping of the reference driver (evaluation.py:286-320): pose error metrics per scan
pair, the ``error_stats`` record list and its ``<exp>.result.npy`` container, and a batched harness that runs
the device pipeline over a stream of scan-pair batches and fills those records ("called
unchanged" is realised by the build's own eval_pairs-style harness).

  angular_distance_np     util.py:176-187
  pose_errors             evaluation.py:291-297
  result_record           evaluation.py:299-305
  save_results / load_results   evaluation.py:319-320 (np.save of the list of dicts)
  evaluate_pairs          evaluation.py:203-320 (loop over the loader, keypoints injected)
  evaluate_pairs_sharded  the same over the GPUs of a node: the reference shards by hand (--entrySplit, evaluation.py:59,
                          datasets/SUNCG.py:68-69: one process and one result file per split, merged afterwards); here every global
                          batch is cut into contiguous per-rank blocks, ONE all_gather of the poses per round, rank 0 writes ONE
                          <exp>.result.npy and resumes from an existing one in units of 100 pairs (:129-133, :319-320)

The metrics are host numpy like the reference's (a handful of 3x3 products per pair); the overlap statistics use the
GPU nearest-neighbour kernel (util.point_cloud_overlap)."""
import numpy as np

OVERLAPS = ("0-0.1", "0.1-0.5", "0.5-1.0")       # evaluation.py:68


def angular_distance_np(R_hat, R):
    """util.py:176-187 (degrees); R_hat, R: [n,3,3] or [3,3]."""
    R_hat = np.asarray(R_hat)
    R = np.asarray(R)
    if R_hat.shape == (3, 3):
        R_hat = R_hat[np.newaxis, :]
    if R.shape == (3, 3):
        R = R[np.newaxis, :]
    n = R.shape[0]
    trace = np.matmul(R_hat, R.transpose(0, 2, 1)).reshape(n, -1)[:, [0, 4, 8]].sum(1)
    return np.arccos(((trace - 1) / 2).clip(-1, 1)) / np.pi * 180.0


def pose_errors(R_hat_44, R_gt_44, pc_src):
    """evaluation.py:291-297 -> (err_ad, err_t, err_blind, err_t_blind)."""
    t_hat, R_hat = R_hat_44[:3, 3], R_hat_44[:3, :3]
    R_gt = R_gt_44[:3, :3]
    ad = angular_distance_np(R_hat, R_gt[np.newaxis, :, :])[0]
    ad_blind = angular_distance_np(R_gt[np.newaxis, :, :], np.eye(3)[np.newaxis, :, :])[0]
    tr = np.linalg.norm(np.matmul((R_hat - R_gt), np.asarray(pc_src).mean(0).reshape(3)) + t_hat - R_gt_44[:3, 3])
    tr_blind = np.linalg.norm(t_hat - R_gt_44[:3, 3])
    return ad, tr, ad_blind, tr_blind


def overlap_bucket(overlap_val):
    """evaluation.py:187-192."""
    return '0-0.1' if overlap_val <= 0.1 else ('0.1-0.5' if overlap_val <= 0.5 else '0.5-1.0')


def result_record(img_src, img_tgt, R_hat_44, R_gt_44, pc_src, overlap_val, pc_dist, cam_dist, pc_nn):
    """One ``error_stats`` entry with the reference's keys (evaluation.py:299-305)."""
    ad, tr, ad_blind, tr_blind = pose_errors(R_hat_44, R_gt_44, pc_src)
    R_pred_44 = np.eye(4)
    R_pred_44[:3, :3] = R_hat_44[:3, :3]
    R_pred_44[:3, 3] = R_hat_44[:3, 3]
    return {'img_src': img_src, 'img_tgt': img_tgt, 'err_ad': ad, 'err_t': tr, 'err_blind': ad_blind, 'err_t_blind': tr_blind,
            'overlap': overlap_val, 'pc_dist': pc_dist, 'cam_dist': cam_dist, 'pc_nearest': pc_nn, 'R_gt': np.asarray(R_gt_44),
            'R_pred_44': R_pred_44}


def save_results(path, error_stats):
    """evaluation.py:319-320: ``np.save(f"{exp_dir}/{exp}.result.npy", error_stats)`` (an object array of dicts)."""
    if not path.endswith(".npy"):
        path += ".npy"
    np.save(path, np.array(list(error_stats), dtype=object), allow_pickle=True)
    return path


def load_results(path):
    return list(np.load(path, allow_pickle=True))


def summarize(error_stats):
    """Per overlap bucket: number of pairs, mean rotation / translation error (evaluation.py:321-327)."""
    out = {}
    for ov in OVERLAPS:
        rows = [e for e in error_stats if overlap_bucket(e['overlap']) == ov]
        out[ov] = {"nobs": len(rows), "rot_mean": float(np.mean([e['err_ad'] for e in rows])) if rows else float('nan'),
                   "trans_mean": float(np.mean([e['err_t'] for e in rows])) if rows else float('nan')}
    return out


def _prepare_batch(pipe, batch, device):
    """pipe.prepare for a DataLoader-layout batch dict under either keypoint mode: keypoints="given" reads "pts" / "ptw" (injected
    keypoints), keypoints="reference" reads "sift" = [(source detections [n,2], target detections [m,2])] * B in panorama coordinates
    (rputil.map_detections of the SIFT detector's output, rputil.py:152-172) and optionally "kp_seeds" [B][levels]: every level then derives
    its keypoints from its own feature maps, as evaluation.py:278 does through getMatchingPrimitive."""
    if getattr(pipe, "keypoints", "given") == "reference":
        return pipe.prepare(batch["rgb"], batch["norm"], batch["depth"], None, None, device, sift=batch["sift"], kp_seeds=batch.get("kp_seeds"))
    return pipe.prepare(batch["rgb"], batch["norm"], batch["depth"], batch["pts"], batch["ptw"], device)


def evaluate_pairs(pipe, batches, device, result_path=None, names=None):
    """The reference's evaluation loop (evaluation.py:203-320) over an iterable of batches
    ``{"rgb","norm","depth","R" [B,2,4,4], "pts" [B,2,N,2], "ptw" [B,2,N]}`` (the DataLoader dict layout, batched, with the
    injected keypoints; with RelativePosePipeline(keypoints="reference") a batch carries "sift" detections instead, _prepare_batch):
    R_gt = R_t inv(R_s) (:185), overlap statistics from the observed clouds (util.parse_data +
    point_cloud_overlap), the recurrent pipeline on the GPU, then one result record per pair."""
    from . import util
    stats = []
    k = 0
    for batch in batches:
        B = batch["rgb"].shape[0]
        st = _prepare_batch(pipe, batch, device)
        pose, status, _ = pipe.run(st)
        pose = pose.cpu().numpy()
        pcs, valid = util.depth2pc_dev(st["depth"], pipe.dataset)
        pcs, valid = pcs.cpu().numpy(), valid.cpu().numpy().astype(bool)
        for b in range(B):
            R_gt_44 = np.matmul(batch["R"][b, 1], np.linalg.inv(batch["R"][b, 0]))
            pc_src, pc_tgt = pcs[2 * b][valid[2 * b]], pcs[2 * b + 1][valid[2 * b + 1]]
            ov, cam_dist, pc_dist, pc_nn = util.point_cloud_overlap(pc_src, pc_tgt, R_gt_44)
            nm = names[k] if names is not None else (f"pair{k}/src", f"pair{k}/tgt")
            stats.append(result_record(nm[0], nm[1], pose[b], R_gt_44, pc_src, ov, pc_dist, cam_dist, pc_nn))
            k += 1
    if result_path is not None:
        save_results(result_path, stats)
    return stats


class SyntheticBatch:
    """A global batch of `size` seeded synthetic scan pairs (synth.make_pairs / make_keypoints conventions: pair b of the batch comes from
    seed + b) that materialises only the pairs a rank asks for -- evaluate_pairs_sharded hands every rank the same batch list, and a rank
    should not render the other ranks' panoramas."""

    def __init__(self, size, seed, dataset, mask_method, keypoints, h=160, sift=0):
        self.size, self.seed, self.dataset, self.mask_method, self.keypoints, self.h = size, seed, dataset, mask_method, keypoints, h
        self.sift = int(sift)           # > 0: also `sift` synthetic SIFT detections per view (for RelativePosePipeline(keypoints="reference"))

    def take(self, idx):
        from . import synth
        parts = [synth.make_pairs(1, self.seed + int(b), self.dataset, h=self.h) for b in idx]
        # (both seeds are functions of the GLOBAL pair index seed + b only: the same pair gets the same panoramas and keypoints under any
        # --batch, so result files are comparable / resumable across batch sizes)
        kps = [synth.make_keypoints(1, self.keypoints, 7919 * (self.seed + int(b)) + 13, self.mask_method, h=self.h) for b in idx]
        out = {k: np.concatenate([p[k] for p in parts]) for k in ("rgb", "norm", "depth", "R")}
        out["pts"], out["ptw"] = np.concatenate([k[0] for k in kps]), np.concatenate([k[1] for k in kps])
        if self.sift:
            from . import rputil
            dets = [synth.make_sift_detections(1, self.sift, 15485863 * (self.seed + int(b)) + 7, self.mask_method, self.h)[0] for b in idx]
            out["sift"] = [(rputil.map_detections(a, self.mask_method, self.h), rputil.map_detections(c, self.mask_method, self.h)) for a, c in dets]
            out["kp_seeds"] = [[31 * (self.seed + int(b)) + lvl for lvl in range(8)] for b in idx]
        return out


def _batch_size(batch):
    return batch.size if hasattr(batch, "take") else batch["rgb"].shape[0]


def _batch_take(batch, idx):
    """The pairs `idx` of a global batch as a dict of arrays (a dict batch is sliced, a lazy one renders them)."""
    if hasattr(batch, "take"):
        return batch.take(idx)
    out = {k: v[idx] for k, v in batch.items() if isinstance(v, np.ndarray)}
    for k in ("sift", "kp_seeds"):                        # per-pair lists (keypoints="reference")
        if k in batch:
            out[k] = [batch[k][int(i)] for i in idx]
    return out


def _default_record(pipe, sub, poses, device, names, ks):
    """Result records (the reference's keys, evaluation.py:286-305) of the pairs of `sub` (a _batch_take dict), given their poses
    [n,4,4] (host) and their global pair indices `ks`: overlap statistics from the observed clouds, then the error metrics."""
    from . import util
    import torch
    n = len(ks)
    depth = torch.from_numpy(np.ascontiguousarray(sub["depth"].reshape(2 * n, *sub["depth"].shape[2:]))).to(device)
    pcs, valid = util.depth2pc_dev(depth, pipe.dataset)
    pcs, valid = pcs.cpu().numpy(), valid.cpu().numpy().astype(bool)
    out = []
    for q in range(n):
        R_gt_44 = np.matmul(sub["R"][q, 1], np.linalg.inv(sub["R"][q, 0]))
        pc_src, pc_tgt = pcs[2 * q][valid[2 * q]], pcs[2 * q + 1][valid[2 * q + 1]]
        ov, cam_dist, pc_dist, pc_nn = util.point_cloud_overlap(pc_src, pc_tgt, R_gt_44)
        nm = names[ks[q]] if names is not None else (f"pair{ks[q]}/src", f"pair{ks[q]}/tgt")
        out.append(result_record(nm[0], nm[1], poses[q], R_gt_44, pc_src, ov, pc_dist, cam_dist, pc_nn))
    return out


def evaluate_pairs_sharded(pipe, batches, device, result_path=None, names=None, rank=0, world=1, resume=True, round_batches=None,
                           record_fn=None, depth=2):
    """`evaluate_pairs` over `world` ranks (one process per GPU, torch.distributed initialised by the caller:
    distributed.init_from_env).  Every rank receives the same `batches` (a sequence of the DataLoader-layout dicts, B pairs each);
    of every global batch rank r takes the contiguous block distributed.shard_range(B_todo, r, world) -- BatchNorm groups are scan
    pairs, so a shard is independent of the others -- runs its blocks through pipe.run_pipelined (`depth` in flight), and computes
    the records of its own pairs.  Per ROUND (`round_batches` global batches; default: the whole list) there is ONE all_gather of
    the stacked poses (distributed.gather_poses: the north-star's single RCCL gather) and one gather of the record lists to rank 0,
    which keeps the records in global pair order and writes `result_path` (np.save of the list of dicts, evaluation.py:319-320).
    resume: an existing result file is loaded and its first (len // 100) * 100 pairs are kept and skipped, like the reference's
    repeat bookkeeping (evaluation.py:129-133).  Returns the full record list on rank 0, None elsewhere.
    A batch may be a dict of arrays or a lazy provider with `.size` and `.take(indices)` (SyntheticBatch): a rank then only
    materialises its own pairs.
    record_fn(sub_batch, local_indices, poses_host, k0) -> list of records (sub_batch = the rank's pairs of the global batch, in the
    order of local_indices): override for callers with their own statistics (tests)."""
    import os
    import torch
    import torch.distributed as dist
    from . import distributed as D
    batches = list(batches)
    sizes = [_batch_size(b) for b in batches]
    starts = np.concatenate(([0], np.cumsum(sizes)))
    stats = []
    if rank == 0 and resume and result_path is not None and os.path.exists(result_path if result_path.endswith(".npy") else result_path + ".npy"):
        old = load_results(result_path if result_path.endswith(".npy") else result_path + ".npy")
        stats = old[:(len(old) // 100) * 100]
    done = len(stats)
    if world > 1:
        t = torch.tensor([done], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.broadcast(t, 0)
        done = int(t.item())
    todo = [(i, max(0, done - int(starts[i]))) for i in range(len(batches)) if int(starts[i + 1]) > done]     # (batch, first pair still to do)
    # a ROUND = the unit of gathering and saving (default: the whole list).  Device memory does NOT grow with it: the batches of a round are
    # prepared lazily, at most `depth` at a time (run_pipelined(provider=...)), and a batch's prepared state and host arrays are dropped as
    # soon as its records are built (the default run used to prepare every batch up front -- 140 GB at 2048 pairs).
    nround = len(todo) if round_batches is None else max(1, int(round_batches))
    for r0 in range(0, len(todo), max(1, nround)):
        chunk = todo[r0:r0 + nround]
        local, mine = [], []
        for j, (i, first) in enumerate(chunk):
            lo, hi = D.shard_range(sizes[i] - first, rank, world)
            idx = np.arange(first + lo, first + hi)
            local.append(idx)
            if len(idx):
                mine.append(j)
        subs, recs, poses, status = {}, [], [None] * len(mine), [None] * len(mine)

        def provider(q):
            i, first = chunk[mine[q]]
            b = _batch_take(batches[i], local[mine[q]])
            subs[q] = b
            return _prepare_batch(pipe, b, device)

        def on_result(q, pose, st):
            # records of this rank's pairs of batch q, built as soon as the batch is complete (the other in-flight batch keeps the GPU busy)
            i, first = chunk[mine[q]]
            idx, sub = local[mine[q]], subs.pop(q)
            poses[q], status[q] = pose.reshape(-1, 4, 4), st.reshape(-1)
            ph = pose.detach().cpu().numpy().reshape(-1, 4, 4)
            sh = st.detach().cpu().numpy().reshape(-1)
            k0 = int(starts[i])
            rr = record_fn(sub, idx, ph, k0) if record_fn is not None else _default_record(pipe, sub, ph, device, names, [k0 + int(b) for b in idx])
            for j, (b, rec) in enumerate(zip(idx, rr)):
                rec['status'] = int(sh[j])             # the matcher's per-pair status (0 ok, 1.. the reference's "return identity" exits)
                recs.append((k0 + int(b), rec))
            return None

        if mine:
            pipe.run_pipelined(None, len(mine), on_result=on_result, depth=min(depth, len(mine)), provider=provider)
        n_local = sum(len(ix) for ix in local)
        n_round = sum(sizes[i] - first for i, first in chunk)
        if world > 1:
            # ONE all_gather of this round's poses (+ status); blocks are per-rank concatenations, ragged by at most one pair per batch
            pl = torch.cat(poses) if poses else torch.zeros(0, 4, 4, dtype=torch.float64, device=device)
            sl = torch.cat(status).to(torch.int32) if status else torch.zeros(0, dtype=torch.int32, device=device)
            counts = [sum(D.shard_range(sizes[i] - first, r, world)[1] - D.shard_range(sizes[i] - first, r, world)[0] for i, first in chunk)
                      for r in range(world)]
            bmax = max(counts)
            buf = torch.zeros(bmax, 17, dtype=torch.float64, device=pl.device)
            buf[:n_local, :16] = pl.reshape(-1, 16); buf[:n_local, 16] = sl.to(torch.float64)
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)
            D.COLLECTIVES["all_gather"] += 1
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(recs, gathered, dst=0)
            problem = None
            if rank == 0:
                allrecs = [kr for part in gathered for kr in part]
                # cross-check: the gathered poses are the ones the records were built from (NaN poses -- e.g. from NaN depth -- compare
                # equal to themselves: a degenerate pair is a result, not a harness failure); the gathered status goes into the record
                by_k = {}
                for r in range(world):
                    ks = [int(starts[i]) + first + q for i, first in chunk for q in range(*D.shard_range(sizes[i] - first, r, world))]
                    for row, k in zip(out[r][:counts[r]].cpu().numpy(), ks):
                        by_k[k] = row
                for k, rec in allrecs:
                    if k not in by_k:
                        problem = f"pair {k}: a record without a gathered pose"
                        break
                    if rec.get('status') != int(by_k[k][16]):
                        problem = f"pair {k}: the gathered status differs from the status in its record"
                        break
                    if 'R_pred_44' in rec and not np.array_equal(by_k[k][:16].reshape(4, 4)[:3, :4], np.asarray(rec['R_pred_44'])[:3, :4], equal_nan=True):
                        problem = f"pair {k}: the gathered pose differs from the pose its record was built from"
                        break
                if problem is None and len(allrecs) != n_round:
                    problem = f"{len(allrecs)} records for {n_round} pairs"
            # every rank learns the verdict BEFORE anyone raises: no rank is left waiting in the next round's collective
            flag = [problem]
            dist.broadcast_object_list(flag, src=0)
            if flag[0] is not None:
                raise RuntimeError("evaluate_pairs_sharded: " + flag[0])
        else:
            allrecs = recs
            if len(allrecs) != n_round:
                raise RuntimeError(f"evaluate_pairs_sharded: {len(allrecs)} records for {n_round} pairs")
        if rank == 0:
            stats += [rec for _, rec in sorted(allrecs, key=lambda kr: kr[0])]
            if result_path is not None:
                save_results(result_path, stats)
    return stats if rank == 0 else None


def main(argv=None):
    """python -m relativepose_amd.evaluation --gpus N ...: the sharded evaluation over seeded synthetic scan pairs (no dataset ships with
    the reference) -- BASELINE configs[3]: a "val split" of --pairs pairs in global batches of --batch, sharded over N GPUs, one pose
    all_gather, one <exp>.result.npy.  Launched plainly it spawns its N ranks (one per GPU, RCCL); under torchrun it uses the launcher's
    environment.  Rank 0 prints one JSON line: pairs, seconds, per-overlap-bucket statistics (evaluation.py:321-327)."""
    import argparse
    import json
    import os
    import socket
    import sys
    import time
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dataset", default="scannet", choices=["suncg", "matterport", "scannet"])
    ap.add_argument("--pairs", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256, help="pairs per global batch (sharded over the ranks)")
    ap.add_argument("--keypoints", type=int, default=200)
    ap.add_argument("--exp", default=None, help="result file prefix: <exp>.result.npy (resumed in units of 100 pairs unless --rm)")
    ap.add_argument("--rm", action="store_true", help="ignore an existing result file (the reference's --rm)")
    ap.add_argument("--round-batches", type=int, default=None, help="global batches per gather + save round (default: all)")
    ap.add_argument("--seed", type=int, default=4000)
    ap.add_argument("--keypoint-mode", choices=["given", "reference"], default="given",
                    help="reference: every level derives its keypoints from its own feature maps like rputil.getKeypoint (synthetic SIFT detections)")
    ap.add_argument("--sift", type=int, default=120, help="--keypoint-mode reference: synthetic SIFT detections per view")
    ap.add_argument("--precision", choices=["f32", "bf16x9", "bf16x6", "f16x3", "bf16x3", "f16"], default="f32",
                    help="conv arithmetic of SCNet (SCNet.set_precision): f32 = the fp32 MFMA kernels (default), bf16x6 = what bench.py runs configs 1-3 in")
    ap.add_argument("--completion", type=int, default=1, choices=[0, 1], help="0 = the reference's 'ours_nc' method (evaluation.py:74): observed-region keypoints only")
    args = ap.parse_args(argv)

    def worker():
        import torch
        from types import SimpleNamespace
        from . import distributed as D, params, weights
        from .model import SCNet
        from .pipeline import RelativePosePipeline
        rank, world, local = D.init_from_env()
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        ds = args.dataset
        mm, S, tanh = ("kinect", 21, 0) if ds == "scannet" else ("second", 21 if ds == "matterport" else 15, 1)
        net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
        net.load_state_dict(weights.make_state_dict(7, S))
        net.set_precision(args.precision)
        pipe = RelativePosePipeline(net, ds, mm, params.final_params(ds), keypoints=args.keypoint_mode, completion=args.completion)
        batches = [SyntheticBatch(min(args.batch, args.pairs - k), args.seed + k, ds, mm, args.keypoints,
                                  sift=args.sift if args.keypoint_mode == "reference" else 0) for k in range(0, args.pairs, args.batch)]
        path = None if args.exp is None else args.exp + ".result.npy"
        D.barrier(world)
        t0 = time.perf_counter()
        n0 = D.COLLECTIVES["all_gather"]
        stats = evaluate_pairs_sharded(pipe, batches, dev, result_path=path, rank=rank, world=world, resume=not args.rm,
                                       round_batches=args.round_batches)
        torch.cuda.synchronize()
        D.barrier(world)
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"pairs": len(stats), "seconds": dt, "pairs_per_s_incl_host_rendering": len(stats) / dt, "n_gpus": world,
                              "pose_all_gathers": D.COLLECTIVES["all_gather"] - n0, "result_file": path, "dataset": ds,
                              "stats": summarize(stats)}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        import torch.multiprocessing as mp
        if torch.cuda.device_count() < args.gpus and not os.environ.get("RELPOSE_FORCE_DEVICE"):
            print(f"--gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) visible; refusing", file=sys.stderr)
            raise SystemExit(2)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_spawned_eval, args=(argv if argv is not None else sys.argv[1:], args.gpus, port), nprocs=args.gpus, join=True)
        return
    worker()


def _spawned_eval(rank, argv, world, port):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    main(argv)


if __name__ == "__main__":
    main()

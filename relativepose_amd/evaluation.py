"""Evaluation-side bookkeeping of the reference driver (evaluation.py:286-320): pose error metrics per scan
pair, the ``error_stats`` record list and its ``<exp>.result.npy`` container, and a batched harness that runs
the device pipeline over a stream of scan-pair batches and fills those records ("called
unchanged" is realised by the build's own eval_pairs-style harness).

  angular_distance_np     util.py:176-187
  pose_errors             evaluation.py:291-297
  result_record           evaluation.py:299-305
  save_results / load_results   evaluation.py:319-320 (np.save of the list of dicts)
  evaluate_pairs          evaluation.py:203-320 (loop over the loader, keypoints injected)

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


def evaluate_pairs(pipe, batches, device, result_path=None, names=None):
    """The reference's evaluation loop (evaluation.py:203-320) over an iterable of batches
    ``{"rgb","norm","depth","R" [B,2,4,4], "pts" [B,2,N,2], "ptw" [B,2,N]}`` (the DataLoader dict layout, batched, with the
    injected keypoints): R_gt = R_t inv(R_s) (:185), overlap statistics from the observed clouds (util.parse_data +
    point_cloud_overlap), the recurrent pipeline on the GPU, then one result record per pair."""
    from . import util
    stats = []
    k = 0
    for batch in batches:
        B = batch["rgb"].shape[0]
        st = pipe.prepare(batch["rgb"], batch["norm"], batch["depth"], batch["pts"], batch["ptw"], device)
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

"""ctypes binding of librelpose_hip.so (the C ABI in include/relpose.h).

The product path has no CPU fallback: if the library is missing or a call is
made without a GPU, it raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RELPOSE_LIB_PATH") or os.path.join(HERE, "librelpose_hip.so")     # (override: experiment builds, tools/build_variant.py)

c_void_p, c_int, c_int64, c_size_t, c_double, c_char_p = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_double, C.c_char_p


class Params(C.Structure):
    _fields_ = [("distThre", c_double), ("distSepThre", c_double), ("angleThre", c_double),
                ("sigmaAngle1", c_double), ("sigmaAngle2", c_double), ("sigmaDist", c_double),
                ("sigmaFeat", c_double), ("mu", c_double), ("topK", c_int), ("method", c_int)]


class Keypoints(C.Structure):
    _fields_ = [("B", c_int), ("ns_max", c_int), ("nt_max", c_int),
                ("ns", c_void_p), ("nt", c_void_p),
                ("pc_s", c_void_p), ("normal_s", c_void_p), ("feat_s", c_void_p), ("weight_s", c_void_p),
                ("pc_t", c_void_p), ("normal_t", c_void_p), ("feat_t", c_void_p), ("weight_t", c_void_p)]


class MatchDebug(C.Structure):
    _fields_ = [("wij", c_void_p), ("corres_j", c_void_p), ("corres_w", c_void_p),
                ("counts", c_void_p), ("trace", c_void_p), ("eig_iters", c_void_p)]


class ForwardArgs(C.Structure):
    """RelposeForwardArgs (include/relpose.h): the argument block of relpose_scnet_forward_ex."""
    _fields_ = [("struct_size", C.c_uint32), ("flags", c_int), ("x", c_void_p), ("out", c_void_p),
                ("n_images", c_int), ("H", c_int), ("W", c_int), ("reserved0", c_int),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("stream", c_void_p), ("tail_stream", c_void_p),
                ("self_tag", C.c_uint64), ("workspace_generation", C.c_uint64), ("reserved1", c_void_p)]


class SCNetConfig(C.Structure):
    """RelposeSCNetConfig (include/relpose.h): the reference constructor's switches, relpose_scnet_create_ex."""
    _fields_ = [("struct_size", C.c_uint32), ("snumclass", c_int), ("use_tanh", c_int), ("batchnorm", c_int), ("skip_layer", c_int),
                ("output_mask", c_int)]


class MatchArgs(C.Structure):
    """RelposeMatchArgs (include/relpose.h): the argument block of relpose_match_pairs_ex -- the per-call choices travel with the call."""
    _fields_ = [("struct_size", C.c_uint32), ("fit_cluster", c_int), ("params_host", C.POINTER(Params)), ("kp_host", C.POINTER(Keypoints)),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("max_edges", c_int64), ("pose", c_void_p), ("status", c_void_p),
                ("debug_host", C.POINTER(MatchDebug)), ("stream", c_void_p), ("affinity_kernel", c_int), ("reserved0", c_int)]


# name -> (restype, argtypes); must list every symbol declared in include/relpose.h
SIGNATURES = {
    "relpose_default_params": (None, [C.POINTER(Params)]),
    "relpose_version": (c_char_p, []),
    "relpose_set_tuning": (c_int, [c_int, c_int]),
    "relpose_match_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int64]),
    "relpose_match_pairs": (c_int, [C.POINTER(Params), C.POINTER(Keypoints), c_void_p, c_size_t, c_int64,
                                    c_void_p, c_void_p, C.POINTER(MatchDebug), c_void_p]),
    "relpose_match_pairs_ex": (c_int, [C.POINTER(MatchArgs)]),
    "relpose_affinity_topk": (c_int, [C.POINTER(Params), C.POINTER(Keypoints), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "relpose_apply_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "relpose_build_view": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_pano2pc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_warp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "relpose_warp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_warp_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_warp_pairs2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "relpose_pose_inverse": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "relpose_sample_primitives": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "relpose_get_pixel": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "relpose_interpolate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "relpose_observed_points": (c_int, [c_int, c_int]),
    "relpose_depth2pc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_depth2pc_full": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_nn_dist": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "relpose_feature_distance_map": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "relpose_nms_sampling": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "relpose_keypoints_reference_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "relpose_keypoints_reference": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "relpose_scnet_create": (c_void_p, [c_int, c_int]),
    "relpose_scnet_create_ex": (c_void_p, [C.POINTER(SCNetConfig)]),
    "relpose_scnet_destroy": (None, [c_void_p]),
    "relpose_scnet_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "relpose_scnet_finalize": (c_int, [c_void_p]),
    "relpose_scnet_set_precision": (c_int, [c_void_p, c_int]),
    "relpose_scnet_num_params": (c_int64, [c_void_p]),
    "relpose_scnet_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "relpose_scnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "relpose_scnet_forward_ex": (c_int, [c_void_p, C.POINTER(ForwardArgs)]),
    "relpose_scnet_plan_macs": (c_int, [c_void_p, c_int, c_int, c_int, C.POINTER(c_double)]),
    "relpose_scnet_read_tap": (c_int64, [c_void_p, c_char_p, c_void_p, c_void_p, c_void_p]),
    "relpose_scnet_profile": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_int,
                                      C.POINTER(c_double), C.POINTER(c_double), C.POINTER(c_int64), c_void_p]),
}

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            _build_locked()          # a fresh checkout on a box with hipcc: compile the HIP library once (no CPU fallback exists)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -m relativepose_amd.build` "
                               "(or __graft_entry__.build()). There is no CPU fallback.")
        # torch ships its own libamdhip64; import it first so this library binds to the SAME HIP runtime
        # (loading ours first pulls in /opt/rocm's copy and the second runtime then sees no device).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        missing = [name for name in SIGNATURES if not hasattr(l, name)]
        if missing:
            raise RuntimeError(f"{LIB_PATH} is stale, missing symbols: {missing}")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _build_locked():
    """Compile librelpose_hip.so with hipcc if it is missing; one process builds, concurrent ranks wait on the lock."""
    import fcntl
    from . import build as _b
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        return
    with open(LIB_PATH + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                _b.build(verbose=False)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


# relpose_set_tuning keys (include/relpose.h RELPOSE_TUNE_*)
TUNE_KEYS = {"affinity_kernel": 0, "fit_max_products": 1, "fit_cluster": 2, "fit_global_vectors": 3, "fit_fixed_checks": 4}
AFFINITY_KERNELS = {"auto": 0, "rows": 1, "tile": 2, "lds": 3, "pool": 4}


class tuning:
    """``with tuning(affinity_kernel="tile"): ...`` -- force a kernel variant for the calls inside (process-wide knob of the
    library, restored on exit).  Every variant produces the same results; the parity tests use this to check each one."""

    def __init__(self, **kw):
        self.kw = {k: (AFFINITY_KERNELS[v] if k == "affinity_kernel" and isinstance(v, str) else int(v)) for k, v in kw.items()}
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = lib().relpose_set_tuning(TUNE_KEYS[k], v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            lib().relpose_set_tuning(TUNE_KEYS[k], v)
        return False


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (HIP error %d)" % (-rc - 1000) if rc <= -1000 else ""))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("relativepose_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

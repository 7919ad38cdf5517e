"""CPU: the C-ABI library loads without a GPU and exports every symbol that
include/relpose.h declares; argument validation paths return RELPOSE_EINVAL
without touching the device; host-only entry points work."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from relativepose_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.lib()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "relpose.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef RP_EXPERIMENTS.*?#endif", "", src, flags=re.S)      # (lost A/Bs of the experiment log: not in the product ABI)
    return sorted(set(re.findall(r"\b(relpose_[a-z0-9_]+)\s*\(", src)))


def test_every_header_symbol_is_exported_and_bound(lib):
    from relativepose_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in relpose.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms


def test_host_only_entry_points(lib):
    from relativepose_amd import _lib
    p = _lib.Params()
    lib.relpose_default_params(C.byref(p))
    assert (p.topK, p.method) == (5, 0) and abs(p.distSepThre - 0.12) < 1e-15 and abs(p.mu - 0.3) < 1e-15
    assert b"gfx950" in lib.relpose_version()
    assert lib.relpose_match_workspace_bytes(32, 200, 200, 5, 0) > 0
    assert lib.relpose_match_workspace_bytes(0, 200, 200, 5, 0) == 0
    assert lib.relpose_match_workspace_bytes(1, 200, 200, 9, 0) == 0        # topK > 8
    assert lib.relpose_warp_workspace_bytes(2, 160) >= 2 * 160 * 640 * 4


def test_invalid_arguments_are_rejected_without_a_device(lib):
    assert lib.relpose_apply_mask(None, None, 1, 7, 160, 0, None) == -1
    assert lib.relpose_warp(None, None, None, None, 1, 160, 0, None) == -1
    assert lib.relpose_pano2pc(None, None, None, 1, 160, 7, None) == -1
    assert lib.relpose_match_pairs(None, None, None, 0, 0, None, None, None, None) == -1
    assert lib.relpose_scnet_forward(None, None, None, 2, 160, 640, None, 0, None) == -1
    assert not lib.relpose_scnet_create(0, 1)
    h = lib.relpose_scnet_create(15, 1)
    assert h and lib.relpose_scnet_num_params(h) == 0
    assert lib.relpose_scnet_workspace_bytes(h, 2, 160, 640) == 0           # not finalised
    lib.relpose_scnet_destroy(h)


def test_product_path_has_no_cpu_fallback():
    import numpy as np
    import torch
    from relativepose_amd import rpmodule, synth
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    S, T, _ = synth.make_match_case(10, 0)
    with pytest.raises(RuntimeError):
        rpmodule.RelativePoseEstimation_helper(S, T, rpmodule.opts())
    with pytest.raises(RuntimeError):
        from relativepose_amd import util
        util.warping(np.zeros((1, 8, 160, 640), np.float32), np.eye(4), "suncg")


def test_precision_codes_match_the_header():
    """SCNet.set_precision's mode names map to the RELPOSE_PREC_* enumerators of include/relpose.h."""
    import re
    from relativepose_amd import model
    hdr = open(os.path.join(ROOT, "include", "relpose.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(RELPOSE_PREC_[A-Z0-9]+)\s*=\s*(\d+)", hdr))
    want = {"f32": "RELPOSE_PREC_F32", "bf16x3": "RELPOSE_PREC_BF16X3", "f16x3": "RELPOSE_PREC_F16X3", "f16": "RELPOSE_PREC_F16",
            "bf16x9": "RELPOSE_PREC_BF16X9", "bf16x6": "RELPOSE_PREC_BF16X6"}
    assert set(model.PRECISION_CODES) == set(want)
    for name, sym in want.items():
        assert enum[sym] == model.PRECISION_CODES[name], (name, sym)


def test_capacity_limit_matches_the_header():
    import re
    from relativepose_amd import rpmodule
    hdr = open(os.path.join(ROOT, "include", "relpose.h")).read()
    assert int(re.search(r"#define RELPOSE_MAX_CORRESPONDENCES\s+(\d+)", hdr).group(1)) == rpmodule.MAX_CORRESPONDENCES
    assert int(re.search(r"#define RELPOSE_MAX_TARGETS\s+(\d+)", hdr).group(1)) == rpmodule.MAX_TARGETS
    from relativepose_amd import rputil
    assert int(re.search(r"#define RELPOSE_KP_MAX_QUERIES_PER_VIEW\s+(\d+)", hdr).group(1)) == rputil.MAX_QUERIES_PER_VIEW

"""GPU: batched sigma-tuning objective vs the per-primitive oracle loop, and the tuning step against the reference's
own run of trainRelativePoseModuleRecFD.py:215-298 (tests/golden/tune.npz)."""
import numpy as np
import pytest

from oracle import tune_oracle
from relativepose_amd import synth

pytestmark = pytest.mark.gpu


def _prims(n=6, N=60):
    out = []
    for i in range(n):
        S, T, G = synth.make_match_case(N + 7 * i, 300 + i, inlier=0.5 + 0.05 * i, noise=0.01)
        out.append({'pc_src': S['pc'], 'normal_src': S['normal'], 'feat_src': S['feat'], 'weight_src': S['weight'],
                    'pc_tgt': T['pc'], 'normal_tgt': T['normal'], 'feat_tgt': T['feat'], 'weight_tgt': T['weight'], 'R_gt': G})
    return out


def test_objective_matches_oracle_loop():
    from relativepose_amd import tune
    prims = _prims()
    for sig in ([0.2615, 0.2615, 0.04, 0.01], [0.3, 0.2, 0.03, 0.012]):
        para = tune.make_para(sig)
        lg, ag = tune.objective(prims, para)
        lo, ao = tune_oracle.objective(prims, para)
        assert abs(lg - lo) < 1e-9 * max(1, abs(lo)) and abs(ag - ao) < 1e-6


def test_tune_step_equals_the_reference_script(golden_dir):
    """Two outer iterations of the tuning loop with the batched GPU objective against what the reference's own lines produced on the
    same primitives and random stream: identical probes, loss differences to 1e-10, the accepted sigmas to 1e-9 relative."""
    from relativepose_amd import tune
    from test_tune_cpu import run_against_golden
    run_against_golden(golden_dir, tune.objective)

"""CPU: relativepose_amd.tune.tune_step -- the probe sequence, the least-squares gradient, the normalised step and the halving line
search of the reference's sigma tuning -- against the reference's OWN lines run on the same primitives (tests/golden/tune.npz:
make_golden.gen_tune execs trainRelativePoseModuleRecFD.py:215-298 in a prepared namespace), with the objective supplied by the numpy
oracle so that no GPU is needed (tests/test_gpu_tune.py runs the same comparison with the batched GPU objective)."""
import os

import numpy as np

from cases import TUNE_CASE
from oracle import tune_oracle
from relativepose_amd import synth, tune


def run_against_golden(golden_dir, objective_fn, prims=None, rtol_sigma=1e-9, atol_loss=1e-12):
    g = np.load(os.path.join(golden_dir, "tune.npz"))
    if prims is None:
        prims = synth.make_tune_primitives(TUNE_CASE["n_prims"], TUNE_CASE["N"], TUNE_CASE["seed0"])
    rng = np.random.RandomState(TUNE_CASE["np_seed"])          # the reference draws from the global stream seeded the same way
    sig = g["sigma_init"]
    assert np.allclose(sig, [0.523 / 2, 0.523 / 2, 0.08 / 2, 0.01], rtol=0, atol=0)
    for it in range(TUNE_CASE["iters"]):
        info = {}
        sig, loss, ad, found = tune.tune_step(prims, sig, rng, n_probe=10, objective_fn=objective_fn, info=info)
        assert np.array_equal(info["eps"][1:], g["eps"][it]), "probe perturbations: not the reference's random stream"
        assert np.allclose(info["losses"][1:] - info["losses"][0], g["dloss"][it], rtol=0, atol=1e-10)
        assert found                                               # (both iterations of the fixture found a descent step)
        assert np.allclose(sig, g["log"][it, 2:], rtol=rtol_sigma, atol=0), (it, sig, g["log"][it, 2:])
        assert abs(loss - g["log"][it, 0]) < atol_loss and abs(ad - g["log"][it, 1]) < 1e-8
    assert bool(g["found_last"])


def test_tune_step_equals_the_reference_script(golden_dir):
    run_against_golden(golden_dir, tune_oracle.objective)

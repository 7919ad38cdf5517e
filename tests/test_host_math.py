"""CPU: the scalar math the HIP matcher kernels use (csrc/rp_math.h) compiled
with g++ and checked against the numpy oracle -- lets the pair tests / Horn
solver be validated in a container without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import rp_oracle as M
from relativepose_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r'''
#include "rp_math.h"
extern "C" {
void t_pair(const double* a, const double* b, const double* k9, double* out) {
    RpPairConsts k; k.dist_thre2=k9[0]; k.sep_thre=k9[1]; k.angle_thre2=k9[2]; k.two_sd2=k9[3]; k.two_sa1_2=k9[4];
    k.two_sa2_2=k9[5]; k.den_both=k9[6]; k.den_other=k9[7]; k.mu=k9[8];
    RpPairEval e = rp_pair_eval(a, a+3, a+6, a+9, b, b+3, b+6, b+9, k);
    out[0]=e.d; out[1]=e.alpha; out[2]=e.beta; out[3]=e.gamma; out[4]=e.pass_dist; out[5]=e.pass_all;
    out[6]=rp_pair_weight(e, a[12], b[12], a[13], b[13], a[14], b[14], k);
}
void t_horn(const double* M, double* R) { double m[3][3], r[3][3]; for(int i=0;i<9;++i) m[i/3][i%3]=M[i];
    rp_horn_rotation(m, r); for(int i=0;i<9;++i) R[i]=r[i/3][i%3]; }
int t_inv4(const double* A, double* o) { return rp_inv4(A, o) ? 1 : 0; }
}
'''


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostmath")
    src = d / "shim.cpp"
    src.write_text(SHIM)
    so = d / "shim.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "relativepose_amd", "csrc"), str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_pair_eval_matches_oracle(shim):
    S, T, _ = synth.make_match_case(40, 5)
    p = M.Params()
    d = {}
    M.relative_pose_helper(S, T, p, d)
    corres, wij, pc = d["corres"], d["wij"], d["pairs"]
    k9 = np.array([p.distThre ** 2, 1.5 * p.distSepThre ** 2, p.angleThre ** 2, 2 * p.sigmaDist ** 2, 2 * p.sigmaAngle1 ** 2,
                   2 * p.sigmaAngle2 ** 2, 0, 0, p.mu])
    surv = {(int(a), int(b)): w for a, b, w in zip(pc["c1"], pc["c2"], pc["w"])}
    n_pass = 0
    rs = np.random.RandomState(0)
    C_ = corres.shape[1]
    pairs = list(surv.keys())[:200] + [tuple(sorted(rs.choice(C_, 2, replace=False))) for _ in range(2000)]
    for c1, c2 in pairs:
        def pack(c):
            i, j = corres[0, c], corres[1, c]
            return np.concatenate((S["pc"][i], S["normal"][i], T["pc"][j], T["normal"][j], [wij[i, j], S["weight"][i], T["weight"][j]]))
        out = np.zeros(7)
        shim.t_pair(_dp(pack(c1)), _dp(pack(c2)), _dp(k9), _dp(out))
        if (c1, c2) in surv:
            assert out[5] == 1
            assert out[6] == surv[(c1, c2)] or abs(out[6] - surv[(c1, c2)]) <= 1e-14 * abs(surv[(c1, c2)])
            n_pass += 1
        else:
            assert out[5] == 0
    assert n_pass >= 100


def test_horn_matches_oracle(shim):
    rs = np.random.RandomState(1)
    for _ in range(50):
        src, tgt, w = rs.randn(3, 30), rs.randn(3, 30), rs.rand(30)
        Mx = np.ascontiguousarray(src @ (tgt * w[None]).T)
        R = np.zeros(9)
        shim.t_horn(_dp(Mx), _dp(R))
        assert np.abs(R.reshape(3, 3) - M.horn87(src, tgt, w)).max() < 1e-12


def test_inv4(shim):
    rs = np.random.RandomState(2)
    for _ in range(20):
        A = np.ascontiguousarray(synth.random_rigid(rs))
        o = np.zeros(16)
        assert shim.t_inv4(_dp(A), _dp(o)) == 1
        assert np.allclose(o.reshape(4, 4), np.linalg.inv(A), atol=1e-14)

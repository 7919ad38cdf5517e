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
void t_horn_fast(const double* M, double* R) { double m[3][3], r[3][3]; for(int i=0;i<9;++i) m[i/3][i%3]=M[i];
    rp_horn_rotation_fast(m, r); for(int i=0;i<9;++i) R[i]=r[i/3][i%3]; }
void t_div100(const float* x, float* out, int* ok, int n) { for (int i = 0; i < n; ++i) { out[i] = rp_div100_fast(x[i]); ok[i] = rp_div100_ok(x[i]) ? 1 : 0; } }
void t_chunks(int nseg, int G, int* out) { const int c = rp_fit_chunk_size(nseg, G); out[0] = c; out[1] = rp_fit_chunk_count(nseg, c);
    for (int k = 0; k <= out[1] && k < 62; ++k) out[2 + k] = rp_fit_chunk_begin(k, c); }
double t_lz_rate(double ra, int ma, double rb, int mb, double prev) { return rp_lz_rate(ra, ma, rb, mb, prev); }
int t_lz_steps(double r, double lrate, double tol, int mx) { return rp_lz_steps_to_check(r, lrate, tol, mx); }
int t_eig4_fast(const double* N, double* q) { double n[4][4]; for(int i=0;i<16;++i) n[i/4][i%4]=N[i]; return rp_sym4_max_eigvec_fast(n, q); }
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


def test_horn_fast_path_matches_jacobi_and_oracle(shim):
    """rp_horn_rotation_fast (Newton on the characteristic quartic + adjugate eigenvector; what the single-workgroup fit
    uses) against the Jacobi solver and the numpy oracle: random weighted covariances, nearly planar / nearly collinear
    point sets, pure rotations (rank-deficient N - lambda I), tiny and huge scales, and nearly degenerate leading
    eigenvalues (where it must fall back to Jacobi)."""
    rs = np.random.RandomState(7)
    worst, n_fast = 0.0, 0
    cases = []
    for _ in range(200):
        n = rs.randint(3, 40)
        src, tgt, w = rs.randn(3, n), rs.randn(3, n), rs.rand(n)
        cases.append(src @ (tgt * w[None]).T)
    for _ in range(100):                                   # a true rigid motion + noise (what IRLS sees near convergence)
        n = rs.randint(4, 60)
        src = rs.randn(3, n) * rs.uniform(0.1, 3)
        T = synth.random_rigid(rs)
        tgt = T[:3, :3] @ src + rs.randn(3, n) * 10 ** rs.uniform(-9, -1)
        cases.append(src @ (tgt * rs.rand(n)[None]).T)
    for _ in range(50):                                    # planar and collinear clouds
        n = 20
        src = rs.randn(3, n)
        src[2] *= 10 ** rs.uniform(-12, -2)
        if rs.rand() < 0.5:
            src[1] *= 10 ** rs.uniform(-12, -2)
        T = synth.random_rigid(rs)
        cases.append(src @ ((T[:3, :3] @ src) * rs.rand(n)[None]).T)
    for sc in (1e-30, 1e-8, 1e8, 1e30):
        cases.append(cases[3] * sc)
    for Mx in cases:
        Mx = np.ascontiguousarray(Mx)
        R0, R1 = np.zeros(9), np.zeros(9)
        shim.t_horn(_dp(Mx), _dp(R0))
        shim.t_horn_fast(_dp(Mx), _dp(R1))
        # conditioning of the leading eigenvector: compare through the oracle's eigen-decomposition
        Nm = np.array([[Mx[0, 0] + Mx[1, 1] + Mx[2, 2], Mx[1, 2] - Mx[2, 1], Mx[2, 0] - Mx[0, 2], Mx[0, 1] - Mx[1, 0]],
                       [Mx[1, 2] - Mx[2, 1], Mx[0, 0] - Mx[1, 1] - Mx[2, 2], Mx[0, 1] + Mx[1, 0], Mx[0, 2] + Mx[2, 0]],
                       [Mx[2, 0] - Mx[0, 2], Mx[0, 1] + Mx[1, 0], Mx[1, 1] - Mx[0, 0] - Mx[2, 2], Mx[1, 2] + Mx[2, 1]],
                       [Mx[0, 1] - Mx[1, 0], Mx[2, 0] + Mx[0, 2], Mx[1, 2] + Mx[2, 1], Mx[2, 2] - Mx[0, 0] - Mx[1, 1]]])
        ev = np.linalg.eigvalsh(Nm)
        gap = (ev[3] - ev[2]) / max(np.abs(ev).max(), 1e-300)
        q = np.zeros(4)
        fast = shim.t_eig4_fast(_dp(np.ascontiguousarray(Nm)), _dp(q))
        n_fast += fast
        err = np.abs(R0 - R1).max()
        assert err < 1e-11 / max(gap, 1e-6), (err, gap, fast)
        if gap > 1e-3:
            worst = max(worst, err)
    assert worst < 5e-13 and n_fast > 250


def test_div100_fast_equals_float32_division(shim):
    """rp_div100_fast (the affinity kernels' feat / 100): bit-equal to numpy's float32 division wherever rp_div100_ok says so --
    random bit patterns over the whole float32 range, the range edges, zeros, and a dense sweep around typical descriptor values."""
    rs = np.random.RandomState(0)
    bits = rs.randint(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    x = np.concatenate([bits.view(np.float32), np.float32([0.0, -0.0, 1e-30, 1.0000001e-30, 9.999999e29, 1e30, 100.0, -100.0, 1.0, 3.0]),
                        np.tanh(rs.randn(500_000)).astype(np.float32), rs.uniform(-50, 50, 500_000).astype(np.float32)])
    x = np.ascontiguousarray(x[np.isfinite(x)])
    out = np.empty_like(x)
    ok = np.empty(len(x), np.int32)
    shim.t_div100(x.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), ok.ctypes.data_as(C.POINTER(C.c_int)),
                  len(x))
    ref = x / np.float32(100.0)
    m = ok == 1
    assert m.sum() > 2_000_000
    assert np.array_equal(out[m].view(np.uint32), ref[m].view(np.uint32))
    assert not ok[np.abs(x) >= 1e30].any() and not ok[(np.abs(x) <= 1e-30) & (x != 0)].any()


def test_fit_chunk_geometry_covers_every_segment_once(shim):
    """Chunks of the fit's distributed products (rp_fit_chunk_*): for every segment count and cluster size the chunks tile [0, nseg)
    exactly -- consecutive, disjoint, on 64-segment (128-byte) boundaries, the leader's chunk 0 twice the size of the others, at most
    one chunk per workgroup once there are enough segments -- so a partial sum is written by exactly one claim."""
    out = (C.c_int * 64)()
    rng = np.random.default_rng(3)
    cases = [(n, g) for g in range(2, 9) for n in (1, 63, 64, 65, 127, 128, 129, 500, 1000, 4999, 5000, 5001, 32768)]
    cases += [(int(n), int(g)) for n, g in zip(rng.integers(1, 40000, 300), rng.integers(2, 9, 300))]
    for nseg, G in cases:
        shim.t_chunks(nseg, G, out)
        csz, cnt = out[0], out[1]
        assert csz % 64 == 0 and csz >= 64 and 1 <= cnt <= 60, (nseg, G, csz, cnt)
        begins = [out[2 + k] for k in range(cnt + 1)]
        assert begins[0] == 0 and begins[1] == 2 * csz and all(b - a == csz for a, b in zip(begins[1:], begins[2:]))
        assert begins[cnt - 1] < nseg or cnt == 1            # the last chunk is not empty ...
        assert begins[cnt] >= nseg                            # ... and reaches the end (the kernel clamps it to nseg)
        if nseg >= 64 * (G + 1):
            assert cnt <= G, (nseg, G, csz, cnt)              # everybody there: one chunk each


def _lanczos_products(A, v0, shim, adaptive, lrate, tol=1e-13, basis=24, every=8):
    """numpy model of csrc/matcher.hip lanczos_top (full re-orthogonalisation, restart from the Ritz vector at `basis` steps) with the
    convergence tests placed by rp_lz_steps_to_check / rp_lz_rate (adaptive) or every 8th step; returns (eigenvector, products, tests, lrate)."""
    nprod = ntest = 0
    vec = v0.copy()
    while True:
        V, al, be = [vec], [], []
        conv = False
        nxt, m_a, r_a = every, 0, 0.0
        for j in range(basis):
            y = A @ V[j]
            nprod += 1
            alpha = 0.0
            for _ in range(2):
                c = np.array([v @ y for v in V])
                nb = y @ y
                y = y - sum(ci * v for ci, v in zip(c, V))
                alpha += c[j]
                if y @ y > 0.25 * nb:
                    break
            beta = float(np.sqrt(y @ y))
            al.append(alpha)
            be.append(beta)
            m = j + 1
            inv = not (beta > 1e-14 * (abs(alpha) + beta))
            if not inv and m < basis:
                V.append(y / beta)
            if j == 0 and adaptive:
                r_a, m_a = beta / abs(alpha), 1
                nxt = 1 if r_a <= tol else 1 + shim.t_lz_steps(r_a, lrate, tol, every - 1)
            if inv or m == basis or m == nxt:
                ntest += 1
                T = np.diag(al) + np.diag(be[:m - 1], 1) + np.diag(be[:m - 1], -1)
                w, S = np.linalg.eigh(T)
                th, sv = w[-1], S[:, -1]
                resid = 0.0 if inv else beta * abs(sv[m - 1])
                if inv or m == basis or resid <= tol * abs(th):
                    conv = resid <= tol * abs(th)
                    break
                if adaptive:
                    r_b = resid / abs(th)
                    lrate = shim.t_lz_rate(r_a, m_a, r_b, m, lrate)
                    r_a, m_a = r_b, m
                    nxt = m + shim.t_lz_steps(r_b, lrate, tol, every)
                else:
                    nxt = m + every
        vec = sum(si * v for si, v in zip(sv, V[:m]))
        vec /= np.linalg.norm(vec)
        if conv or nprod > 190:
            return vec, nprod, ntest, lrate


def test_lanczos_check_placement_never_costs_products(shim):
    """rp_lz_steps_to_check / rp_lz_rate (the placement of the eigen-solve's convergence tests): on sequences of slowly changing
    nonnegative sparse matrices with small and large spectral gaps (warm-started like the fit's five rounds) the predicted placement
    converges to the same eigenvector as the test-every-8th-step rule, never with more products, with clearly fewer in total, and
    with a bounded number of extra tests.  Edge cases of the two functions: no information -> the fixed interval; converged -> test now."""
    shim.t_lz_rate.restype = C.c_double
    shim.t_lz_rate.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.c_double]
    shim.t_lz_steps.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    assert shim.t_lz_steps(1e-3, 0.0, 1e-13, 8) == 8 and shim.t_lz_steps(float("nan"), -1.0, 1e-13, 8) == 8
    assert shim.t_lz_steps(float("inf"), -1.0, 1e-13, 8) == 8 and shim.t_lz_steps(1e-14, -1.0, 1e-13, 8) == 1
    assert shim.t_lz_steps(1e-3, np.log(0.1), 1e-13, 8) == 8 and shim.t_lz_steps(1e-10, np.log(0.1), 1e-13, 8) == 2
    assert shim.t_lz_steps(1e-12, np.log(0.5), 1e-13, 8) == 2 and shim.t_lz_steps(0.5, -0.05, 1e-13, 8) == 8
    assert shim.t_lz_rate(1e-2, 3, 1e-5, 6, -9.0) == pytest.approx(np.log(0.1)) and shim.t_lz_rate(1e-2, 3, 1e-1, 6, -9.0) == -9.0
    assert shim.t_lz_rate(1e-2, 3, 1e-2 * 0.999, 4, 0.0) == -0.05 and shim.t_lz_rate(1.0, 1, 1e-300, 2, 0.0) == -12.0
    assert shim.t_lz_rate(float("inf"), 1, 1e-3, 4, -2.0) == -2.0 and shim.t_lz_rate(1e-2, 4, 1e-3, 4, -2.0) == -2.0
    rng = np.random.default_rng(5)
    tot = {False: [0, 0], True: [0, 0]}
    for case in range(12):
        n = int(rng.integers(150, 600))
        dens = float(rng.choice([0.02, 0.06, 0.2]))
        B = np.triu((rng.random((n, n)) < dens) * rng.random((n, n)), 1)
        k = int(rng.integers(8, n // 3))
        B[:k, :k] += np.triu(rng.random((k, k)) * float(rng.choice([0.3, 1.0, 3.0])), 1)        # a planted consistent cluster: the gap
        B = B + B.T
        h = 1.0 + rng.random(n)
        for adaptive in (False, True):
            u = np.full(n, 1.0 / np.sqrt(n))
            lrate, hh, us = 0.0, h.copy(), []
            rs = np.random.default_rng(100 + case)
            for rnd in range(5):
                A = B * (hh[:, None] + hh[None, :])
                u, nprod, ntest, lrate = _lanczos_products(A, u, shim, adaptive, lrate)
                assert np.linalg.norm(A @ u - (u @ A @ u) * u) <= 1e-11 * abs(u @ A @ u), (case, adaptive, rnd)
                tot[adaptive][0] += nprod
                tot[adaptive][1] += ntest
                us.append((u.copy(), nprod))
                hh = hh * (1.0 + 0.05 * rs.standard_normal(n) / (1 + rnd))                          # the next round's reweighting
            if adaptive:
                for (ua, pa), (uf, pf) in zip(us, fixed_us):
                    assert min(np.linalg.norm(ua - uf), np.linalg.norm(ua + uf)) < 1e-9
                    assert pa <= pf, (case, pa, pf)
            else:
                fixed_us = us
    assert tot[True][0] < 0.9 * tot[False][0], tot                    # >= 10 % fewer products ...
    assert tot[True][1] <= 1.8 * tot[False][1] + 5, tot               # ... for a bounded number of extra tests (a test costs 0.3-0.5 of a product step)

// Scalar double-precision math shared by the matcher kernels.  Everything here is
// __host__ __device__ so tests/test_host_math.py can compile it with g++ and check
// it against the numpy oracle on a box without a GPU.
//
// Operation order follows the numpy expressions of the reference
// (RPModule/rpmodule.py:399-467 pair tests and weights, :17-58 Horn) so that
// thresholded decisions agree with the float64 reference; this file is compiled
// with -ffp-contract=off (no FMA fusion).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD inline
#endif

// RN(x / 100) in float32 (numpy's feat / 100, rpmodule.py:342-343) by Markstein's correction of x * RN(1/100): three operations
// instead of the hardware division sequence.  Exhaustively equal to x / 100.0f for every float with 1e-30 < |x| < 1e30 and for
// +-0 (all 3.3e9 of them were compared on the host; tests/test_host_math.py keeps a sample); rp_div100_ok says whether x is one.
RP_HD float rp_div100_fast(float x) {
    const float q0 = x * 0.01f;
    const float rem = fmaf(-q0, 100.0f, x);
    return copysignf(fmaf(rem, 0.01f, q0), x);
}
RP_HD bool rp_div100_ok(float x) { const float ax = fabsf(x); return (ax < 1e30f) && (ax > 1e-30f || ax == 0.0f); }

struct RpPairConsts {       // derived on the host in double, exactly as numpy does
    double dist_thre2;      // np.power(distThre, 2)
    double sep_thre;        // 1.5 * np.power(distSepThre, 2)   (sic: a distance vs a squared threshold)
    double angle_thre2;     // np.power(angleThre, 2)
    double two_sd2;         // 2 * sigmaDist**2
    double two_sa1_2;       // 2 * sigmaAngle1**2
    double two_sa2_2;       // 2 * sigmaAngle2**2
    double den_both;        // 2 * np.power((sigmaFeat/1.2)/5, 2)   both keypoints observed
    double den_other;       // 2 * np.power(sigmaFeat/5, 2)
    double mu;
};

// Chunks of the fit's distributed matrix-vector products (matcher.hip, fit_work_loop) over nseg segments and G workgroups per scan
// pair: chunk 0 is the LEADER's (never claimed: it starts on it the moment the product is published, some microseconds before a helper
// has seen the control word and loaded the vector) and twice as long as the others, which are claimed in order; sizes are whole
// 128-byte lines of partial sums (multiples of 64 segments).  With everybody there, every workgroup does one chunk.
RP_HD int rp_fit_chunk_size(int nseg, int G) { const int c = ((nseg + G) / (G + 1) + 63) & ~63; return c < 64 ? 64 : c; }
RP_HD int rp_fit_chunk_count(int nseg, int csz) { const int r = (nseg - 2 * csz + csz - 1) / csz; return 1 + (r < 0 ? 0 : r); }
RP_HD int rp_fit_chunk_begin(int ch, int csz) { return ch == 0 ? 0 : (ch + 1) * csz; }

// Placement of the Lanczos convergence tests (matcher.hip, lanczos_top).  A test (tridiagonal eigen-solve on one wave) costs about
// half a {product, re-orthogonalisation} step and a step taken after convergence a whole one, so instead of testing every 8th step
// (3.5 wasted steps per eigen-solve on average) the next test goes where the relative residual estimate r = beta_m |s_m| / |theta| is
// predicted to cross the tolerance, assuming geometric decay at the rate between the last two data points (the first data point,
// after step 1, is free: beta_1 / |alpha_1|; the rate carries over from the previous eigen-solve of the same scan pair, whose matrix
// differs only by the reweighting).  Convergence accelerates, so the prediction errs late: 0.85 of the predicted distance.
//   rp_lz_rate: log of the decay per step from (m_a, r_a) -> (m_b, r_b), clamped to [-12, -0.05]; `prev` when the points carry no information
//   rp_lz_steps_to_check: steps from the last data point to the next test, 1..max_steps (lrate >= 0: unknown -> max_steps)
RP_HD double rp_lz_rate(double r_a, int m_a, double r_b, int m_b, double prev) {
    if (!(r_a > 0.0) || !(r_b > 0.0) || !(r_b < r_a) || !(r_a < 1e300) || m_b <= m_a) return prev;
    const double lr = log(r_b / r_a) / (double)(m_b - m_a);
    return lr > -0.05 ? -0.05 : (lr < -12.0 ? -12.0 : lr);
}
RP_HD int rp_lz_steps_to_check(double r, double lrate, double tol, int max_steps) {
    if (!(lrate < 0.0) || !(r > 0.0) || !(r < 1e300)) return max_steps;
    if (!(r > tol)) return 1;
    const double n = 0.85 * (log(tol / r) / lrate);
    if (!(n < (double)max_steps)) return max_steps;
    const int k = (int)n;
    return k < 1 ? 1 : k;
}

RP_HD double rp_norm3(double x, double y, double z) { return sqrt((x * x + y * y) + z * z); }
RP_HD double rp_dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
RP_HD double rp_clip1(double v) { return v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v); }   // NaN stays NaN

struct RpPairEval { double d, alpha, beta, gamma; int pass_dist, pass_all; };

// One correspondence pair: "1" = (ps1,ns1 -> pt1,nt1), "2" = (ps2,ns2 -> pt2,nt2).
// rpmodule.py:399-404 (distance test) and :424-436 (angle test).
// The distance test alone (rpmodule.py:399-404): edge vectors, their lengths, d = (|es| - |et|)^2.  rp_pair_eval starts with
// exactly this, so a kernel that screens pairs with rp_pair_dist and evaluates the survivors with rp_pair_eval sees the same bits.
RP_HD bool rp_pair_dist(const double* ps1, const double* pt1, const double* ps2, const double* pt2, const RpPairConsts& k,
                        double* es, double* et, double& dis_s, double& dis_t, double& d) {
    es[0] = ps1[0] - ps2[0]; es[1] = ps1[1] - ps2[1]; es[2] = ps1[2] - ps2[2];
    et[0] = pt1[0] - pt2[0]; et[1] = pt1[1] - pt2[1]; et[2] = pt1[2] - pt2[2];
    dis_s = rp_norm3(es[0], es[1], es[2]);
    dis_t = rp_norm3(et[0], et[1], et[2]);
    const double dd = dis_s - dis_t;
    d = dd * dd;
    const double mn = dis_s < dis_t ? dis_s : dis_t;     // np.minimum (NaN-propagation irrelevant: compare below is false)
    return (d < k.dist_thre2) && (mn > k.sep_thre);
}

RP_HD RpPairEval rp_pair_eval(const double* ps1, const double* ns1, const double* pt1, const double* nt1,
                              const double* ps2, const double* ns2, const double* pt2, const double* nt2,
                              const RpPairConsts& k) {
    RpPairEval r;
    double es[3], et[3], dis_s, dis_t;
    r.pass_dist = rp_pair_dist(ps1, pt1, ps2, pt2, k, es, et, dis_s, dis_t, r.d);
    r.alpha = r.beta = r.gamma = 0.0;
    r.pass_all = 0;
    if (!r.pass_dist) return r;
    for (int a = 0; a < 3; ++a) { es[a] = es[a] / dis_s; et[a] = et[a] / dis_t; }
    double a0 = acos(rp_clip1(rp_dot3(ns1, ns2))) - acos(rp_clip1(rp_dot3(nt1, nt2)));
    double b0 = acos(rp_clip1(rp_dot3(ns1, es))) - acos(rp_clip1(rp_dot3(nt1, et)));
    double g0 = acos(rp_clip1(rp_dot3(ns2, es))) - acos(rp_clip1(rp_dot3(nt2, et)));
    r.alpha = a0 * a0; r.beta = b0 * b0; r.gamma = g0 * g0;
    r.pass_all = (r.alpha < k.angle_thre2) && (r.beta < k.angle_thre2) && (r.gamma < k.angle_thre2);
    return r;
}

// rpmodule.py:457-467: f1*f2*exp(-d/2sd^2 - a/2sa1^2 - b/2sa2^2 - g/2sa2^2), x0.6 unless all four observed.
RP_HD double rp_pair_weight(const RpPairEval& e, double f1, double f2, double ws1, double ws2, double wt1, double wt2,
                            const RpPairConsts& k) {
    double ex = exp(((-e.d / k.two_sd2 - e.alpha / k.two_sa1_2) - e.beta / k.two_sa2_2) - e.gamma / k.two_sa2_2);
    double w = (f1 * f2) * ex;
    double ww = ((ws1 * ws2) * wt1) * wt2;
    if (ww != 1.0) w *= 0.6;
    return w;
}

// Leading eigenvector of a symmetric 4x4 (cyclic Jacobi), the quaternion of Horn's
// method (rpmodule.py:46-53 uses np.linalg.eig + argmax).  N is destroyed.
// Every index is a compile-time constant after unrolling so N and V stay in registers on the GPU.
#define RP_JROT(P, Q)                                                                          \
    {                                                                                          \
        const double apq = N[P][Q];                                                            \
        if (apq != 0.0) {                                                                      \
            const double theta = (N[Q][Q] - N[P][P]) / (2.0 * apq);                            \
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                               \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                   \
                const double nkp = N[k][P], nkq = N[k][Q];                                     \
                N[k][P] = c * nkp - s * nkq; N[k][Q] = s * nkp + c * nkq;                      \
            }                                                                                  \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                   \
                const double npk = N[P][k], nqk = N[Q][k];                                     \
                N[P][k] = c * npk - s * nqk; N[Q][k] = s * npk + c * nqk;                      \
            }                                                                                  \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                   \
                const double vkp = V[k][P], vkq = V[k][Q];                                     \
                V[k][P] = c * vkp - s * vkq; V[k][Q] = s * vkp + c * vkq;                      \
            }                                                                                  \
        }                                                                                      \
    }

RP_HD void rp_sym4_max_eigvec(double N[4][4], double q[4]) {
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = ((N[0][1] * N[0][1] + N[0][2] * N[0][2]) + (N[0][3] * N[0][3] + N[1][2] * N[1][2])) +
                           (N[1][3] * N[1][3] + N[2][3] * N[2][3]);
        const double diag = (N[0][0] * N[0][0] + N[1][1] * N[1][1]) + (N[2][2] * N[2][2] + N[3][3] * N[3][3]);
        if (off == 0.0 || off <= 1e-34 * diag) break;
        RP_JROT(0, 1) RP_JROT(0, 2) RP_JROT(0, 3) RP_JROT(1, 2) RP_JROT(1, 3) RP_JROT(2, 3)
    }
    double best = N[0][0];
    double v0 = V[0][0], v1 = V[1][0], v2 = V[2][0], v3 = V[3][0];
    if (N[1][1] > best) { best = N[1][1]; v0 = V[0][1]; v1 = V[1][1]; v2 = V[2][1]; v3 = V[3][1]; }
    if (N[2][2] > best) { best = N[2][2]; v0 = V[0][2]; v1 = V[1][2]; v2 = V[2][2]; v3 = V[3][2]; }
    if (N[3][3] > best) { best = N[3][3]; v0 = V[0][3]; v1 = V[1][3]; v2 = V[2][3]; v3 = V[3][3]; }
    const double nrm = sqrt((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3));
    q[0] = v0 / nrm; q[1] = v1 / nrm; q[2] = v2 / nrm; q[3] = v3 / nrm;
}

// The same leading eigenvector WITHOUT the Jacobi sweeps (a serial chain of ~50 rotations with a sqrt and three divisions
// each: ~25 us on one GPU lane, and the IRLS loop calls it 30 times per pair and level): the largest root of the
// characteristic polynomial by Newton's method from an upper bound (monotone: the quartic is convex and increasing to
// the right of its largest root -- Horn's own closed-form route, JOSA A 4(4) 1987 section 4.E), then the eigenvector as
// the best-conditioned column of adj(N - lambda I).  Falls back to the Jacobi solver when the leading eigenvalue is not
// well separated (|p'(lambda)| small: the adjugate then loses digits).  N is preserved.  Returns 1 if the fast path was taken.
RP_HD double rp_det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
RP_HD int rp_sym4_max_eigvec_fast(const double N[4][4], double q[4]) {
    const double n00 = N[0][0], n01 = N[0][1], n02 = N[0][2], n03 = N[0][3], n11 = N[1][1], n12 = N[1][2], n13 = N[1][3],
                 n22 = N[2][2], n23 = N[2][3], n33 = N[3][3];
    const double fro2 = ((n00 * n00 + n11 * n11) + (n22 * n22 + n33 * n33)) + 2.0 * (((n01 * n01 + n02 * n02) + (n03 * n03 + n12 * n12)) + (n13 * n13 + n23 * n23));
    const double scale = sqrt(fro2);
    if (!(scale > 0.0) || !(scale < 1e150)) return 0;
    // characteristic polynomial  p(x) = x^4 + a3 x^3 + a2 x^2 + a1 x + a0  of N / scale
    const double is = 1.0 / scale;
    const double m00 = n00 * is, m01 = n01 * is, m02 = n02 * is, m03 = n03 * is, m11 = n11 * is, m12 = n12 * is, m13 = n13 * is,
                 m22 = n22 * is, m23 = n23 * is, m33 = n33 * is;
    const double tr = (m00 + m11) + (m22 + m33);
    const double a3 = -tr;
    const double a2 = 0.5 * (tr * tr - 1.0);                                   // tr(M^2) = 1 after scaling
    const double p012 = rp_det3(m00, m01, m02, m01, m11, m12, m02, m12, m22), p013 = rp_det3(m00, m01, m03, m01, m11, m13, m03, m13, m33);
    const double p023 = rp_det3(m00, m02, m03, m02, m22, m23, m03, m23, m33), p123 = rp_det3(m11, m12, m13, m12, m22, m23, m13, m23, m33);
    const double a1 = -((p012 + p013) + (p023 + p123));
    const double a0 = m00 * p123 - m01 * rp_det3(m01, m12, m13, m02, m22, m23, m03, m23, m33)
                    + m02 * rp_det3(m01, m11, m13, m02, m12, m23, m03, m13, m33) - m03 * rp_det3(m01, m11, m12, m02, m12, m22, m03, m13, m23);
    double x = 1.0 + 1e-12;                                                     // every eigenvalue of M is <= |M|_F = 1
    double dp = 0.0;
    for (int it = 0; it < 64; ++it) {
        const double px = (((x + a3) * x + a2) * x + a1) * x + a0;
        dp = ((4.0 * x + 3.0 * a3) * x + 2.0 * a2) * x + a1;
        if (!(dp > 0.0)) return 0;
        const double step = px / dp;
        const double xn = x - step;
        if (!(step > 4e-16 * fabs(x))) { x = (step > 0.0) ? xn : x; break; }
        x = xn;
    }
    if (!(dp > 1e-5)) return 0;                                                 // leading eigenvalue not separated: use Jacobi
    // adj(M - x I): all ten distinct cofactors, the column with the largest diagonal cofactor
    const double b00 = m00 - x, b11 = m11 - x, b22 = m22 - x, b33 = m33 - x;
    const double c00 = rp_det3(b11, m12, m13, m12, b22, m23, m13, m23, b33);
    const double c11 = rp_det3(b00, m02, m03, m02, b22, m23, m03, m23, b33);
    const double c22 = rp_det3(b00, m01, m03, m01, b11, m13, m03, m13, b33);
    const double c33 = rp_det3(b00, m01, m02, m01, b11, m12, m02, m12, b22);
    const double c01 = -rp_det3(m01, m12, m13, m02, b22, m23, m03, m23, b33);
    const double c02 = rp_det3(m01, b11, m13, m02, m12, m23, m03, m13, b33);
    const double c03 = -rp_det3(m01, b11, m12, m02, m12, b22, m03, m13, m23);
    const double c12 = -rp_det3(b00, m01, m03, m02, m12, m23, m03, m13, b33);
    const double c13 = rp_det3(b00, m01, m02, m02, m12, b22, m03, m13, m23);
    const double c23 = -rp_det3(b00, m01, m02, m01, b11, m12, m03, m13, m23);
    const double d0 = fabs(c00), d1 = fabs(c11), d2 = fabs(c22), d3 = fabs(c33);
    double v0 = c00, v1 = c01, v2 = c02, v3 = c03, best = d0;
    if (d1 > best) { best = d1; v0 = c01; v1 = c11; v2 = c12; v3 = c13; }
    if (d2 > best) { best = d2; v0 = c02; v1 = c12; v2 = c22; v3 = c23; }
    if (d3 > best) { best = d3; v0 = c03; v1 = c13; v2 = c23; v3 = c33; }
    if (!(best > 1e-7)) return 0;
    // one step of inverse-free refinement: v <- normalised (M - x I + I) v would not help; instead polish with one
    // Rayleigh-quotient correction through the adjugate's own structure: v is already the null vector to O(eps/gap)
    const double nrm = sqrt((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3));
    q[0] = v0 / nrm; q[1] = v1 / nrm; q[2] = v2 / nrm; q[3] = v3 / nrm;
    return 1;
}

RP_HD void rp_quat_to_rot_impl(const double* q, double R[3][3]);
RP_HD void rp_quat_to_rot(const double* q, double R[3][3]) { rp_quat_to_rot_impl(q, R); }

// Horn '87: rotation from the 3x3 weighted covariance M = sum_k w_k s_k t_k^T (rpmodule.py:43-56).
RP_HD void rp_horn_rotation(const double M[3][3], double R[3][3]) {
    double N[4][4] = {
        {M[0][0] + M[1][1] + M[2][2], M[1][2] - M[2][1], M[2][0] - M[0][2], M[0][1] - M[1][0]},
        {M[1][2] - M[2][1], M[0][0] - M[1][1] - M[2][2], M[0][1] + M[1][0], M[0][2] + M[2][0]},
        {M[2][0] - M[0][2], M[0][1] + M[1][0], M[1][1] - M[0][0] - M[2][2], M[1][2] + M[2][1]},
        {M[0][1] - M[1][0], M[2][0] + M[0][2], M[1][2] + M[2][1], M[2][2] - M[0][0] - M[1][1]}};
    double q[4];
    rp_sym4_max_eigvec(N, q);
    rp_quat_to_rot(q, R);
}

// Horn with the Newton / adjugate eigen-solve (Jacobi only as the fallback for a badly separated leading eigenvalue)
RP_HD void rp_horn_rotation_fast(const double M[3][3], double R[3][3]) {
    double N[4][4] = {
        {M[0][0] + M[1][1] + M[2][2], M[1][2] - M[2][1], M[2][0] - M[0][2], M[0][1] - M[1][0]},
        {M[1][2] - M[2][1], M[0][0] - M[1][1] - M[2][2], M[0][1] + M[1][0], M[0][2] + M[2][0]},
        {M[2][0] - M[0][2], M[0][1] + M[1][0], M[1][1] - M[0][0] - M[2][2], M[1][2] + M[2][1]},
        {M[0][1] - M[1][0], M[2][0] + M[0][2], M[1][2] + M[2][1], M[2][2] - M[0][0] - M[1][1]}};
    double q[4];
    if (!rp_sym4_max_eigvec_fast(N, q)) rp_sym4_max_eigvec(N, q);
    rp_quat_to_rot(q, R);
}

RP_HD void rp_quat_to_rot_impl(const double* q, double R[3][3]) {
    double a = q[0], b = q[1], c = q[2], d = q[3];
    R[0][0] = a * a + b * b - c * c - d * d; R[0][1] = 2 * (b * c - a * d); R[0][2] = 2 * (b * d + a * c);
    R[1][0] = 2 * (c * b + a * d); R[1][1] = a * a - b * b + c * c - d * d; R[1][2] = 2 * (c * d - a * b);
    R[2][0] = 2 * (d * b - a * c); R[2][1] = 2 * (d * c + a * b); R[2][2] = a * a - b * b - c * c + d * d;
}

// General 4x4 inverse by Gauss-Jordan with partial pivoting (np.linalg.inv, evaluation.py:235).
RP_HD bool rp_inv4(const double* A, double* out) {
    double m[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { m[i][j] = A[i * 4 + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(m[r][c]) > fabs(m[piv][c])) piv = r;
        if (m[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
        double inv = 1.0 / m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) {
            double f = m[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = m[i][4 + j];
    return true;
}

// Internals shared by the matcher translation units (matcher.hip: pair consistency + fit; affinity.hip: N x N affinity + top-K).
#pragma once
#include "common.h"
#include "rp_math.h"

#define RP_MAXK 8
#define RP_FEAT 32

// rpmodule.py:342-363 / :399-467 constants, derived on the host in double exactly as numpy does
static inline RpPairConsts rp_make_consts(const RelposeParams& p) {
    RpPairConsts k;
    k.dist_thre2 = p.distThre * p.distThre;
    k.sep_thre = 1.5 * (p.distSepThre * p.distSepThre);
    k.angle_thre2 = p.angleThre * p.angleThre;
    k.two_sd2 = 2 * (p.sigmaDist * p.sigmaDist);
    k.two_sa1_2 = 2 * (p.sigmaAngle1 * p.sigmaAngle1);
    k.two_sa2_2 = 2 * (p.sigmaAngle2 * p.sigmaAngle2);
    double s1 = (p.sigmaFeat / 1.2) / 5, s0 = p.sigmaFeat / 5;
    k.den_both = 2 * (s1 * s1);
    k.den_other = 2 * (s0 * s0);
    k.mu = p.mu;
    return k;
}

// Process-wide kernel-selection knobs (relpose_set_tuning, include/relpose.h).  They choose between kernels that produce the
// same results; the parity tests use them to push every variant through the same checks.  Read once per C-ABI call.
extern int32_t g_rp_tune[RELPOSE_TUNE_COUNT];

// affinity.hip: rpmodule.py:342-379 for a batch of pairs (wij may be null: fused variant)
// sel_call: the call's own kernel choice (RelposeMatchArgs::affinity_kernel), 0 = the process-wide test knob / by size
int rp_launch_affinity(const RelposeParams& p, const RelposeKeypoints& kp, float* wij, int32_t* cj, double* cw, int32_t* keff, hipStream_t s, int sel_call);

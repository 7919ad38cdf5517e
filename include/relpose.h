/*
 * relpose.h -- C ABI of librelpose_hip.so, the MI355X (gfx950) implementation of
 * the relative-pose inference hot path of zhenpeiyang/RelativePose.
 *
 * The reference has no FFI: its boundary is a set of Python call sites.  Each
 * entry point below names the reference function it replaces (file:line in the
 * upstream tree).  INTEGRATION.md shows the ctypes stubs a maintainer would add
 * to the reference to route those call sites here.
 *
 * Conventions
 *  - every pointer argument is DEVICE memory unless its name ends in _host;
 *  - the caller owns all buffers; the library never frees caller memory.  Its own
 *    device allocations belong to a RelposeSCNet handle and are freed by
 *    relpose_scnet_destroy: the packed weights (relpose_scnet_finalize) and one
 *    launch-descriptor table per (workspace, n_images) pair, built by the FIRST
 *    relpose_scnet_forward on that workspace (that first call hipMallocs + copies
 *    synchronously; later calls only enqueue).  Scratch comes from caller-provided
 *    workspaces whose size is returned by the *_workspace_bytes functions;
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream); apart from the first-forward plan build and
 *    relpose_scnet_finalize / relpose_scnet_profile no entry point synchronises;
 *  - return value: 0 = enqueued, <0 = invalid argument (RELPOSE_EINVAL) or HIP
 *    error (-(1000+hipError_t)).  Per-pair degenerate inputs are NOT errors: the
 *    reference returns identity for them (rpmodule.py:346-348,377-379,406-408,
 *    440-443,469-472) and so do we, with the reason in status[b];
 *  - panoramas are four-face skyboxes, h rows x 4h columns (the reference
 *    hard-codes h = 160); images are NCHW float32 like the reference tensors.
 */
#ifndef RELPOSE_H
#define RELPOSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELPOSE_EINVAL (-1)
#define RELPOSE_ENOMEM (-2)

/* status[b] of relpose_match_pairs */
enum {
    RELPOSE_OK = 0,
    RELPOSE_FEW_KEYPOINTS = 1, /* <3 keypoints or <3 correspondences      rpmodule.py:346,377 */
    RELPOSE_DIST_FILTER = 2,   /* <3 pairs pass the distance test          rpmodule.py:406 */
    RELPOSE_ANGLE_FILTER = 3,  /* <3 pairs pass the angle test             rpmodule.py:440 */
    RELPOSE_ZERO_WEIGHT = 4,   /* every pair weight is 0                   rpmodule.py:469 */
    RELPOSE_EDGE_OVERFLOW = 5, /* more surviving pairs than the workspace was sized for */
    RELPOSE_NOT_CONVERGED = 6  /* a leading-eigenvector solve (rpmodule.py:273, ARPACK there) did not reach its residual
                                  tolerance within the restart budget; the pose is computed from the last iterate */
};

enum { RELPOSE_SUNCG = 0, RELPOSE_MATTERPORT = 1, RELPOSE_SCANNET = 2 };   /* dataset conventions */
enum { RELPOSE_MASK_SECOND = 0, RELPOSE_MASK_KINECT = 1 };                   /* util.apply_mask methods */
enum { RELPOSE_COMPOSE_EVAL = 0, RELPOSE_COMPOSE_LIB = 1 };                       /* output composition variants */
enum { RELPOSE_FIT_IRLS_SM = 0, RELPOSE_FIT_HORN87 = 1, RELPOSE_FIT_IRLS = 2, RELPOSE_FIT_SPECTRAL = 3 };

/* Hyper-parameters of the pose module: class opts, RPModule/rputil.py:11-22. */
typedef struct RelposeParams {
    double distThre;    /* 0.08 */
    double distSepThre; /* 1.5*0.08 */
    double angleThre;   /* pi/4 */
    double sigmaAngle1, sigmaAngle2, sigmaDist, sigmaFeat;
    double mu;          /* 0.3 */
    int32_t topK;       /* 5 (<= 8) */
    int32_t method;     /* RELPOSE_FIT_* ; 'irls+sm' by default */
} RelposeParams;

void relpose_default_params(RelposeParams* p_host);

/* Process-wide kernel-selection knobs.  Every setting produces the same results (the parity tests force each variant
 * through the same checks); they exist for those tests and for tuning, NOT for the data path: nothing in the product path sets them
 * (round 6: the one per-call choice the serving loop makes, the fit's workgroups per pair, is an argument of relpose_match_pairs_ex).
 * Returns the previous value, RELPOSE_EINVAL for an unknown key.  Not synchronised with calls in flight on other threads.
 *   RELPOSE_TUNE_AFFINITY_KERNEL    0 = by batch size (default), 1 = row kernel (targets in registers), 2 = tile kernel
 *                                   (fp16-MFMA candidates + exact arithmetic on them), 3 = LDS kernel (the nt_max > 512 path),
 *                                   4 = pool variant (round 5: the tile kernel's stages as separate dense launches; auto-selected for the
 *                                   fused form beyond 256 targets at large batches, where it measured faster; it keeps one grow-only scratch
 *                                   block per (device, stream) it was called on for the life of the process: ~0.3 KB per source row)
 *   RELPOSE_TUNE_FIT_MAX_PRODUCTS   0 = default budget (192) of matrix-vector products per eigen-solve; a tiny budget
 *                                   forces RELPOSE_NOT_CONVERGED
 *   RELPOSE_TUNE_FIT_CLUSTER        0 = by problem size (default), n = workgroups per scan pair in the fit (1, 2, 4, 8)
 *   RELPOSE_TUNE_FIT_GLOBAL_VECTORS 0 = by problem size, 1 = keep the fit's per-correspondence vectors in global memory
 *                                   (the > 4500-correspondence layout) whatever the size
 *   RELPOSE_TUNE_FIT_FIXED_CHECKS   0 = the eigen-solve places its convergence tests where the residual estimate is predicted to
 *                                   reach the tolerance (default), 1 = a test every 8 products (the earlier rule; A/B switch).  The
 *                                   one knob whose settings agree to round-off only (both converge to 1e-13; the number of Lanczos
 *                                   steps differs) */
enum { RELPOSE_TUNE_AFFINITY_KERNEL = 0, RELPOSE_TUNE_FIT_MAX_PRODUCTS = 1, RELPOSE_TUNE_FIT_CLUSTER = 2,
       RELPOSE_TUNE_FIT_GLOBAL_VECTORS = 3, RELPOSE_TUNE_FIT_FIXED_CHECKS = 4, RELPOSE_TUNE_COUNT = 8 };
int relpose_set_tuning(int32_t key, int32_t value);
const char* relpose_version(void);
/* ------------------------------------------------------------------ matcher
 * Keypoint sets of B scan pairs, padded to ns_max / nt_max rows.
 * Mirrors the dict the reference helper takes (rpmodule.py:317-326):
 * 'pc'[k,3] f64, 'normal'[k,3] f64, 'feat'[k,32] f32, 'weight'[k] f64. */
typedef struct RelposeKeypoints {
    int32_t B, ns_max, nt_max;
    const int32_t* ns;      /* [B] valid source keypoints per pair */
    const int32_t* nt;      /* [B] */
    const double* pc_s;     /* [B, ns_max, 3] */
    const double* normal_s; /* [B, ns_max, 3] */
    const float* feat_s;    /* [B, ns_max, 32] (unscaled; /100 happens inside, rpmodule.py:342) */
    const double* weight_s; /* [B, ns_max] */
    const double* pc_t;     /* [B, nt_max, 3] */
    const double* normal_t;
    const float* feat_t;
    const double* weight_t;
} RelposeKeypoints;

/* Optional intermediate outputs of the matcher (any pointer may be NULL). */
typedef struct RelposeMatchDebug {
    float* wij;          /* [B, ns_max, nt_max] row-normalised affinity (rpmodule.py:354-363), f32 */
    int32_t* corres_j;   /* [B, ns_max, topK]   target index of each top-K correspondence (:367-374) */
    double* corres_w;    /* [B, ns_max, topK]   wij at those entries, f64 */
    int32_t* counts;     /* [B, 4]  {#pairs passing distance test, #surviving pairs M, #nonzero weights, K_eff} */
    double* trace;       /* [B, 6, 16] pose after the initial IRLS and after each of the 5 spectral rounds */
    int32_t* eig_iters;  /* [B, 5]  power iterations used per spectral round */
} RelposeMatchDebug;

/* max_edges = capacity, per pair, of the symmetric pair-compatibility graph
 * (2 x surviving pairs); 0 = worst case ns_max*topK*(ns_max*topK-1).
 * LIMITS (the reference has none; its largest shipped configuration uses 400 keypoints per view): ns_max * topK <=
 * RELPOSE_MAX_CORRESPONDENCES (the pair-consistency kernels keep a row's correspondence list in LDS) and nt_max <= RELPOSE_MAX_TARGETS
 * (beyond 512 targets the affinity kernel keeps all target descriptors of a pair in LDS).  Outside them
 * relpose_match_workspace_bytes returns 0 and relpose_match_pairs / relpose_affinity_topk RELPOSE_EINVAL. */
#define RELPOSE_MAX_CORRESPONDENCES 8192
#define RELPOSE_MAX_TARGETS 1152
size_t relpose_match_workspace_bytes(int32_t B, int32_t ns_max, int32_t nt_max, int32_t topK, int64_t max_edges);

/* Replaces RelativePoseEstimation_helper (RPModule/rpmodule.py:317-508) for a
 * batch of pairs: affinity + top-K + pairwise consistency + fit.  pose [B,16]
 * row-major 4x4 float64, status [B]. */
int relpose_match_pairs(const RelposeParams* params_host, const RelposeKeypoints* kp_host,
                        void* workspace, size_t workspace_bytes, int64_t max_edges,
                        double* pose, int32_t* status, const RelposeMatchDebug* debug_host, void* stream);

/* relpose_match_pairs with its per-call choices IN the call (round 6; the reference call carries its own `para`, rpmodule.py:317-326): a
 * struct_size-versioned argument block like RelposeForwardArgs -- fields beyond the caller's struct_size take their defaults (0).
 *   params_host .. stream  as relpose_match_pairs
 *   fit_cluster            workgroups per scan pair in the robust fit (rpmodule.py:212-315): 0 = by problem size (small batches of large pairs
 *                          get helper workgroups: a latency tool), 1 = none (what a serving loop with several batches in flight wants: helpers
 *                          take compute units from the other batches' convolutions), 2 / 4 / 8.  Every setting gives bitwise the same poses.
 *                          This is the per-call form of RELPOSE_TUNE_FIT_CLUSTER; a non-zero value here wins over the process-wide test knob,
 *                          so concurrent callers with different choices do not interfere (pipeline.run_pipelined uses it).
 *   affinity_kernel        0 = by batch size, 1..4 as RELPOSE_TUNE_AFFINITY_KERNEL (same results); a non-zero value wins over the test knob. */
typedef struct RelposeMatchArgs {
    uint32_t struct_size;
    int32_t fit_cluster;
    const RelposeParams* params_host;
    const RelposeKeypoints* kp_host;
    void* workspace;
    size_t workspace_bytes;
    int64_t max_edges;
    double* pose;
    int32_t* status;
    const RelposeMatchDebug* debug_host;
    void* stream;
    int32_t affinity_kernel;
    int32_t reserved0;
} RelposeMatchArgs;
int relpose_match_pairs_ex(const RelposeMatchArgs* args);

/* Stage A+B alone (rpmodule.py:342-379): N x N affinity build and row top-K.
 * wij may be NULL (fused variant: the matrix is never materialised). */
int relpose_affinity_topk(const RelposeParams* params_host, const RelposeKeypoints* kp_host,
                          float* wij, int32_t* corres_j, double* corres_w, int32_t* k_eff, void* stream);

/* ----------------------------------------------------------------- geometry */

/* util.apply_mask (util.py:209-232): x [n,c,h,4h] -> x*mask in place, mask [n,1,h,4h] (may be NULL). */
int relpose_apply_mask(float* x, float* mask, int32_t n, int32_t c, int32_t h, int32_t method, void* stream);

/* evaluation.py:217-230: view [n,8,h,4h] = mask*(rgb,norm,depth) ++ (masked depth != 0). */
int relpose_build_view(const float* rgb, const float* norm, const float* depth, float* view,
                       int32_t n, int32_t h, int32_t method, void* stream);

/* util.Pano2PointCloud (util.py:751-811): depth [n,h,4h] f32 -> pc [n,3,4*h*h] f64 in face-major point
 * order; valid [n,4*h*h] u8 marks the points the reference keeps (scannet drops depth==0). */
int relpose_pano2pc(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t h, int32_t dataset, void* stream);

/* util.warping (util.py:94-172) incl. depth2pc (:468-523) and reproj_helper (:537-749):
 * view [n,8,h,4h] f32, pose [n,16] f64 -> out [n,8,h,4h] f32 (the f64 result cast like torch_op.v);
 * identity pose gives zeros.  workspace: relpose_warp_workspace_bytes(n,h). */
size_t relpose_warp_workspace_bytes(int32_t n, int32_t h);
int relpose_warp(const float* view, const double* pose, float* out, void* workspace,
                 int32_t n, int32_t h, int32_t dataset, void* stream);

/* The warp of evaluation.py:221-222,235-236 for a batch of pairs, in place on the network input
 * x [n,16,h,4h] (n even; images 2b, 2b+1 are a pair): channels 0:8 of every image hold its own view
 * (caller-filled, only read here); channels 8:16 of image i receive the view of its partner (image i^1)
 * warped by pose[i] -- i.e. torch.cat((view, warping(other, pose)), 1) without materialising `other`
 * or the concatenation.  Same workspace as relpose_warp. */
int relpose_warp_pairs(float* x, const double* pose, void* workspace, int32_t n, int32_t h, int32_t dataset, void* stream);
/* The same with flags.  RELPOSE_WARP_KEYS_CLEAN: the first 4 n h 4h bytes of the workspace (the per-pixel winner keys of the scatter
 * pass) are all zeros -- true for a workspace that was zero-initialised once and has only been used by relpose_warp / relpose_warp_pairs*
 * since: every call resets the keys it consumed.  The 26 MB memset in front of the scatter pass (64 panoramas) is then skipped. */
enum { RELPOSE_WARP_KEYS_CLEAN = 1 };
int relpose_warp_pairs2(float* x, const double* pose, void* workspace, int32_t n, int32_t h, int32_t dataset, int32_t flags, void* stream);

/* np.linalg.inv of n 4x4 poses (evaluation.py:235). */
int relpose_pose_inverse(const double* pose, double* inv, int32_t n, void* stream);

/* evaluation.py:246-253 + getMatchingPrimitive after keypoint detection (rpmodule.py:526-532):
 * compose completed normal/depth, bilinear-sample them at the keypoints (rputil.getPixel :88-119),
 * unproject, and gather 32-d descriptors (rputil.interpolate :43-58).
 *   f        [n, cf, h, 4h]  network output; normal = ch 3:6, depth = ch 6, feat = ch feat_off:feat_off+32
 *   obs_norm [n,3,h,4h], obs_depth [n,h,4h]  the complete input scan (evaluation.py:248-253)
 *   pts      [n, npts_max, 2] f64 pixel coords (x,y), x<=4h-2, y<=h-2 ; npts [n]
 *   compose  RELPOSE_COMPOSE_EVAL: normal / (|obs normal| + 1e-6)   evaluation.py:250-251
 *            RELPOSE_COMPOSE_LIB : normal / (|normal| + 1e-12)      rpmodule.py:633-634 (RelativePoseEstimationViaCompletion)
 *   outputs  pc [n,npts_max,3] f64, normal [n,npts_max,3] f64, feat [n,npts_max,32] f32 */
int relpose_sample_primitives(const float* f, int32_t cf, int32_t feat_off,
                              const float* obs_norm, const float* obs_depth,
                              const double* pts, const int32_t* npts, int32_t npts_max,
                              double* pc, double* normal, float* feat,
                              int32_t n, int32_t h, int32_t mask_method, int32_t compose, int32_t dataset, void* stream);

/* The same two samplers on caller-composed maps, for the reference-named host shims:
 * rputil.getPixel (rputil.py:88-119, incl. getPixel_helper :61-86): depth [h,4h] f64, normal [h,4h,3] f64 (HWC),
 * pts [k,2] f64 pixel coords -> pc [k,3] f64 (the reference returns the transpose), nn [k,3] f64 (renormalised, not rotated). */
int relpose_get_pixel(const double* depth, const double* normal, const double* pts, int32_t k, int32_t h, int32_t dataset,
                      double* pc, double* nn, void* stream);
/* rputil.interpolate (rputil.py:43-58): feat [c,h,w] f32, pt [k,2] f32 normalised to [0,1] -> out [c,k] f32. */
int relpose_interpolate(const float* feat, const float* pt, float* out, int32_t c, int32_t h, int32_t w, int32_t k, void* stream);

/* ------------------------------------------------- evaluation-side statistics
 * util.depth2pc (util.py:468-523) of the observed block of each panorama (the face / kinect crop that
 * util.parse_data :42-92 feeds it): pc [n, P, 3] f64 in pixel order, valid [n, P] = depth != 0, with
 * P = relpose_observed_points(h, dataset) (25600 or 66*88 at h = 160). */
int32_t relpose_observed_points(int32_t h, int32_t dataset);
int relpose_depth2pc(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t h, int32_t dataset, void* stream);
/* util.depth2pc's full-resolution kinect branch (util.py:497-507, reached from util.parse_data :79-90 for the baseline methods on
 * ScanNet): depth [n, 480, 640] f32 IMAGES (no panorama around them) -> pc [n, 480*640, 3] f64 in pixel order, valid = depth != 0.
 * The reference defines the branch for this one shape; any other returns RELPOSE_EINVAL (round 6). */
int relpose_depth2pc_full(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t hh, int32_t ww, void* stream);

/* Nearest-neighbour distances behind util.point_cloud_overlap (util.py:21-40, sklearn KDTree there):
 * dist[i] = min_j || pose*query_i - ref_j || over valid ref points (pose [12+] row-major 3x4/4x4, or NULL);
 * invalid query points get -1.  query [nq,3], ref [nr,3] f64. */
int relpose_nn_dist(const double* query, const uint8_t* query_valid, int32_t nq, const double* ref, const uint8_t* ref_valid,
                    int32_t nr, const double* pose, double* dist, void* stream);

/* ------------------------------------ feature-guided keypoint augmentation
 * rputil.getKeypoint :182-190: dist [nsel,H,W] f32 = sum_c (query[s][c] - feat[c][y][x])^2
 * (query [nsel,32] = descriptors of the selected keypoints, feat [32,H,W] the other view's feature map). */
int relpose_feature_distance_map(const float* query, const float* feat, float* dist, int32_t nsel, int32_t H, int32_t W, void* stream);
/* rputil.Sampling :355-371 on exp(-dist/2): K x {argmax, suppress a +-window box}; pts [nmaps,K,2] f64 (x,y). */
int relpose_nms_sampling(const float* dist, double* pts, int32_t nmaps, int32_t H, int32_t W, int32_t K, int32_t window, void* stream);

/* The per-level keypoint derivation of the reference behind its SIFT detector, batched over the views of a batch of scan pairs
 * (rputil.getKeypoint :141-237 / getKeypoint_kinect :240-353, called at every recurrent level through getMatchingPrimitive,
 * rpmodule.py:511-533, evaluation.py:278): descriptors at the query points (interpolate :43-58), for every query the `topk`
 * non-maximum-suppressed minima of its squared descriptor distance over the OTHER view's 32-channel feature map (:182-190, :212-214,
 * Sampling :355-371) -- the [n,H,W] distance maps are never materialised --, the validity filter (:192-196), concatenation and the
 * weights 1 / 0.99 (:226-235).  The np.random draws of getKeypoint do not depend on the features: the host pre-draws them in the
 * reference's call order (relativepose_amd.rputil.keypoint_plan) and passes
 *   f [n_views, ., H, W] the network output (features = channels feat_off .. feat_off+32 of every image; image_stride floats per image)
 *   q_src_view [nq] image whose features a query samples, q_pt [nq,2] its normalised (x/W, y/H) float32 point,
 *   q_map_view [nq] the view whose map it searches, queries grouped by that view: q_off [n_views+1]; nq_view_max = max group size
 *   slot_kind [n_views,L]: the keypoint slots of every view in the reference's concatenation order: -1 empty, -2 a host coordinate
 *   (slot_xy [n_views,L,2]: SIFT detections, random points), >= 0 the pick with linear index query * topk + k
 * Outputs: pts [n_views,L,2] f64 pixel coordinates (x,y) compacted in slot order, weight [n_views,L] f64, npts [n_views].
 * flags: RELPOSE_KP_OBSERVED_ONLY = getMatchingPrimitive(..., doCompletion = 0), rpmodule.py:534-537 (the 'ours_nc' method, evaluation.py:74): only
 * the keypoints of weight 1 (inside the observed region) are kept -- the reference filters the sampled primitives by ptsW == 1, which is the same
 * selection in the same order.  nq = 0 is legal (round 6): a batch whose views have no SIFT detections has no queries; a view whose slots are
 * all empty gets npts = 0 and its pair the matcher's "return identity" status, like the reference (rputil.py:156-166, rpmodule.py:522-523,
 * evaluation.py:280-282).  nq_view_max <= RELPOSE_KP_MAX_QUERIES_PER_VIEW (the per-view query descriptors and running bests live in LDS). */
enum { RELPOSE_KP_OBSERVED_ONLY = 1 };
#define RELPOSE_KP_MAX_QUERIES_PER_VIEW 1000
size_t relpose_keypoints_reference_workspace_bytes(int32_t nq, int32_t H, int32_t W, int32_t topk);
int relpose_keypoints_reference(const float* f, int64_t image_stride, int32_t feat_off, int32_t n_views, int32_t H, int32_t W,
                                const int32_t* q_src_view, const float* q_pt, const int32_t* q_map_view, const int32_t* q_off, int32_t nq,
                                int32_t nq_view_max, int32_t topk, int32_t window, const int32_t* slot_kind, const double* slot_xy, int32_t L,
                                int32_t mask_method, int32_t flags, double* pts, double* weight, int32_t* npts, void* workspace, size_t workspace_bytes,
                                void* stream);

/* -------------------------------------------------------------------- SCNet
 * Replaces SCNet (model/mymodel.py:141-380).  relpose_scnet_create builds the configuration evaluation.py runs
 * (skipLayer=1, batchnorm=1, outputType 'rgbdnsf'); relpose_scnet_create_ex (round 6) the other constructor variants. */
typedef struct RelposeSCNet RelposeSCNet;

RelposeSCNet* relpose_scnet_create(int32_t snumclass, int32_t use_tanh);
void relpose_scnet_destroy(RelposeSCNet* net);

/* The reference constructor's switches (model/mymodel.py:145-149, 189-243: args.batchnorm, args.skipLayer, args.outputType).
 *   batchnorm   1: conv -> BatchNorm(batch statistics) -> LeakyReLU (mymodel.py:16-21); 0: conv + bias -> LeakyReLU (:22-25;
 *               state-dict keys "<block>.0.bias" instead of "<block>.1.weight|bias").
 *   skip_layer  1: every decoder block also reads the encoder activation of its resolution (:302-307); 0: the plain chain (:335-340).
 *   output_mask which heads exist, RELPOSE_OUT_* bits in the reference's concatenation order rgb, n, d, s, f (:309-376).  A head that
 *               was not constructed has no parameters and its decoder branch does not run.
 * NULL for what the reference cannot run either: skip_layer = 0 with any of rgb / n / d (their 1x1 output convs always take 64
 * input channels, 32 of them the skip: the reference fails inside torch, mymodel.py:192 vs :347) and the 'k' head (reads an
 * undefined `xsift`, :328).
 * The forward's `out` keeps the layout [n, 7 + snumclass + 32, H, W] for every variant: the channels of a head that does not
 * exist are 0 (the reference concatenates only the existing heads; relativepose_amd.model.SCNet gathers them).  Variants run the
 * same kernels under the plain launch plan on ONE stream: RELPOSE_FWD_ZERO_WARP / _POSE_OUTPUTS, `self_tag` and `tail_stream`
 * are accepted and ignored (each is "bitwise the plain forward" by contract). */
enum { RELPOSE_OUT_RGB = 1, RELPOSE_OUT_N = 2, RELPOSE_OUT_D = 4, RELPOSE_OUT_S = 8, RELPOSE_OUT_F = 16, RELPOSE_OUT_ALL = 31 };
typedef struct RelposeSCNetConfig {
    uint32_t struct_size;       /* sizeof(RelposeSCNetConfig) as the caller compiled it */
    int32_t snumclass, use_tanh;
    int32_t batchnorm, skip_layer, output_mask;
} RelposeSCNetConfig;
RelposeSCNet* relpose_scnet_create_ex(const RelposeSCNetConfig* cfg);

/* One state_dict entry (key names = the reference module tree, e.g. "conv4.0.weight",
 * "deconv3rgb.1.bias", "deconv1f.weight"); data_host float32 in torch layout. */
int relpose_scnet_set_param(RelposeSCNet* net, const char* key, const float* data_host, size_t numel);
/* Pack + upload once all keys are set; returns <0 and lists nothing if a key is missing. */
int relpose_scnet_finalize(RelposeSCNet* net);
int64_t relpose_scnet_num_params(const RelposeSCNet* net);

/* Arithmetic of the implicit-GEMM convolutions.  RELPOSE_PREC_F32 (default): exact fp32 products on
 * v_mfma_f32_32x32x2_f32 -- the parity configuration.  RELPOSE_PREC_BF16X3 (opt-in): both operands are split
 * into bfloat16 hi + lo and a*b ~= hi*hi + hi*lo + lo*hi runs on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (products exact to ~2^-16 instead of 2^-24; activations, BatchNorm statistics, conv1 and the heads
 * stay fp32).  RELPOSE_PREC_F16X3 (opt-in): the same with float16 halves (11 + 11 mantissa bits: products exact to
 * ~2^-21 for O(1) operands; float16 range, so inputs must stay below 65504).  RELPOSE_PREC_F16 (opt-in): plain float16 products
 * (the hi halves only, one v_mfma_f32_32x32x16_f16 per product, fp32 accumulation and fp32 BatchNorm statistics: SURVEY 8d
 * config 5's "fp16 MFMA convs"; products exact to 2^-11).
 * RELPOSE_PREC_BF16X9 (round 6): EXACT fp32 products on the bf16 matrix pipe -- every fp32 operand is cut into three bfloat16
 * pieces (8 + 8 + 8 = 24 significand bits: a = a1 + a2 + a3 exactly, fp32 exponent range), all nine partial products
 * ai * bj (each exact in the fp32 accumulator) are issued as v_mfma_f32_32x32x16_bf16, smallest first, fp32 accumulation:
 * the arithmetic of the fp32 MFMA path (exact products, fp32 sums in another order) at 9/16 of its matrix-pipe cycles.
 * RELPOSE_PREC_BF16X6: the same without the three smallest partial products a2 b3, a3 b2, a3 b3: what is dropped is at most 2^-23 |a b| (two terms of
 * <= 2^-24 |a b|), observed maximum 2^-24.3 and rms 2^-27.4 over 2e6 random float32 pairs -- a correctly rounded fp32 multiply errs by up to 2^-24 |a b| with an rms
 * of 2^-25.2 (tests/test_split_arithmetic_cpu.py): the size of one fp32 rounding per product, where the fp32 accumulation that follows rounds once per product anyway.
 * May be switched at any time after finalize. */
enum { RELPOSE_PREC_F32 = 0, RELPOSE_PREC_BF16X3 = 1, RELPOSE_PREC_F16X3 = 2, RELPOSE_PREC_F16 = 3, RELPOSE_PREC_BF16X9 = 4, RELPOSE_PREC_BF16X6 = 5 };
int relpose_scnet_set_precision(RelposeSCNet* net, int32_t mode);

size_t relpose_scnet_workspace_bytes(const RelposeSCNet* net, int32_t n_images, int32_t H, int32_t W);

/* forward: x [n,16,H,W] -> out [n,7+S+32,H,W]; n even, BatchNorm statistics over each
 * consecutive group of 2 images (the reference always feeds batch 2, evaluation.py:242).
 * Replaces the reference call `f = net(x)` (evaluation.py:242, rpmodule.py:623; SCNet.forward, model/mymodel.py:259-380). */
int relpose_scnet_forward(RelposeSCNet* net, const float* x, float* out, int32_t n_images, int32_t H, int32_t W,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same forward with every option of the serving loop, in ONE struct_size-versioned argument block (round 5; round 6 removed the
 * accreted relpose_scnet_forward2 / 3 / 4 exports, which were thin wrappers filling this block).
 *   struct_size           sizeof(RelposeForwardArgs) as the CALLER compiled it: fields beyond it take their defaults (0), so the block can grow
 *   x, out, n_images, H, W, workspace, workspace_bytes, stream   as relpose_scnet_forward
 *   tail_stream           NULL = `stream`.  Otherwise the HBM-bound ends of the forward -- the head (input resize mymodel.py:261 + conv1*
 *                         :266-286) and the tail (the five 1x1 heads deconv1* :312-376 + the final resize :379) -- are enqueued on `tail_stream` and
 *                         the MFMA-bound middle on `stream`, ordered by events: `stream` is busy with this forward only between conv2 and deconv2,
 *                         so the next forward -- of ANOTHER workspace -- overlaps its convolutions with this head / tail
 *                         (pipeline.run_pipelined).  x is read and `out` written on tail_stream only.
 *   flags                 RELPOSE_FWD_ZERO_WARP: the caller guarantees that channels 8:16 of EVERY image are zero -- level 0 of the recurrence,
 *                         where the pose estimate is the identity and util.warping returns zeros (util.py:95-96, evaluation.py:232-236).  The
 *                         three warped-view encoder streams (conv1*..conv3* on rgb_t2s / norm_t2s / depth_t2s, mymodel.py:278-288) then produce
 *                         the same activations for every image, so they run for the first BatchNorm group only: results are bitwise those of
 *                         the flag-less forward.  With the flag set and a non-zero warped view the output is undefined.
 *                         RELPOSE_FWD_POSE_OUTPUTS: compute only the outputs the pose path consumes -- normal (channels 3:6), depth (6) and the
 *                         32 feature channels (7+S:), evaluation.py:246-253 / rpmodule.py:629-636 -- and skip the decoder branches that feed
 *                         nothing else (deconv3/2/1 of the rgb and semantic heads, mymodel.py:312-316,364-368); channels 0:3 and 7:7+S of `out`
 *                         are written as zeros, the others are bitwise those of the full forward.  Opt-in; never the default.
 *                         RELPOSE_FWD_NEW_WORKSPACE: the caller (re)allocated the workspace since its last forward -- possibly at the same
 *                         address --: whatever self-stream cache the library associates with the pointer is dropped before this forward.
 *   self_tag              the self-stream cache.  Inside one scan pair's recurrence (evaluation.py:217-242) the masked own views -- channels 0:8
 *                         of every image -- never change between the levels, only the warped partner view (channels 8:16) does, and the
 *                         reference runs the self-view encoder streams as module calls of their own with their own batch statistics
 *                         (conv1/2/3{rgb,n,d} on x[:,0:8], mymodel.py:266-276; the warped-view calls are :278-288).  `self_tag` names the
 *                         content of channels 0:8: when it is non-zero and equal to the tag (and n, H, W) of the PREVIOUS forward on this
 *                         workspace, the self-view blocks of conv1 / conv2 / conv3, their BatchNorm scale / shift, conv4's three self K slices
 *                         and the skip-connection halves of the decoder are taken from the workspace instead of being recomputed -- bitwise the
 *                         values the full forward would produce.  Any other tag (or 0) runs the full forward and leaves the cache filled.
 *                         Contract: two forwards on one workspace that carry the same non-zero tag have identical channels 0:8 (the caller
 *                         draws a fresh tag whenever it rewrites them); set_param / finalize / set_precision drop the cache.
 *                         RELPOSE_FWD_ZERO_WARP forwards always compute (and cache) the self streams.
 *   workspace_generation  the caller's name for THIS ALLOCATION of `workspace` (e.g. a counter bumped whenever the buffer is re-allocated),
 *                         sent with every call: the self-stream cache is used only when the previous forward on the pointer carried the
 *                         same generation -- a workspace re-created at a recycled address is never mistaken for the old one, whichever call
 *                         touches it first (RELPOSE_FWD_NEW_WORKSPACE needs the first call to carry the flag).  0 = not tracked.
 *   reserved1             must be NULL (the experiments build -- RP_EXPERIMENTS -- reads a third stream here, see below).
 * Error behaviour: a call that returns != 0 leaves no self-stream record on the workspace (the next forward recomputes everything); a
 * self-cached plan that cannot be built falls back to the full forward (same output) instead of failing. */
enum { RELPOSE_FWD_ZERO_WARP = 1, RELPOSE_FWD_POSE_OUTPUTS = 2, RELPOSE_FWD_NEW_WORKSPACE = 4 };
typedef struct RelposeForwardArgs {
    uint32_t struct_size;
    int32_t flags;
    const float* x;
    float* out;
    int32_t n_images, H, W, reserved0;
    void* workspace;
    size_t workspace_bytes;
    void* stream;
    void* tail_stream;
    uint64_t self_tag;
    uint64_t workspace_generation;
    void* reserved1;
} RelposeForwardArgs;
int relpose_scnet_forward_ex(RelposeSCNet* net, const RelposeForwardArgs* args);

#ifdef RP_EXPERIMENTS
/* Experiments build only (tools/build_variant.py xp -DRP_EXPERIMENTS): scheduling variants of the serving loop that were measured in round 5
 * and LOST (profiles/r05_loop_experiments.txt); they are not part of the product ABI.
 * RELPOSE_FWD_PART_FRONT / _BACK: one forward enqueued by two calls cut behind the bottleneck chain (conv4's split-K reduction .. deconv6),
 * the chain on the stream in RelposeForwardArgs::reserved1.  relpose_stream_create_cu_limited: a HIP stream confined to the first n_cus compute units. */
enum { RELPOSE_FWD_PART_FRONT = 8, RELPOSE_FWD_PART_BACK = 16 };
int relpose_stream_create_cu_limited(void** stream_out, int32_t n_cus);
int relpose_stream_destroy(void* stream);
#endif

/* Multiply-accumulates one forward of a plan family executes (host-only, no device access): flags as above, self_cached != 0 = the plan
 * a forward takes when it finds its self_tag on the workspace.  The full forward (flags 0, self_cached 0) counts every member of
 * model/mymodel.py:259-380; the level-0 plan and the self-stream cache count what they really launch.  bench.py divides the two for
 * its plan-aware in-loop roofline ("roofline.in_loop"). */
int relpose_scnet_plan_macs(RelposeSCNet* net, int32_t n_images, int32_t flags, int32_t self_cached, double* macs_host);

/* Debug: copy a raw (pre-BatchNorm) layer output of the last forward, NHWC float32, to out (device).
 * Returns the number of floats written (or needed if out is NULL), <0 if unknown. */
int64_t relpose_scnet_read_tap(RelposeSCNet* net, const char* layer, float* out, void* workspace, void* stream);

/* Per-kernel timing of the conv stack for the bench roofline (HIP events on `stream`):
 * runs `iters` forwards and returns mean ms of the implicit-GEMM kernels and of everything else. */
int relpose_scnet_profile(RelposeSCNet* net, const float* x, float* out, int32_t n_images, int32_t H, int32_t W,
                          void* workspace, size_t workspace_bytes, int32_t iters,
                          double* ms_gemm_host, double* ms_other_host, int64_t* n_gemm_launches_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELPOSE_H */

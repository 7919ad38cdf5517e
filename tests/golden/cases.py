"""Case tables shared by make_golden.py (generation, needs the reference) and
the parity tests (which only read the .npz fixtures)."""

MATCH_CASES = [  # (N, Nt, seed, dataset, param row, inlier fraction)
    (50, 50, 11, "suncg", 0, 0.6), (50, 37, 12, "matterport", 1, 0.6), (120, 120, 13, "scannet", 2, 0.6),
    (200, 200, 14, "suncg", 0, 0.6), (200, 200, 15, "suncg", 2, 0.6), (200, 180, 16, "matterport", 0, 0.3),
    (400, 400, 17, "matterport", 1, 0.6), (400, 400, 18, "matterport", 2, 0.1),
    (2, 2, 19, "suncg", 0, 0.6), (6, 4, 20, "suncg", 0, 0.6), (30, 30, 21, "suncg", 0, 0.0),
]
MATCH_METHODS = ("irls+sm", "horn87", "irls", "spectral")
MATCH_BIG512 = (1000, 512, 78, "suncg", 0, 0.8, 0.002)   # 1000 source x 512 target keypoints: the largest shape the affinity tile / pool kernels take (matcher_stages_big512.npz)
MATCH_BIG = (1000, 1000, 77, "suncg", 0, 0.8, 0.002)      # (N, Nt, seed, dataset, param row, inlier fraction, noise): matcher_big.npz

GEOM_CASES = (("suncg", "second", 100), ("matterport", "second", 200), ("scannet", "kinect", 300))
WARP_ANGLES = (0.3, 1.5, 3.141592653589793)

SCNET_CASES = (("a", 15, 1, 3, "suncg", "second"), ("b", 21, 0, 4, "scannet", "kinect"))

# constructor variants of the reference SCNet (mymodel.py:145-149, 189-243): (tag, snumclass, useTanh, weight seed, dataset, mask,
# batchnorm, skipLayer, outputType) -- every combination the reference itself can run is represented: no BatchNorm, no skip connections
# (s / f heads only: the rgb / n / d heads fail inside torch without them), head subsets with and without the skip heads
SCNET_VARIANT_CASES = (("bn0", 15, 1, 31, "suncg", "second", 0, 1, "rgbdnsf"),
                       ("noskip_sf", 15, 1, 32, "suncg", "second", 1, 0, "sf"),
                       ("dnf", 15, 0, 33, "matterport", "second", 1, 1, "dnf"),
                       ("bn0_noskip_f", 15, 1, 34, "suncg", "second", 0, 0, "f"),
                       ("rgbdnf", 21, 1, 35, "scannet", "kinect", 1, 1, "rgbdnf"))

E2E_CASES = [("suncg", "second", 15, 1, 1000 + i) for i in range(4)] + \
    [("matterport", "second", 21, 1, 3000), ("scannet", "kinect", 21, 0, 4000)]
E2E_N = 80
E2E_WEIGHT_SEED = 7

# perturbation envelope of the reference loop (make_golden.gen_e2e_env / gen_e2e_wc): uniform noise of this amplitude on
# every network output (= the measured float32 kernel-vs-reference output difference), this many noise seeds
ENV_AMP = 3e-5
ENV_SEEDS = 8
TF_CASES = (0, 4, 5)           # E2E_CASES indices of the teacher-forced pipeline test (tests/test_gpu_pipeline.py) -> e2e_env_tf.npz

# well-conditioned end-to-end fixtures: synth.make_wc_pair(seed, **WC_KW) + weights.make_descriptor_state_dict
WC_CASES = (9000, 9002, 9004)
WC_KW = dict(n_match=170, n_free=30, angle=0.15, shift=0.2)
WC_S = 15
WC_WEIGHT_SEED = 21
WC_SIGMAS = (0.26, 0.26, 0.04, 0.1)      # sigmaAngle1, sigmaAngle2, sigmaDist, sigmaFeat for all three steps

# the same kind of fixture under the other two dataset conventions (e2e_wc2.npz): Matterport = 'second' mask, S=21, face rotations
# Rs[(i-1)%4]; ScanNet = 'kinect' mask (66x88 observed crop: smaller margin, smaller motion so that the matches stay inside), S=21, no tanh
WC2_CASES = (
    ("matterport", "second", 21, 1, 9106, dict(n_match=170, n_free=30, angle=0.15, shift=0.2, margin=14.0)),
    ("matterport", "second", 21, 1, 9108, dict(n_match=170, n_free=30, angle=0.15, shift=0.2, margin=14.0)),
    ("scannet", "kinect", 21, 0, 9200, dict(n_match=170, n_free=30, angle=0.08, shift=0.1, margin=6.0)),
    ("scannet", "kinect", 21, 0, 9202, dict(n_match=170, n_free=30, angle=0.08, shift=0.1, margin=6.0)),
)

# rputil.getKeypoint / getKeypoint_kinect minus the SIFT detector (getkeypoint.npz): (kind, seed) for synth.make_keypoint_case
GK_CASES = (("second", 31), ("second", 32), ("kinect", 33), ("kinect", 34))

# sigma tuning (tune.npz, make_golden.gen_tune): synth.make_tune_primitives(n_prims, N, seed0), np.random.seed(np_seed), outer iterations
TUNE_CASE = dict(n_prims=3, N=40, seed0=300, np_seed=12345, iters=2)

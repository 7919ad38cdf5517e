"""Tuned per-step sigmas of the pose module: the contents of the reference's
``data/relativePoseModule/final_param_{suncg,matterport,scannet}_rlevel_3.txt`` (12 floats per dataset = configuration
data, read by evaluation.py:156-166).  Rows = recurrent step 0..2, columns = sigmaAngle1, sigmaAngle2, sigmaDist,
sigmaFeat."""

FINAL_PARAMS = {
    "suncg": [[1.2974606092423399, 0.3175894350240194, 0.03550027008268734, 0.008724826760113507],
              [0.2661679311712867, 0.27346134356554863, 0.0401644219835845, 0.009082928258883453],
              [0.2409777299425633, 0.2194442859809447, 0.0415661282255053, 0.009570152123967265]],
    "matterport": [[0.28884460993320005, 0.3723397110060548, 0.04471146704846696, 0.008681938149233242],
                   [0.3301724627277194, 0.22653872741771977, 0.03371542612584658, 0.009278392068704865],
                   [0.44732243168057817, 0.3039564896467746, 0.029312830444192497, 0.011085327519146518]],
    "scannet": [[0.2854414393717704, 0.30015281360048773, 0.042452156783564204, 0.011473403141306663],
                [0.2660066454205768, 0.2745321183723086, 0.03985717942432734, 0.0098980301959599],
                [0.2598844973027187, 0.26241581965965866, 0.04568229318416542, 0.010173330555772159]],
}


def final_params(dataset):
    for k, v in FINAL_PARAMS.items():
        if k in dataset:
            return [list(r) for r in v]
    raise ValueError(f"unknown dataset {dataset}")


def load_param_file(path):
    """evaluation.py:156-166: a text file of 3 x 4 floats -> [[sigmaAngle1, sigmaAngle2, sigmaDist, sigmaFeat]] * 3."""
    import numpy as np
    return np.loadtxt(path).reshape(-1, 4).tolist()

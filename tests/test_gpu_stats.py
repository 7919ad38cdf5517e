"""GPU: depth2pc / point_cloud_overlap vs the oracle (sklearn KDTree like the reference)
and the reference golden values."""
import os

import numpy as np
import pytest

from cases import GEOM_CASES
from gpu_util import log
from oracle import stats_oracle as S
from relativepose_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ds,mm,seed", GEOM_CASES)
def test_depth2pc_and_overlap(golden_dir, ds, mm, seed):
    import torch
    from relativepose_amd import util
    gst = np.load(os.path.join(golden_dir, "stats.npz"))
    d = synth.make_pairs(1, seed + 40, ds)
    dev = torch.device("cuda:0")
    pc, valid = util.depth2pc_dev(torch.from_numpy(d["depth"][0]).to(dev), ds)
    clouds = [pc[v].cpu().numpy()[valid[v].cpu().numpy().astype(bool)] for v in range(2)]
    o_src, o_tgt = S.observed_clouds(d["depth"][0], ds)
    assert np.array_equal(clouds[0], o_src) and np.array_equal(clouds[1], o_tgt)
    # numpy-signature mirror of util.depth2pc on the face / crop
    crop = d["depth"][0, 0][47:113, 196:284] if ds == "scannet" else d["depth"][0, 0][:, 160:320]
    pcm, mask = util.depth2pc(crop, ds)
    assert np.array_equal(pcm, o_src)
    Rgt = gst[f"stats_{ds}_Rgt"]
    got = util.point_cloud_overlap(clouds[0], clouds[1], Rgt)
    want = S.point_cloud_overlap(o_src, o_tgt, Rgt)
    log("overlap", ds=ds, got=[float(x) for x in got], want=[float(x) for x in want], reference=gst[f"stats_{ds}_overlap"].tolist())
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)
    assert np.allclose(got, gst[f"stats_{ds}_overlap"], rtol=1e-9, atol=1e-12)

"""GPU parity: HIP SCNet (through the C ABI) vs the torch-fp32 oracle (which is
pinned bit-for-bit to the reference module by tests/golden/scnet.npz).

Tolerance: float32 kernel, so parity is within float32 round-off of a different
summation order.  Raw conv outputs are compared layer by layer relative to the
layer's scale; raw layers within 5e-4 of the layer scale, final output within 5e-4 absolute (outputs are O(1-10); measured 4e-5..1.1e-4).
The BatchNorm-at-the-bottleneck sensitivity is why the bound is
not tighter: conv9's statistics are over 2 values per channel."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from cases import SCNET_CASES, SCNET_VARIANT_CASES
from gpu_util import log
from oracle.scnet_oracle import SCNetOracle
from relativepose_amd import weights
from test_oracle_golden import oracle_scnet_input

pytestmark = pytest.mark.gpu

# buffer -> list of (oracle tap name, call index, channel offset, channels)
TAP_MAP = {
    "A1": [("conv1rgb", 0, 0, 32), ("conv1rgb", 1, 32, 32), ("conv1n", 0, 64, 32), ("conv1n", 1, 96, 32), ("conv1d", 0, 128, 32), ("conv1d", 1, 160, 32)],
    "A2": [("conv2rgb", 0, 0, 64), ("conv2rgb", 1, 64, 64), ("conv2n", 0, 128, 64), ("conv2n", 1, 192, 64), ("conv2d", 0, 256, 64), ("conv2d", 1, 320, 64)],
    "A3": [("conv3rgb", 0, 0, 128), ("conv3rgb", 1, 128, 128), ("conv3n", 0, 256, 128), ("conv3n", 1, 384, 128), ("conv3d", 0, 512, 128), ("conv3d", 1, 640, 128)],
    "A4": [("conv4", 0, 0, 256)], "A5": [("conv5", 0, 0, 512)], "A6": [("conv6", 0, 0, 512)], "A7": [("conv7", 0, 0, 512)],
    "A8": [("conv8", 0, 0, 512)], "A9": [("conv9", 0, 0, 1024)], "D9": [("deconv9", 0, 0, 512)], "D8": [("deconv8", 0, 0, 512)],
    "D7": [("deconv7", 0, 0, 512)], "D6": [("deconv6", 0, 0, 512)], "D5": [("deconv5", 0, 0, 256)], "D4": [("deconv4", 0, 0, 128)],
    "D3": [("deconv3rgb", 0, 0, 64), ("deconv3n", 0, 64, 64), ("deconv3d", 0, 128, 64), ("deconv3s", 0, 192, 64), ("deconv3f", 0, 256, 64)],
    "D2": [("deconv2rgb", 0, 0, 32), ("deconv2n", 0, 32, 32), ("deconv2d", 0, 64, 32), ("deconv2s", 0, 96, 64), ("deconv2f", 0, 160, 64)],
}


class TapOracle(SCNetOracle):
    """Records every call of a (shared-weight) block, not only the last."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.calls = {}

    def conv(self, x, name, stride, pad):
        y = super().conv(x, name, stride, pad)
        self.calls.setdefault(name, []).append(self.taps[name])
        return y

    def deconv(self, x, name, stride, pad):
        y = super().deconv(x, name, stride, pad)
        self.calls.setdefault(name, []).append(self.taps[name])
        return y


def make_net(S, tanh, seed):
    from relativepose_amd.model import SCNet
    sd = weights.make_state_dict(seed, S)
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(sd)
    assert net.num_params() == weights.num_params(S)
    return net, sd


@pytest.mark.parametrize("prec", ["f32", "bf16x9", "bf16x6", "f16x3", "bf16x3"])
@pytest.mark.parametrize("case", SCNET_CASES)
def test_scnet_layers_and_output_vs_oracle(case, prec, golden_dir):
    """prec = f32 is the parity configuration.  The opt-in split-16-bit modes run through the SAME layer-by-layer check
    against the fp32 oracle with the SAME bounds (measured worst layer error: f32 3e-5, f16x3 3e-5, bf16x3 2e-4)."""
    import torch
    tag, S, tanh, seed, ds, mm = case
    net, sd = make_net(S, tanh, seed)
    net.set_precision(prec)
    tag_log = tag if prec == "f32" else f"{tag}/{prec}"
    x = oracle_scnet_input(500 + seed, ds, mm)
    xd = torch.from_numpy(x).cuda()
    y = net(xd)
    torch.cuda.synchronize()
    orc = TapOracle(sd, S, tanh)
    with torch.no_grad():
        yo = orc.forward(torch.from_numpy(x)).numpy()
    worst = 0.0
    for bname, blocks in TAP_MAP.items():
        t = net.read_tap(bname).cpu().numpy()            # [n,H,H,C]
        for (oname, ci, off, ch) in blocks:
            o = orc.calls[oname][ci].numpy().transpose(0, 2, 3, 1)
            g = t[..., off:off + ch]
            scale = np.abs(o).max() + 1e-30
            err = np.abs(g - o).max() / scale
            log("scnet_layer", case=tag_log, buffer=bname, layer=oname, call=ci, rel_err=err, scale=scale)
            worst = max(worst, err)
            assert err < 5e-4, (bname, oname, ci, err)
    o224 = orc.taps["out224"].numpy().transpose(0, 2, 3, 1)
    g224 = net.read_tap("OUT").cpu().numpy()
    e224 = np.abs(g224 - o224).max()
    yg = y.cpu().numpy()
    eout = np.abs(yg - yo).max()
    gs = np.load(os.path.join(golden_dir, "scnet.npz"))
    eref = np.abs(yg.reshape(-1)[gs[f"{tag}_out_idx"]] - gs[f"{tag}_out_val"]).max()
    log("scnet_output", case=tag_log, worst_layer_rel_err=worst, out224_abs_err=e224, out_abs_err=eout, out_abs_err_vs_reference=eref,
        out_absmax=float(np.abs(yo).max()))
    assert eout < 5e-4 and eref < 5e-4


def make_variant_net(S, tanh, seed, bn, skip, otype, prec="f32"):
    from relativepose_amd.model import SCNet
    sd = weights.make_state_dict(seed, S, bn, skip, otype)
    net = SCNet(SimpleNamespace(batchnorm=bn, useTanh=tanh, skipLayer=skip, outputType=otype, snumclass=S))
    net.load_state_dict(sd)
    net.set_precision(prec)
    return net, sd


@pytest.mark.parametrize("prec", ["f32", "bf16x6"])
@pytest.mark.parametrize("case", SCNET_VARIANT_CASES, ids=[c[0] for c in SCNET_VARIANT_CASES])
def test_scnet_constructor_variants_vs_oracle_and_reference(case, prec, golden_dir):
    """The reference constructor's other switches (mymodel.py:145-149, 189-243; relpose_scnet_create_ex): batchnorm=0 (conv bias through the
    loader's {scale, shift} table), skipLayer=0 (single-source decoder), head subsets.  Output against the fp32 oracle of the same variant and
    against samples of the REFERENCE module's output (tests/golden/scnet_variants.npz), relative to the output's scale; the pre-activation
    bottleneck and trunk layers against the oracle's like the default configuration."""
    import torch
    tag, S, tanh, seed, ds, mm, bn, skip, otype = case
    net, sd = make_variant_net(S, tanh, seed, bn, skip, otype, prec)
    x = oracle_scnet_input(500 + seed, ds, mm)
    y = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    orc = TapOracle(sd, S, tanh, bn, skip, otype)
    with torch.no_grad():
        yo = orc.forward(torch.from_numpy(x)).numpy()
    yg = y.cpu().numpy()
    assert yg.shape == yo.shape
    worst = 0.0
    for bname, blocks in TAP_MAP.items():
        t = net.read_tap(bname).cpu().numpy()
        for (oname, ci, off, ch) in blocks:
            if oname not in orc.calls:
                assert not np.any(t[..., off:off + ch]), (bname, oname)      # a head that does not exist: its block is zero, not garbage
                continue
            o = orc.calls[oname][ci].numpy().transpose(0, 2, 3, 1)
            # (batchnorm=0: the reference's conv output includes the bias; the library adds it in the consumer's loader)
            if not bn:
                o = o - sd[f"{oname}.0.bias"][None, None, None, :]
            scale = np.abs(o).max() + 1e-30
            err = np.abs(t[..., off:off + ch] - o).max() / scale
            log("scnet_variant_layer", case=f"{tag}/{prec}", buffer=bname, layer=oname, call=ci, rel_err=err, scale=scale)
            worst = max(worst, err)
            assert err < 5e-4, (bname, oname, ci, err)
    gv = np.load(os.path.join(golden_dir, "scnet_variants.npz"))
    ref = gv[f"{tag}_out_val"]
    scale = max(float(np.abs(ref).max()), 1e-30)
    eout = np.abs(yg - yo).max() / scale
    eref = np.abs(yg.reshape(-1)[gv[f"{tag}_out_idx"]] - ref).max() / scale
    log("scnet_variant_output", case=f"{tag}/{prec}", worst_layer_rel_err=worst, out_rel_err=eout, out_rel_err_vs_reference=eref, out_absmax=scale)
    assert eout < 5e-4 and eref < 5e-4
    # the flags of the evaluation.py plans are accepted and ignored by a variant: bitwise the plain forward
    y2 = net(torch.from_numpy(x).cuda(), zero_warp=True, outputs="pose", self_tag=77)
    y3 = net(torch.from_numpy(x).cuda(), self_tag=77)
    assert torch.equal(y, y2) and torch.equal(y, y3)
    # batches: BatchNorm groups of 2 images, as in the default configuration
    xb = np.concatenate((x, x[::-1].copy(), x))
    yb = net(torch.from_numpy(xb).cuda()).cpu().numpy()
    assert np.abs(yb[:2] - yg).max() <= 1e-5 * scale and np.abs(yb[4:] - yg).max() <= 1e-5 * scale


def test_scnet_variants_the_reference_cannot_run_are_refused():
    from relativepose_amd import _lib
    from relativepose_amd.model import SCNet
    import ctypes as C
    with pytest.raises(ValueError):
        SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=0, outputType="rgbdnsf", snumclass=15))
    with pytest.raises(ValueError):
        SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnksf", snumclass=15))
    cfg = _lib.SCNetConfig()
    cfg.struct_size = C.sizeof(_lib.SCNetConfig)
    cfg.snumclass, cfg.use_tanh, cfg.batchnorm, cfg.skip_layer, cfg.output_mask = 15, 1, 1, 0, 31
    assert not _lib.lib().relpose_scnet_create_ex(C.byref(cfg))
    cfg.output_mask = 0
    assert not _lib.lib().relpose_scnet_create_ex(C.byref(cfg))
    cfg.output_mask, cfg.struct_size = 24, 8
    assert not _lib.lib().relpose_scnet_create_ex(C.byref(cfg))
    # a variant net refuses a state dict of another variant (missing keys), loudly
    net = SCNet(SimpleNamespace(batchnorm=0, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=15))
    with pytest.raises(RuntimeError):
        net.load_state_dict(weights.make_state_dict(1, 15))


F16_LAYER_BOUND = 2e-2        # plain fp16 products: per-layer max error relative to the layer's scale (measured worst: see the log)
F16_OUT_BOUND = 1e-1          # final output, absolute (outputs are O(1-10))


@pytest.mark.parametrize("case", SCNET_CASES)
def test_scnet_plain_f16_layers_and_output_vs_oracle(case):
    """RELPOSE_PREC_F16 -- SURVEY 8(d) config 5's literal "fp16 MFMA convs" (one v_mfma_f32_32x32x16_f16 per product, operands
    rounded to fp16 when a tile is staged, fp32 accumulation, fp32 BatchNorm statistics) -- layer by layer against the fp32
    ORACLE like the parity configuration, with its own (measured, logged) bounds: an accuracy trade the caller opts into."""
    import torch
    tag, S, tanh, seed, ds, mm = case
    net, sd = make_net(S, tanh, seed)
    net.set_precision("f16")
    x = oracle_scnet_input(500 + seed, ds, mm)
    y = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    orc = TapOracle(sd, S, tanh)
    with torch.no_grad():
        yo = orc.forward(torch.from_numpy(x)).numpy()
    worst, worst_at, mean_rel = 0.0, None, []
    for bname, blocks in TAP_MAP.items():
        t = net.read_tap(bname).cpu().numpy()
        for (oname, ci, off, ch) in blocks:
            o = orc.calls[oname][ci].numpy().transpose(0, 2, 3, 1)
            g = t[..., off:off + ch]
            scale = np.abs(o).max() + 1e-30
            err = float(np.abs(g - o).max() / scale)
            mean_rel.append(float(np.abs(g - o).mean() / scale))
            log("scnet_layer", case=f"{tag}/f16", buffer=bname, layer=oname, call=ci, rel_err=err, scale=float(scale))
            if err > worst:
                worst, worst_at = err, oname
    yg = y.cpu().numpy()
    eout, emean = float(np.abs(yg - yo).max()), float(np.abs(yg - yo).mean())
    log("scnet_output", case=f"{tag}/f16", worst_layer_rel_err=worst, worst_layer=worst_at, mean_layer_rel_err=float(np.mean(mean_rel)),
        out_abs_err=eout, out_mean_abs_err=emean, out_absmax=float(np.abs(yo).max()))
    assert worst < F16_LAYER_BOUND, (worst_at, worst)
    assert eout < F16_OUT_BOUND and emean < 5e-3, (eout, emean)


def test_scnet_batched_groups_equal_single_pairs():
    """n = 2B: every consecutive pair of images is its own BatchNorm group -> same result as B separate calls."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    xs = [torch.from_numpy(oracle_scnet_input(600 + i, ds, mm)).cuda() for i in range(3)]
    yb = net(torch.cat(xs)).clone()
    for i, x in enumerate(xs):
        y1 = net(x)
        assert torch.equal(y1, yb[2 * i:2 * i + 2]), i      # deterministic kernels: bitwise equal


@pytest.mark.parametrize("prec", ["f32", "bf16x9", "bf16x6", "f16x3", "bf16x3"])
def test_scnet_zero_warp_plan_is_bitwise_the_full_forward(prec):
    """Level 0 of the recurrence (util.py:95-96: the identity pose warps to zeros): with channels 8:16 zero in every image the
    RELPOSE_FWD_ZERO_WARP plan -- warped-view streams of conv2 / conv3 on the first BatchNorm group only, their K slices of conv4
    computed once and shared by every image pair -- must give BITWISE the flag-less forward, for the output and for the raw A4 activations; also
    when the zero-warp forward runs first on a fresh workspace (no left-overs of a full forward to hide behind) and for one pair (n = 2:
    the flag is a no-op)."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    net.set_precision(prec)
    xs = [torch.from_numpy(oracle_scnet_input(700 + i, ds, mm)).cuda() for i in range(4)]
    x = torch.cat(xs)
    x[:, 8:] = 0
    yz = net.forward(x, zero_warp=True).clone()                       # first forward of this workspace
    a3z, a4z = net.read_tap("A3").clone(), net.read_tap("A4").clone()
    y = net.forward(x).clone()
    a3, a4 = net.read_tap("A3").clone(), net.read_tap("A4").clone()
    # A3 = six 128-channel stream blocks, odd = warped view: those are computed for the first image pair only (and, where conv4 does
    # not take them as shared K slices, copied to the other images)
    a3z, a3 = a3z.view(8, 56, 56, 6, 128), a3.view(8, 56, 56, 6, 128)
    assert torch.equal(a3z[:, :, :, 0::2], a3[:, :, :, 0::2]) and torch.equal(a3z[:2], a3[:2])
    assert torch.equal(a4z, a4)
    assert torch.equal(yz, y)
    assert torch.equal(net.forward(x, zero_warp=True), y)             # and after a full forward
    y1 = net.forward(x[:2].contiguous(), zero_warp=True)
    assert torch.equal(y1, y[:2])
    log("scnet_zero_warp", prec=prec, images=int(x.shape[0]), bitwise=True)


@pytest.mark.parametrize("prec", ["f32", "bf16x9", "bf16x6", "f16x3", "bf16x3"])
def test_scnet_self_stream_cache_is_bitwise_the_full_forward(prec):
    """Levels >= 1 of the recurrence: channels 0:8 (the masked own views) are those of level 0, only the warped view changed
    (evaluation.py:217-242), and the reference runs the self-view streams as separate module calls with their own batch statistics
    (mymodel.py:266-276).  A forward carrying the tag of the previous forward of its workspace skips the self members of
    conv1 / conv2 / conv3 and conv4's self K slices: output, raw A1..A4 and everything else must be BITWISE the tag-less forward's --
    after a level-0 (zero-warp) forward, after a full forward, as the first forward of a fresh workspace (nothing cached: it must
    compute), after the tag changed (a new batch in a rotating slot), and with the pose-outputs plan."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    net.set_precision(prec)
    xa = torch.cat([torch.from_numpy(oracle_scnet_input(900 + i, ds, mm)).cuda() for i in range(3)])
    xb = torch.cat([torch.from_numpy(oracle_scnet_input(910 + i, ds, mm)).cuda() for i in range(3)])
    taps = ("A1", "A2", "A3", "A4", "D3", "D2")

    def fwd(x, **kw):
        y = net.forward(x, **kw).clone()
        return y, {t: net.read_tap(t).clone() for t in taps}

    def same(got, ref, what):
        assert torch.equal(got[0], ref[0]), what
        for t in taps:
            assert torch.equal(got[1][t], ref[1][t]), (what, t)

    # a pass over batch a: level 0 = zero warp, levels 1 and 2 = two different warped views
    x0 = xa.clone(); x0[:, 8:] = 0
    x1 = xa.clone()
    x2 = xa.clone(); x2[:, 8:] = xb[:, 8:]
    ref1, ref2 = fwd(x1), fwd(x2)                                      # tag-less: always the full forward
    t1 = net.new_self_tag()
    same(fwd(x1, self_tag=t1), ref1, "first forward with a new tag computes everything")
    same(fwd(x2, self_tag=t1), ref2, "cached after a full forward")
    same(fwd(x1, self_tag=t1), ref1, "cached again")
    t2 = net.new_self_tag()
    fwd(x0, zero_warp=True, self_tag=t2)                               # level 0 fills the cache (its warped streams: first pair only)
    same(fwd(x1, self_tag=t2), ref1, "cached after the level-0 plan")
    same(fwd(x2, self_tag=t2), ref2, "second cached level")
    # another batch takes the workspace (new tag): its self streams must be recomputed, then cached
    refb = fwd(xb)
    xb2 = xb.clone(); xb2[:, 8:] = xa[:, 8:]
    refb2 = fwd(xb2)
    t3 = net.new_self_tag()
    same(fwd(xb, self_tag=t3), refb, "new tag after another batch's cached forwards")
    same(fwd(xb2, self_tag=t3), refb2, "cached for the new batch")
    # a tag-less forward of other data in between invalidates the cache (tag 0 never matches)
    fwd(xa)
    same(fwd(xb2, self_tag=t3), refb2, "tag-less forward in between: recompute")
    # pose-outputs plan combined with the cache
    yp = net.forward(xb, outputs="pose", self_tag=t3)
    assert torch.equal(yp[:, 3:7], refb[0][:, 3:7]) and torch.equal(yp[:, 7 + S:], refb[0][:, 7 + S:])
    # a fresh workspace (another batch size) whose first forward already carries a known tag
    y2 = net.forward(xb[:2].contiguous(), self_tag=t3)
    assert torch.equal(y2, refb[0][:2])
    # the workspace is dropped and re-created at a recycled address with other contents: the tag must not be trusted (RELPOSE_FWD_NEW_WORKSPACE)
    t5 = net.new_self_tag()
    same(fwd(x1, self_tag=t5), ref1, "fill before the workspace is dropped")
    nbytes = net._ws.numel()
    net._wss.clear(); net._ws = None
    poison = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")       # NaN patterns where the cache was
    torch.cuda.synchronize()
    del poison
    same(fwd(x2, self_tag=t5), ref2, "same tag on a re-created workspace: full forward")
    same(fwd(x1, self_tag=t5), ref1, "and cached again afterwards")
    log("scnet_self_stream_cache", prec=prec, images=int(xa.shape[0]), bitwise=True)


def test_scnet_pose_outputs_are_bitwise_the_full_forward_on_the_pose_channels():
    """RELPOSE_FWD_POSE_OUTPUTS (opt-in): the decoder branches of the rgb and semantic heads (mymodel.py:312-316,364-368) feed nothing the
    pose loop reads (evaluation.py:246-253 takes normal 3:6, depth 6, features 7+S:); without them those channels must be BITWISE the
    full forward's and the skipped channels zeros -- alone and combined with the level-0 plan."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    x = torch.cat([torch.from_numpy(oracle_scnet_input(800 + i, ds, mm)).cuda() for i in range(2)])
    y = net.forward(x).clone()
    yp = net.forward(x, outputs="pose").clone()
    assert torch.equal(yp[:, 3:7], y[:, 3:7]) and torch.equal(yp[:, 7 + S:], y[:, 7 + S:])
    assert float(yp[:, :3].abs().max()) == 0.0 and float(yp[:, 7:7 + S].abs().max()) == 0.0
    x[:, 8:] = 0
    y0 = net.forward(x).clone()
    yp0 = net.forward(x, zero_warp=True, outputs="pose")
    assert torch.equal(yp0[:, 3:7], y0[:, 3:7]) and torch.equal(yp0[:, 7 + S:], y0[:, 7 + S:])
    log("scnet_pose_outputs", bitwise=True)


def test_scnet_is_a_torch_module_with_the_reference_state_dict():
    """The reference's call sites treat the network as a torch.nn.Module (isinstance, .cuda(), .eval(), state_dict()): the shim is one,
    its state_dict() returns the loaded parameters under the reference's key names (mymodel.py:142-257)."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, sd = make_net(S, tanh, seed)
    assert isinstance(net, torch.nn.Module) and net.cuda() is net and net.eval() is net and net.to("cuda") is net
    got = net.state_dict()
    assert set(got) == set(sd) and all(np.array_equal(got[k].numpy(), np.asarray(sd[k], dtype=np.float32)) for k in sd)
    x = torch.from_numpy(oracle_scnet_input(601, ds, mm)).cuda()
    assert torch.equal(net(x), net.forward(x))


def test_scnet_rejects_odd_batch_like_reference():
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 16, 160, 640, device="cuda"))


@pytest.mark.parametrize("mode,bound_max,bound_mean", [("bf16x9", 5e-4, 2e-5), ("bf16x6", 5e-4, 2e-5), ("f16x3", 5e-4, 2e-5), ("bf16x3", 2e-3, 1e-4), ("f16", 1e-1, 5e-3)])
@pytest.mark.parametrize("hw", [(160, 640), (320, 1280)])
def test_scnet_split_precision_options_close_to_f32_and_reversible(hw, mode, bound_max, bound_mean):
    """relpose_scnet_set_precision(F16X3 / BF16X3): split 16-bit MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate) are
    OPT-INs, not the parity configuration: outputs stay close to the fp32 kernels (measured max / mean abs on O(1-5)
    outputs: f16x3 1e-4 / 3e-6, bf16x3 7e-4 / 2e-5), are batch-invariant like the fp32 path, and switching back restores
    fp32 bit for bit.  (320, 1280) is BASELINE configs[4]'s resolution, whose 16-bit MFMA conv path these options are.)"""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, _ = make_net(S, tanh, seed)
    H, W = hw
    torch.manual_seed(3)
    x = torch.randn(4, 16, H, W, device="cuda")
    y32 = net(x).clone()
    net.set_precision(mode)
    y16 = net(x).clone()
    y16_single = net(x[2:4]).clone()
    net.set_precision("f32")
    y32b = net(x)
    d = (y16 - y32).abs()
    log("scnet_split_precision", mode=mode, hw=list(hw), max_abs=float(d.max()), mean_abs=float(d.mean()), out_abs_mean=float(y32.abs().mean()))
    assert float(d.max()) < bound_max and float(d.mean()) < bound_mean
    assert float(d.max()) > 0                                   # the option really ran a different kernel
    assert torch.equal(y16_single, y16[2:4])                    # batch-invariant
    assert torch.equal(y32b, y32)                               # fp32 path untouched


def test_scnet_reload_state_dict_rebuilds_launch_plans():
    """load_state_dict twice on ONE net (after a forward has cached launch plans that hold absolute weight pointers):
    each result must equal a fresh net with that state dict (plans were not invalidated by finalize)."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    x = torch.from_numpy(oracle_scnet_input(610, ds, mm)).cuda()
    netA, sdA = make_net(S, tanh, seed)
    netB, sdB = make_net(S, tanh, seed + 50)
    yA, yB = netA(x).clone(), netB(x).clone()
    assert not torch.equal(yA, yB)
    netA.load_state_dict(sdB)                 # same object, same workspace, new weights
    assert torch.equal(netA(x), yB)
    netA.load_state_dict({"module." + k: v for k, v in sdA.items()})       # DataParallel-prefixed keys
    assert torch.equal(netA(x), yA)


@pytest.mark.parametrize("hw", [(56, 224), (80, 320), (64, 250)])
def test_scnet_small_panoramas_resize_out_generic_path(hw):
    """W < 560: the 64-pixel output segments of resize_out span more source pixels than its LDS staging holds (ADVICE r1:
    silent overrun); the kernel now gathers directly for such shapes.  Whole forward vs the fp32 oracle."""
    import torch
    tag, S, tanh, seed, ds, mm = SCNET_CASES[0]
    net, sd = make_net(S, tanh, seed)
    H, W = hw
    rs = np.random.RandomState(H)
    x = rs.randn(2, 16, H, W).astype(np.float32)
    want = SCNetOracle(sd, S, tanh).forward_pairs(x).numpy()
    got = net(torch.from_numpy(x).cuda()).cpu().numpy()
    err = float(np.abs(got - want).max())
    log("scnet_small_pano", hw=list(hw), max_abs_err=err)
    assert got.shape == want.shape and err < 5e-4

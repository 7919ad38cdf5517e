"""SCNet parameter inventory + a portable seeded generator.

Key names and shapes follow the reference module tree (model/mymodel.py:142-257):
``<block>.0.weight`` conv / transposed-conv weight (no bias when BatchNorm
follows), ``<block>.1.weight|bias`` BatchNorm affine, ``deconv1{rgb,n,d,s,f}``
1x1 heads with bias.  ConvTranspose2d weights are ``[Cin, Cout, kh, kw]``.

The real checkpoints are Google-Drive downloads that do not ship with the
reference, so tests and the bench use ``make_state_dict`` (RandomState driven,
fixed key order, independent of torch's RNG so the same numbers come out on any
box).  A real ``state_dict`` with these keys loads the same way.
"""
from collections import OrderedDict

import numpy as np

NGF = 64


HEADS = ("rgb", "n", "d", "s", "f")          # the reference's concatenation order (mymodel.py:309-376)


def parse_output_type(output_type):
    """args.outputType -> tuple of head names in the reference's order.  The reference tests substrings ('rgb' in outputType, 'n' in
    outputType, ...: mymodel.py:189-243), so 'rgbdnsf', 'dnf', 'sf' are all valid spellings; 'k' names a head whose forward reads an
    undefined variable (mymodel.py:328) and is rejected."""
    if "k" in output_type:
        raise ValueError("outputType 'k': the reference's keypoint head reads an undefined xsift (mymodel.py:328); it cannot run there either")
    heads = tuple(h for h in HEADS if h in output_type)
    if not heads:
        raise ValueError(f"outputType {output_type!r} selects no head")
    return heads


def head_channels(snumclass=15):
    return {"rgb": 3, "n": 3, "d": 1, "s": snumclass, "f": 32}


def layer_table(snumclass=15, skip_layer=1, output_type="rgbdnsf"):
    """[(name, kind, cin, cout, k, stride, pad)] in forward order; kind in
    {'conv','deconv','head'}."""
    g = NGF
    sm = 2 if skip_layer else 1
    heads = parse_output_type(output_type)
    if not skip_layer and any(h in heads for h in ("rgb", "n", "d")):
        raise ValueError("skipLayer=0 with an rgb / n / d head: the reference's 1x1 output convs take 64 channels, 32 of them the skip "
                         "(mymodel.py:192 vs :347) -- that combination fails inside torch")
    t = []
    for m, cin in (("rgb", 4), ("n", 4), ("d", 2)):
        t += [(f"conv1{m}", "conv", cin, g // 2, 3, 1, 1),
              (f"conv2{m}", "conv", g // 2, g, 4, 2, 1),
              (f"conv3{m}", "conv", g, g * 2, 4, 2, 1)]
    t += [("conv4", "conv", g * 12, g * 4, 4, 2, 1),
          ("conv5", "conv", g * 4, g * 8, 4, 2, 1),
          ("conv6", "conv", g * 8, g * 8, 4, 2, 1),
          ("conv7", "conv", g * 8, g * 8, 3, 2, 0),
          ("conv8", "conv", g * 8, g * 8, 3, 1, 1),
          ("conv9", "conv", g * 8, g * 16, 3, 1, 0),
          ("deconv9", "deconv", g * 16, g * 8, 3, 1, 0),
          ("deconv8", "deconv", g * 8 * sm, g * 8, 3, 1, 1),
          ("deconv7", "deconv", g * 8 * sm, g * 8, 3, 2, 0),
          ("deconv6", "deconv", g * 8 * sm, g * 8, 4, 2, 1),
          ("deconv5", "deconv", g * 8 * sm, g * 4, 4, 2, 1),
          ("deconv4", "deconv", g * 4 * sm, g * 2, 4, 2, 1)]
    for m, cout in (("rgb", 3), ("n", 3), ("d", 1)):
        if m in heads:
            t += [(f"deconv3{m}", "deconv", g * 4, g, 4, 2, 1),
                  (f"deconv2{m}", "deconv", g * 2, g // 2, 4, 2, 1),
                  (f"deconv1{m}", "head", g, cout, 1, 1, 0)]
    for m, cout in (("s", snumclass), ("f", 32)):
        if m in heads:
            t += [(f"deconv3{m}", "deconv", g * 2, g, 4, 2, 1),
                  (f"deconv2{m}", "deconv", g, g, 4, 2, 1),
                  (f"deconv1{m}", "head", g, cout, 1, 1, 0)]
    return t


def state_dict_spec(snumclass=15, batchnorm=1, skip_layer=1, output_type="rgbdnsf"):
    """OrderedDict key -> shape, in the reference's registration order.  batchnorm=0 (mymodel.py:22-25, 35-38): the blocks are
    Sequential(conv(bias=True), LeakyReLU) -- '<block>.0.bias' instead of the BatchNorm affine '<block>.1.weight|bias'."""
    spec = OrderedDict()
    for name, kind, cin, cout, k, s, p in layer_table(snumclass, skip_layer, output_type):
        if kind == "head":
            spec[f"{name}.weight"] = (cout, cin, 1, 1)
            spec[f"{name}.bias"] = (cout,)
            continue
        spec[f"{name}.0.weight"] = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
        if batchnorm:
            spec[f"{name}.1.weight"] = (cout,)
            spec[f"{name}.1.bias"] = (cout,)
        else:
            spec[f"{name}.0.bias"] = (cout,)
    return spec


def make_state_dict(seed=0, snumclass=15, batchnorm=1, skip_layer=1, output_type="rgbdnsf"):
    """Xavier-normal conv weights, BN gamma ~ N(1,.02), beta = small noise,
    head bias ~ N(0,.05) -- the reference's init (mymodel.py:6-13) with non-zero
    betas/biases so that every parameter is exercised by parity tests."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape in state_dict_spec(snumclass, batchnorm, skip_layer, output_type).items():
        if len(shape) == 4:
            rf = shape[2] * shape[3]
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            std = np.sqrt(2.0 / (fan_in + fan_out))
            sd[key] = (rs.randn(*shape) * std).astype(np.float32)
        elif key.endswith(".1.weight"):
            sd[key] = (1.0 + 0.02 * rs.randn(*shape)).astype(np.float32)
        elif key.endswith(".1.bias"):
            sd[key] = (0.05 * rs.randn(*shape)).astype(np.float32)
        else:
            sd[key] = (0.05 * rs.randn(*shape)).astype(np.float32)
    return sd


def num_params(snumclass=15):
    return int(sum(np.prod(s) for s in state_dict_spec(snumclass).values()))


def make_descriptor_state_dict(seed=0, snumclass=15):
    """``make_state_dict`` with the feature head cut down to a LOCAL function of the colour streams: conv4 reads only
    the own-view and warped-view rgb encoders (normal / depth streams: zero weights), and deconv4 reads only its
    conv4 skip input (the deep conv5..deconv5 path: zero weights).  On the synthetic rooms (view-invariant wall
    texture) the 32-d descriptors then agree for the same world point seen from two nearby cameras, which makes the
    matching problem of tests/golden "wc" fixtures well-conditioned.  Every layer still runs; only values change."""
    sd = make_state_dict(seed, snumclass)
    g = NGF
    w = sd["conv4.0.weight"]                     # [256, 768, 4, 4]; inputs: rgb, rgb_t2s, n, n_t2s, d, d_t2s (128 each)
    w[:, 2 * 2 * g:] = 0
    w[:, :2 * 2 * g] *= np.float32(np.sqrt(3.0))
    w = sd["deconv4.0.weight"]                   # [512, 128, 4, 4]; inputs: deconv5 output (256), conv4 skip (256)
    w[:4 * g] = 0
    # the three stride-2 transposed convs of the feature path become (bilinear x2 upsampling) o (random channel mix):
    # random 4x4 taps would give every output-pixel parity its own filter, i.e. a descriptor that depends on the pixel
    # position modulo 8 more than on the surface it sees
    u = np.array([0.25, 0.75, 0.75, 0.25], np.float32)
    for key in ("deconv4.0.weight", "deconv3f.0.weight", "deconv2f.0.weight"):
        w = sd[key]
        mix = w[:, :, 1, 1] * np.float32(4.0)
        w[...] = mix[:, :, None, None] * (u[:, None] * u[None, :])[None, None]
    return sd


def load_checkpoint(path, map_location="cpu"):
    """The reference's checkpoint container (evaluation.py:143-153: ``torch.load('...comp.pth.tar')['state_dict']``;
    written by train_op.save_checkpoint from ``netG.state_dict()``, optionally under torch.nn.DataParallel, i.e. with
    ``module.``-prefixed keys, mainPanoCompletion2view.py:154-156).  Returns {reference key: float32 numpy array}; keys of
    sub-modules this build does not run (e.g. a discriminator) are dropped by SCNet.load_state_dict, not here."""
    import torch
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    out = OrderedDict()
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        out[k] = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
    return out

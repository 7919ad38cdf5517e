"""ORACLE (test infrastructure, not product code): torch-CPU float32 restatement
of SCNet.forward, the completion + feature network.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  Follows /root/reference/model/mymodel.py: block builders :15-39, module
tree :142-257, forward :259-380.  Defaults ``skipLayer=1``, ``batchnorm=1``,
``outputType='rgbdnsf'`` are the configuration evaluation.py runs; the other
constructor variants (``batchnorm=0``: conv + bias + LeakyReLU, :22-25;
``skipLayer=0``: the decoder without concatenations, :335-340; ``outputType``:
which heads exist, :189-243) are restated too (round 6).  BatchNorm
layers are built with track_running_stats=False (:19,32) so batch statistics
are used at inference; the net is always fed a batch of 2 (evaluation.py:242).
Pinned against the reference module (same state_dict) by make_golden.py.

This is a functional restatement driven by a plain ``{key: array}`` state
dict; it keeps a float32 torch reference because the kernel is floating point.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


class SCNetOracle:
    def __init__(self, state_dict, snumclass=15, use_tanh=1, batchnorm=1, skip_layer=1, output_type="rgbdnsf"):
        self.p = {k: _t(v).float() for k, v in state_dict.items()}
        self.S = snumclass
        self.use_tanh = use_tanh
        self.batchnorm, self.skip, self.output_type = batchnorm, skip_layer, output_type
        self.taps = {}

    def _bn_act(self, x, name):
        if self.batchnorm:                                                      # mymodel.py:16-21 / :29-34
            x = F.batch_norm(x, None, None, self.p[f"{name}.1.weight"], self.p[f"{name}.1.bias"], True, 0.1, 1e-5)
        return F.leaky_relu(x, 0.1)                                             # (:22-25 / :35-38: the conv carried its bias)

    def conv(self, x, name, stride, pad):
        y = F.conv2d(x, self.p[f"{name}.0.weight"], None if self.batchnorm else self.p[f"{name}.0.bias"], stride, pad)
        self.taps[name] = y
        return self._bn_act(y, name)

    def deconv(self, x, name, stride, pad):
        y = F.conv_transpose2d(x, self.p[f"{name}.0.weight"], None if self.batchnorm else self.p[f"{name}.0.bias"], stride, pad)
        self.taps[name] = y
        return self._bn_act(y, name)

    def head(self, x, name):
        return F.conv2d(x, self.p[f"{name}.weight"], self.p[f"{name}.bias"])

    def forward(self, x):
        """x [n,16,H,W] f32 (n = 2 per scan pair; for n>2 BatchNorm would pool
        over all n, so callers loop over pairs) -> [n,7+S+32,H,W]."""
        x = _t(x).float()
        in_shape = x.shape[2:]
        x = F.interpolate(x, [224, 224], mode='bilinear', align_corners=False)
        cat = torch.cat
        enc = {}
        for m, ch in (("rgb", (0, 3)), ("n", (3, 6)), ("d", (6, 7))):
            for s, off in (("", 0), ("_t2s", 8)):
                xi = cat((x[:, off + ch[0]:off + ch[1]], x[:, off + 7:off + 8]), 1)
                x1 = self.conv(xi, f"conv1{m}", 1, 1)
                x2 = self.conv(x1, f"conv2{m}", 2, 1)
                x3 = self.conv(x2, f"conv3{m}", 2, 1)
                enc[m + s] = (x1, x2, x3)
        xin = cat([enc[k][2] for k in ("rgb", "rgb_t2s", "n", "n_t2s", "d", "d_t2s")], 1)
        x4 = self.conv(xin, "conv4", 2, 1)
        x5 = self.conv(x4, "conv5", 2, 1)
        x6 = self.conv(x5, "conv6", 2, 1)
        x7 = self.conv(x6, "conv7", 2, 0)
        x8 = self.conv(x7, "conv8", 1, 1)
        x9 = self.conv(x8, "conv9", 1, 0)
        sk = (lambda a, b: cat((a, b), 1)) if self.skip else (lambda a, b: a)   # mymodel.py:302-307 vs :335-340
        d9 = self.deconv(x9, "deconv9", 1, 0)
        d8 = self.deconv(sk(d9, x8), "deconv8", 1, 1)
        d7 = self.deconv(sk(d8, x7), "deconv7", 2, 0)
        d6 = self.deconv(sk(d7, x6), "deconv6", 2, 1)
        d5 = self.deconv(sk(d6, x5), "deconv5", 2, 1)
        d4 = self.deconv(sk(d5, x4), "deconv4", 2, 1)
        outs = []
        for m in ("rgb", "n", "d"):
            if m not in self.output_type:
                continue
            x1, x2, x3 = enc[m]
            d3 = self.deconv(sk(d4, x3), f"deconv3{m}", 2, 1)
            d2 = self.deconv(sk(d3, x2), f"deconv2{m}", 2, 1)
            # (skipLayer=0, :347: the 1x1 conv takes 64 channels and gets 32 -- torch raises there, and so does this head())
            outs.append(self.head(sk(d2, x1), f"deconv1{m}"))
        for m in ("s", "f"):
            if m not in self.output_type:
                continue
            d3 = self.deconv(d4, f"deconv3{m}", 2, 1)
            d2 = self.deconv(d3, f"deconv2{m}", 2, 1)
            o = self.head(d2, f"deconv1{m}")
            if m == "f" and self.use_tanh:
                o = torch.tanh(o)
            outs.append(o)
        self.taps["out224"] = cat(outs, 1)
        return F.interpolate(self.taps["out224"], in_shape, mode='bilinear', align_corners=False)

    def forward_pairs(self, x):
        """x [2B,16,H,W]: consecutive samples (2b,2b+1) form one BN group."""
        with torch.no_grad():
            return torch.cat([self.forward(x[i:i + 2]) for i in range(0, x.shape[0], 2)])

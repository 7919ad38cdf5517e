"""CPU: the index algebra behind the round-2 tile kernels of csrc/scnet.hip, checked against torch's own convolutions.

* conv_s2_tile_kernel / conv_s2_strip_kernel: a 4x4 stride-2 pad-1 convolution (model/mymodel.py:15-24, conv2 ... conv9) equals the sum
  over the four input parity planes (p, q) of a 2x2 STRIDE-1 convolution: plane pixel (a, b) = in(2a + p, 2b + q), taps
  ky = p ? {0, 2} : {1, 3}, output (y, x) reads plane rows a = y - p + ty (ty = 0, 1); the kernels' padded-plane local row is r = a + p,
  i.e. input row 2r - p, and the strip kernel's linear position is P = (img (Hp + 1) + y)(Wp + 1) + x with tap offsets {0, 1, Wp + 1, Wp + 2}.
* deconv_tile_kernel: a 4x4 stride-2 pad-1 transposed convolution (mymodel.py:26-39) splits into 4 sub-pixel phases, each a 2x2-tap
  stride-1 correlation over the 3x3 neighbourhood of the input pixel.
"""
import numpy as np
import torch
import torch.nn.functional as F


def test_stride2_conv_equals_four_parity_plane_convs():
    torch.manual_seed(0)
    n, cin, cout, H = 2, 3, 5, 12                     # H even: Hp = H / 2
    x = torch.randn(n, cin, H, H, dtype=torch.float64)
    w = torch.randn(cout, cin, 4, 4, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=2, padding=1)
    Hp = H // 2
    out = torch.zeros_like(ref)
    for p in range(2):
        for q in range(2):
            # padded plane: local (r, c), r in [0, Hp], c in [0, Hp]  <->  input pixel (2r - p, 2c - q); outside the image = 0
            plane = torch.zeros(n, cin, Hp + 1, Hp + 1, dtype=torch.float64)
            for r in range(Hp + 1):
                for c in range(Hp + 1):
                    iy, ix = 2 * r - p, 2 * c - q
                    if 0 <= iy < H and 0 <= ix < H:
                        plane[:, :, r, c] = x[:, :, iy, ix]
            for ty in range(2):
                for tx in range(2):
                    ky = 2 * ty if p else 1 + 2 * ty
                    kx = 2 * tx if q else 1 + 2 * tx
                    # output (y, x) reads local (y + ty, x + tx): a shifted window of the padded plane
                    win = plane[:, :, ty:ty + Hp, tx:tx + Hp]
                    out += torch.einsum("nchw,oc->nohw", win, w[:, :, ky, kx])
    assert torch.allclose(out, ref, rtol=0, atol=1e-12)


def test_strip_positions_are_uniform_tap_offsets():
    """conv_s2_strip_kernel: with P(img, y, x) = (img (Hp + 1) + y)(Wp + 1) + x every tap of every output is P + {0, 1, Wp + 1, Wp + 2},
    and a run of 128 consecutive outputs touches at most 127 + ceil(127 / Wp) + (Wp + 1) [one image crossing] + (Wp + 1) + 2 positions."""
    for Hp, Wp in ((28, 28), (14, 14)):
        hw, W1, H1 = Hp * Wp, Wp + 1, Hp + 1
        pos = lambda m: ((m // hw) * H1 + (m % hw) // Wp) * W1 + (m % hw) % Wp
        M = 6 * hw
        worst = 0
        for m0 in range(0, M - 128, 7):
            lo, hi = pos(m0), pos(m0 + 127) + W1 + 1
            worst = max(worst, hi - lo + 1)
            for m in (m0, m0 + 63, m0 + 127):
                img, y, x = m // hw, (m % hw) // Wp, (m % hw) % Wp
                for ty in range(2):
                    for tx in range(2):
                        assert (img * H1 + y + ty) * W1 + x + tx == pos(m) + ty * W1 + tx
        bound = 127 + (127 + Wp - 1) // Wp + W1 + W1 + 2
        assert worst <= bound <= 224, (Hp, worst, bound)


def test_transposed_conv_phases_read_the_3x3_neighbourhood():
    torch.manual_seed(1)
    n, cin, cout, H = 2, 3, 4, 7
    x = torch.randn(n, cin, H, H, dtype=torch.float64)
    w = torch.randn(cin, cout, 4, 4, dtype=torch.float64)            # ConvTranspose2d weight layout [Cin, Cout, k, k]
    ref = F.conv_transpose2d(x, w, stride=2, padding=1)              # [n, cout, 2H, 2H]
    xp = F.pad(x, (1, 1, 1, 1))                                      # halo tile: origin (+1, +1), zero padding stored as zeros
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            # out(2y + py, 2x + px) = sum over ky with (py + 1 - ky) % 2 == 0 of in(y + (py + 1 - ky) / 2, ...) w(ky, kx)
            for ky in range(4):
                if (py + 1 - ky) % 2:
                    continue
                oy = (py + 1 - ky) // 2
                for kx in range(4):
                    if (px + 1 - kx) % 2:
                        continue
                    ox = (px + 1 - kx) // 2
                    assert -1 <= oy <= 1 and -1 <= ox <= 1           # all 16 (phase, tap) products stay inside the 3x3 neighbourhood
                    win = xp[:, :, 1 + oy:1 + oy + H, 1 + ox:1 + ox + H]
                    out[:, :, py::2, px::2] += torch.einsum("nchw,co->nohw", win, w[:, :, ky, kx])
    assert torch.allclose(out, ref, rtol=0, atol=1e-12)

"""Host-side mirror of the reference network module (model/mymodel.py:141-380).

``SCNet(args)`` keeps the reference constructor (reads ``args.batchnorm``,
``useTanh``, ``skipLayer``, ``outputType``, ``snumclass``), ``load_state_dict``
takes the reference's key names, and ``net(x)`` maps ``[n,16,H,W]`` to
``[n,7+S+32,H,W]`` -- but every layer runs in the HIP library
(csrc/scnet.hip): there is no torch.nn graph behind it and no CPU path.

BatchNorm statistics are taken over consecutive groups of 2 images, which is
what the reference computes for its fixed batch of 2 (evaluation.py:242); with
n = 2B the call processes B scan pairs at once.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .weights import HEADS, head_channels, parse_output_type, state_dict_spec

# relpose_scnet_set_precision modes (include/relpose.h: RELPOSE_PREC_*)
PRECISION_CODES = {"f32": 0, "bf16x3": 1, "f16x3": 2, "f16": 3, "bf16x9": 4, "bf16x6": 5}

BUFFER_SHAPES = {"X0": (224, 16), "A1": (224, 192), "A2": (112, 384), "A3": (56, 768), "A4": (28, 256), "A5": (14, 512),
                 "A6": (7, 512), "A7": (3, 512), "A8": (3, 512), "A9": (1, 1024), "D9": (3, 512), "D8": (3, 512),
                 "D7": (7, 512), "D6": (14, 512), "D5": (28, 256), "D4": (56, 128), "D3": (112, 320), "D2": (224, 224)}


_REGISTRY = {}          # C handle -> weakref(SCNet): lets torch.ops.relpose.scnet_forward(x, net.handle) find the workspace cache


class SCNet(torch.nn.Module):
    """A torch.nn.Module like the reference's (isinstance checks, .to() / .cuda() / .eval() / .state_dict() call sites keep working) whose
    parameters live in the HIP library, not in torch: `state_dict()` returns the loaded tensors (host copies), `parameters()` is empty."""

    def __init__(self, args):
        super().__init__()
        self.snumclass = int(args.snumclass)
        self.useTanh = int(getattr(args, "useTanh", 1))
        # the constructor variants (mymodel.py:145-149, 189-243).  What the reference itself cannot run raises ValueError here: 'k' in
        # outputType (its forward reads an undefined xsift, :328) and skipLayer=0 with an rgb / n / d head (64-channel 1x1 convs fed 32, :347)
        self.batchnorm = int(bool(getattr(args, "batchnorm", 1)))
        self.skipLayer = int(bool(getattr(args, "skipLayer", 1)))
        self.outputType = str(getattr(args, "outputType", "rgbdnsf"))
        self.heads = parse_output_type(self.outputType)
        self._spec = state_dict_spec(self.snumclass, self.batchnorm, self.skipLayer, self.outputType)
        hc = head_channels(self.snumclass)
        # the library keeps the full layout [rgb 3 | n 3 | d 1 | s S | f 32] (absent heads = 0); the reference concatenates the heads that exist
        off, self._sel = 0, []
        for h in HEADS:
            if h in self.heads:
                self._sel += list(range(off, off + hc[h]))
            off += hc[h]
        self.full_channels = off
        self.out_channels = len(self._sel)
        self.is_variant = not (self.batchnorm and self.skipLayer and self.heads == HEADS)
        if self.is_variant:
            cfg = _lib.SCNetConfig()
            cfg.struct_size = C.sizeof(_lib.SCNetConfig)
            cfg.snumclass, cfg.use_tanh, cfg.batchnorm, cfg.skip_layer = self.snumclass, self.useTanh, self.batchnorm, self.skipLayer
            cfg.output_mask = sum(1 << i for i, h in enumerate(HEADS) if h in self.heads)
            self._h = _lib.lib().relpose_scnet_create_ex(C.byref(cfg))
        else:
            self._h = _lib.lib().relpose_scnet_create(self.snumclass, self.useTanh)
        if not self._h:
            raise RuntimeError("relpose_scnet_create failed")
        self._state = {}
        self._wss = {}          # one workspace (and launch plan) per (stream, n): streams must not share scratch
        self._ws = None
        self._loaded = False
        import weakref
        _REGISTRY[int(self._h)] = weakref.ref(self)

    @property
    def handle(self):
        """The RelposeSCNet* as an int: the ``net_handle`` argument of torch.ops.relpose.scnet_forward."""
        return int(self._h)

    @staticmethod
    def from_handle(handle):
        ref = _REGISTRY.get(int(handle))
        net = ref() if ref is not None else None
        if net is None:
            raise RuntimeError(f"relpose: no live SCNet with handle {handle}")
        return net

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _REGISTRY.pop(int(self._h), None)
                _lib.lib().relpose_scnet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # (nn.Module's own cuda() / to() / eval() / train() apply: there are no torch parameters to move; the weights were uploaded by
    # load_state_dict and the kernels always use batch statistics, like the reference's track_running_stats=False BatchNorm)
    def state_dict(self, *args, **kwargs):
        """The loaded parameters under the reference's key names (host tensors)."""
        return dict(self._state)

    def load_state_dict(self, state_dict, strict=True):
        """Takes the reference's key names (model/mymodel.py:142-257); a ``module.`` prefix (checkpoints saved from a
        torch.nn.DataParallel wrapper, mainPanoCompletion2view.py:154-156) is stripped.  May be called again with a
        different state dict (cached launch plans are rebuilt)."""
        L = _lib.lib()
        spec = self._spec
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing = [k for k in spec if k not in state_dict]
        if missing and strict:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing[:4]}...")
        for k, shape in spec.items():
            v = state_dict[k]
            a = np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {a.shape} vs {shape}")
            _lib.check(L.relpose_scnet_set_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_param {k}")
            self._state[k] = torch.from_numpy(a)
        _lib.require_gpu()
        _lib.check(L.relpose_scnet_finalize(self._h), "relpose_scnet_finalize")
        self._loaded = True
        return self

    def load_checkpoint(self, path):
        """evaluation.py:143-153: ``torch.load(path)['state_dict']`` -> load_state_dict."""
        from .weights import load_checkpoint
        return self.load_state_dict(load_checkpoint(path))

    def set_precision(self, mode):
        """'f32' (default, the parity configuration), 'bf16x3' or 'f16x3' (split 16-bit MFMA products, fp32 accumulation) or 'f16'
        (plain fp16 MFMA products, fp32 accumulation): relpose_scnet_set_precision.  Not part of the reference interface."""
        code = PRECISION_CODES[mode]
        _lib.check(_lib.lib().relpose_scnet_set_precision(self._h, code), "relpose_scnet_set_precision")
        return self

    def num_params(self):
        return int(_lib.lib().relpose_scnet_num_params(self._h))

    MAX_WORKSPACES = 8          # cached workspaces (148 MB per image each); the C side caches 16 launch plans keyed by (n, workspace)

    _ws_generation = 0          # names every workspace ALLOCATION of the process (RelposeForwardArgs.workspace_generation)

    def _workspace(self, n, H, W, dev, ws_key=None, also_streams=()):
        import torch
        nbytes = _lib.lib().relpose_scnet_workspace_bytes(self._h, n, H, W)
        if nbytes == 0:
            raise RuntimeError("relpose_scnet_workspace_bytes: invalid shape (n must be even) or weights not loaded")
        key = (torch.cuda.current_stream().cuda_stream if ws_key is None else ("key", ws_key), n, dev.index)
        ent = self._wss.pop(key, None)                     # (re-inserted below: the dict keeps least-recently-used order)
        if ent is None or ent[0].numel() < nbytes:
            # a fresh allocation (maybe at a recycled address): it gets a new generation id, which EVERY call on it carries -- the library
            # uses the self-stream cache of a workspace pointer only under the generation that filled it, whoever touches the new block first
            SCNet._ws_generation += 1
            ent = (torch.empty(nbytes, dtype=torch.uint8, device=dev), SCNet._ws_generation)
            while len(self._wss) >= self.MAX_WORKSPACES:
                # evict ONE entry, the least recently used.  Dropping it is stream-safe: every stream that ever ran a kernel on a
                # workspace was recorded on it (record_stream below), so the caching allocator keeps the block until those kernels
                # have finished, whichever stream asks for memory next.
                self._wss.pop(next(iter(self._wss)))
        self._wss[key] = ent
        ws, self._ws_gen = ent
        ws.record_stream(torch.cuda.current_stream())
        for s in also_streams:
            ws.record_stream(s)
        self._ws = ws
        return ws

    _tag_counter = 0

    @classmethod
    def new_self_tag(cls):
        """A fresh non-zero tag for forward(self_tag=...): draw one whenever channels 0:8 of the input are (re)written."""
        cls._tag_counter += 1
        return cls._tag_counter

    FLAG_ZERO_WARP, FLAG_POSE_OUTPUTS = 1, 2       # RELPOSE_FWD_* (include/relpose.h)

    def forward(self, x, out=None, tail_stream=None, ws_key=None, zero_warp=False, outputs="all", self_tag=0):
        """tail_stream (a torch stream; not part of the reference interface): run the HBM-bound tail of the forward (heads + final
        resize) there, behind the convolutions on the current stream (RelposeForwardArgs::tail_stream) -- `out` is then valid on tail_stream only.
        ws_key: name of the workspace to use (forwards that may overlap need different workspaces; default: one per stream).
        zero_warp: the caller guarantees x[:, 8:16] == 0 for every image (level 0 of the recurrence: util.warping returns zeros for the
        identity pose, util.py:95-96); the warped-view encoder streams then run once per batch instead of once per image
        (RELPOSE_FWD_ZERO_WARP; bitwise the same output).
        outputs: "all" (the reference's output) or "pose" (RELPOSE_FWD_POSE_OUTPUTS: only what the pose path reads -- normal 3:6, depth 6,
        features 7+S: -- the rgb and semantic channels are zeros and their decoder branches are not run; the other channels are bitwise
        those of the full forward).
        self_tag: non-zero = the caller's name for the content of x[:, 0:8] (the masked own views, constant across the levels of a scan
        pair's recurrence, evaluation.py:217-242): a forward that finds the previous forward of its workspace carried the same tag reuses
        that forward's self-view encoder streams (RelposeForwardArgs::self_tag; bitwise the same output).  0 = always recompute."""
        flags = (self.FLAG_ZERO_WARP if zero_warp else 0) | {"all": 0, "pose": self.FLAG_POSE_OUTPUTS}[outputs]
        return self.forward_flags(x, out, flags, self_tag, tail_stream, ws_key)

    def forward_flags(self, x, out=None, flags=0, self_tag=0, tail_stream=None, ws_key=None):
        """The forward as the C ABI sees it (relpose_scnet_forward_ex): `flags` = RELPOSE_FWD_* bits, `tail_stream` a torch stream, a raw
        HIP stream handle (int) or None.  `forward` and torch.ops.relpose.scnet_forward both end here."""
        import torch
        _lib.require_gpu()
        if not self._loaded:
            raise RuntimeError("load_state_dict first")
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 16
        x = x.contiguous()
        n, _, H, W = x.shape
        if isinstance(tail_stream, int):
            tail_stream = None if tail_stream == 0 else torch.cuda.ExternalStream(tail_stream)
        ws = self._workspace(n, H, W, x.device, ws_key, tuple(s_ for s_ in (tail_stream,) if s_ is not None))
        if tail_stream is not None:
            x.record_stream(tail_stream)                   # (the input resize runs there)
        user_out = None
        if self.is_variant:
            # constructor variants: the library writes its full channel layout (absent heads = 0) on the current stream only; the
            # reference's output is the concatenation of the heads that exist (mymodel.py:378)
            tail_stream, user_out = None, out
            out = torch.empty(n, self.full_channels, H, W, dtype=torch.float32, device=x.device)
        elif out is None:
            out = torch.empty(n, self.out_channels, H, W, dtype=torch.float32, device=x.device)
            if tail_stream is not None:
                out.record_stream(tail_stream)
        else:
            assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n, self.out_channels, H, W)
        a = _lib.ForwardArgs()
        a.struct_size = C.sizeof(_lib.ForwardArgs)
        a.flags = int(flags) & 3
        a.x, a.out = x.data_ptr(), out.data_ptr()
        a.n_images, a.H, a.W = n, H, W
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.stream = torch.cuda.current_stream().cuda_stream
        a.tail_stream = a.stream if tail_stream is None else tail_stream.cuda_stream
        a.self_tag = int(self_tag)
        a.workspace_generation = self._ws_gen
        _lib.check(_lib.lib().relpose_scnet_forward_ex(self._h, C.byref(a)), "relpose_scnet_forward_ex")
        if self.is_variant:
            sel = out if self.out_channels == self.full_channels else out[:, self._sel]
            if user_out is None:
                return sel.contiguous()
            user_out.copy_(sel)
            return user_out
        return out

    def plan_macs(self, n, flags=0, self_cached=False):
        """Multiply-accumulates one forward of n images executes under a plan family (relpose_scnet_plan_macs; host-only)."""
        v = C.c_double()
        _lib.check(_lib.lib().relpose_scnet_plan_macs(self._h, int(n), int(flags), 1 if self_cached else 0, C.byref(v)), "relpose_scnet_plan_macs")
        return v.value

    def read_tap(self, name):
        """Raw (pre-BatchNorm) NHWC activations of buffer ``name`` from the last forward: [n,H,H,C]."""
        import torch
        L = _lib.lib()
        cnt = L.relpose_scnet_read_tap(self._h, name.encode(), None, None, None)
        if cnt < 0:
            raise KeyError(name)
        H, Cc = BUFFER_SHAPES[name] if name != "OUT" else (224, self.full_channels)
        out = torch.empty(cnt, dtype=torch.float32, device=self._ws.device)
        L.relpose_scnet_read_tap(self._h, name.encode(), _lib.ptr(out), _lib.ptr(self._ws), _lib.stream_ptr())
        return out.view(-1, H, H, Cc)

    def profile(self, x, iters=3):
        """(ms in implicit-GEMM kernels, ms in other kernels, #GEMM launches) per forward, HIP events."""
        import torch
        n, _, H, W = x.shape
        ws = self._workspace(n, H, W, x.device)
        out = torch.empty(n, self.full_channels, H, W, dtype=torch.float32, device=x.device)
        g, o, k = C.c_double(), C.c_double(), C.c_int64()
        rc = _lib.lib().relpose_scnet_profile(self._h, _lib.ptr(x.contiguous()), _lib.ptr(out), n, H, W, _lib.ptr(ws), ws.numel(),
                                              iters, C.byref(g), C.byref(o), C.byref(k), _lib.stream_ptr())
        _lib.check(rc, "relpose_scnet_profile")
        return g.value, o.value, k.value

"""Import the upstream reference (read-only at /root/reference) on CPU.

Only usable in the build container: the reference does not travel to the GPU
box.  Used by make_golden.py (fixture generation) and by the optional
``-m refcheck`` tests that assert oracle == reference.  Nothing from the
reference is copied; it is imported in place with three shims:

 1. stub modules for cv2 / open3d / torchvision / progress / tensorboardX /
    ipdb (imported at module top in the reference but unused by the hot path);
 2. ``Tensor.cuda`` made a no-op (hard ``.cuda()`` calls in utils/torch_op.py);
 3. RPModule/rpmodule.py lines 342-343 do not parse as shipped
    (``/ FEAT_SCALING.``); the text is patched in memory before exec.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("RELPOSE_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "RPModule"))


class _Anything(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Anything(self.__name__ + "." + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return None


_loaded = {}


def load():
    """Returns dict(util=..., rputil=..., rpmodule=..., mymodel=..., torch_op=...)."""
    if _loaded:
        return _loaded
    import torch

    for name in ["cv2", "open3d", "torchvision", "torchvision.utils", "torchvision.models",
                 "progress", "progress.bar", "tensorboardX", "ipdb"]:
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import util as ref_util  # noqa
    from utils import torch_op as ref_torch_op  # noqa

    pkg = types.ModuleType("RPModule")
    pkg.__path__ = [os.path.join(REF, "RPModule")]
    sys.modules["RPModule"] = pkg
    spec = importlib.util.spec_from_file_location("RPModule.rputil", os.path.join(REF, "RPModule", "rputil.py"))
    rputil = importlib.util.module_from_spec(spec)
    sys.modules["RPModule.rputil"] = rputil
    spec.loader.exec_module(rputil)
    src = open(os.path.join(REF, "RPModule", "rpmodule.py")).read()
    assert "/ FEAT_SCALING.\n" in src
    src = src.replace("/ FEAT_SCALING.\n", "/ FEAT_SCALING\n")
    rpmodule = types.ModuleType("RPModule.rpmodule")
    rpmodule.__package__ = "RPModule"
    rpmodule.__file__ = os.path.join(REF, "RPModule", "rpmodule.py")
    sys.modules["RPModule.rpmodule"] = rpmodule
    exec(compile(src, rpmodule.__file__, "exec"), rpmodule.__dict__)

    spec = importlib.util.spec_from_file_location("ref_mymodel", os.path.join(REF, "model", "mymodel.py"))
    mymodel = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mymodel)
    _loaded.update(util=ref_util, rputil=rputil, rpmodule=rpmodule, mymodel=mymodel, torch_op=ref_torch_op)
    return _loaded

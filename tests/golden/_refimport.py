"""Import the read-only reference tree (``/root/reference``) on CPU in THIS container.

Used only by ``make_golden.py`` (golden-vector generation) -- never by the tests,
``smoke()`` or ``bench.py``: ``/root/reference`` does not exist on the GPU box.
The reference imports a number of cosmetic third-party modules that are absent
here; they are replaced by auto-attribute stub modules.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = "/root/reference"

_STUBS = [
    "shapely", "shapely.geometry", "open3d", "cv2", "efficientnet_pytorch", "spconv",
    "spconv.pytorch", "torchvision", "torchvision.models", "torchvision.models.resnet",
    "torchvision.models.mnasnet", "torchvision.models.mobilenetv2", "torchvision.models.utils",
    "torchvision.transforms", "torchvision.ops", "timm", "timm.models", "timm.models.layers",
    "easydict", "numba", "pyquaternion", "skimage", "skimage.io", "h5py", "tensorboardX",
    "lzf", "pypcd", "pypcd.pypcd", "matplotlib", "matplotlib.pyplot",
]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def install():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated in the build container")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    ic = types.ModuleType("icecream"); ic.ic = lambda *a, **k: None
    tc = types.ModuleType("termcolor"); tc.colored = lambda s, *a, **k: s
    sys.modules.setdefault("icecream", ic)
    sys.modules.setdefault("termcolor", tc)
    for name in _STUBS:
        try:
            if name.split(".")[0] in ("matplotlib",):
                importlib.import_module(name)
                continue
        except Exception:
            pass
        if name in sys.modules:
            continue
        m = _Stub(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

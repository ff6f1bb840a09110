"""Import the read-only reference (/root/reference) in THIS container to generate golden vectors.

The reference imports seven third-party packages at module top that are absent from this image
(cv2, open3d, torchvision, ema_pytorch, pytorch_fid, imageio, coloredlogs).  None of the arithmetic
we pin lives in them, so they are replaced by empty stub modules in ``sys.modules``; the reference
files themselves are untouched.  This module never travels to the GPU box in a usable form: it
needs /root/reference, which only exists here.  Only tools/make_goldens.py uses it.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Anything(types.ModuleType):
    """A module whose every attribute is another permissive stub (enough for `from x import y`)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Anything(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")


_STUBS = [
    "cv2", "open3d", "torchvision", "torchvision.transforms", "torchvision.utils",
    "ema_pytorch", "pytorch_fid", "pytorch_fid.inception", "pytorch_fid.fid_score",
    "imageio", "coloredlogs", "matplotlib", "matplotlib.pyplot",
]


def import_reference():
    for name in _STUBS:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import denoising_diffusion_pytorch.successive_ddnm_diffusion as sd
    import depth_correction_pytorch.depth_correction as dc
    return sd, dc

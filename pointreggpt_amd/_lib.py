"""ctypes binding of libprg_hip.so (include/prg.h).  No CPU fallback: if the library is missing or a call
fails this raises — the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PRG_HIP_LIB", _HERE / "libprg_hip.so"))

PRG_F32, PRG_BF16, PRG_MXFP8, PRG_F16X3 = 0, 1, 2, 3


class PrgError(RuntimeError):
    pass


class UnetConfigC(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_levels", C.c_int32), ("dim_mults", C.c_int32 * 8),
                ("in_channels", C.c_int32), ("conditional", C.c_int32), ("param_cond_dim", C.c_int32),
                ("groups", C.c_int32), ("sigmoid_out", C.c_int32)]


class StepC(C.Structure):
    _fields_ = [("t", C.c_int32), ("clip_pred", C.c_int32), ("c_x0", C.c_float), ("c_x", C.c_float),
                ("c_eps", C.c_float), ("sigma", C.c_float), ("sqrt_recip", C.c_float),
                ("sqrt_recipm1", C.c_float)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes): every symbol include/prg.h declares
PROTOTYPES = {
    "prg_abi_version": (C.c_int, []),
    "prg_last_error": (C.c_char_p, []),
    "prg_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "prg_depth2pc": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P]),
    "prg_pc2depth": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "prg_project_points_zbuffer": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "prg_reproject_zbuffer": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P]),
    "prg_unproject_f64": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P]),
    "prg_depth_augment": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "prg_apply_mask": (C.c_int, [_P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "prg_occlusion_filter": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "prg_overlap_counts": (C.c_int, [_P, _P, _I, _L, C.c_double, _P, _P]),
    "prg_unet_param_count": (_L, [C.POINTER(UnetConfigC)]),
    "prg_unet_create": (C.c_int, [C.POINTER(UnetConfigC), _P, _L, _I, C.POINTER(_P)]),
    "prg_unet_destroy": (C.c_int, [_P]),
    "prg_unet_reserve": (C.c_int, [_P, _I, _I]),
    "prg_unet_set_time_freqs": (C.c_int, [_P, _P, _I]),
    "prg_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "prg_maskunet_forward": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "prg_debug_conv3x3": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "prg_debug_block_pair": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "prg_debug_upsample_conv3x3": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "prg_debug_conv4x4s2": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "prg_debug_conv": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "prg_debug_upsample_conv": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "prg_unet_set_taps": (C.c_int, [_P, _I]),
    "prg_unet_get_tap": (C.c_int, [_P, C.c_char_p, _P, _L, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _P]),
    "prg_sampler_create": (C.c_int, [_P, C.POINTER(StepC), _I, _I, _I, C.POINTER(_P)]),
    "prg_sampler_destroy": (C.c_int, [_P]),
    "prg_sampler_set_graph": (C.c_int, [_P, _I]),
    "prg_sampler_run": (C.c_int, [_P, _P, _P, _P, _L, _P, _P, _P]),
    "prg_sampler_set_profile": (C.c_int, [_P, _I]),
    "prg_sampler_get_profile": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]),
    "prg_sampler_get_profile_bytes": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "prg_sampler_get_profile_executed": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "prg_sampler_get_profile_step": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_L)]),
    "prg_debug_sampler_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    "prg_sampler_get_profile_shapes": (C.c_int, [_P, _P, _I, C.POINTER(C.c_int32)]),
    "prg_host_crop_aabb": (C.c_int, [_P, _L, _P, _P, _P, C.POINTER(_L)]),
    "prg_host_voxel_down_sample": (C.c_int, [_P, _L, C.c_double, _P, C.POINTER(_L)]),
    "prg_host_write_ply": (C.c_int, [C.c_char_p, _P, _L]),
    "prg_pool_create": (C.c_int, [_I, C.POINTER(_P)]),
    "prg_pool_destroy": (C.c_int, [_P]),
    "prg_pool_wait": (C.c_int, [_P, C.POINTER(_L)]),
    "prg_pool_submit_cloud": (C.c_int, [_P, C.c_char_p, _P, _L, _P, _P, _I, _P, _P, C.c_double, _P]),
    "prg_pool_submit_image": (C.c_int, [_P, C.c_char_p, _P, _I, _I, _I]),
    "prg_pool_submit_text": (C.c_int, [_P, C.c_char_p, _P, _I, _I]),
}

_lib = None


def load():
    """dlopen the HIP library and attach prototypes.  Raises PrgError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch's wheel carries its own libamdhip64: import it FIRST so this library binds to the HIP runtime torch already
    # initialised (loading /opt/rocm's copy first leaves the process with two runtimes, one of which sees no device)
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise PrgError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C pointreggpt_amd/csrc` (there is no CPU fallback)")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:  # missing ROCm runtime etc.
        raise PrgError(f"cannot load {LIB_PATH}: {e}") from e
    # PRG_HIP_LIB_ALLOW_MISSING=1 (diagnostics only, with PRG_HIP_LIB=<an older build>: same-box A/B against a previous round's
    # library, tools/gpu_r5_call2.sh): entry points that library predates are skipped; calling one fails at the call site
    lenient = os.environ.get("PRG_HIP_LIB_ALLOW_MISSING", "0") not in ("", "0") and bool(os.environ.get("PRG_HIP_LIB"))
    for name, (res, args) in PROTOTYPES.items():
        if lenient and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.prg_abi_version() != 1:
        raise PrgError(f"ABI version mismatch: library {lib.prg_abi_version()}, binding 1")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().prg_last_error()
        raise PrgError(f"{what or 'libprg_hip'} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise PrgError("no HIP device visible: pointreggpt_amd has no CPU path")

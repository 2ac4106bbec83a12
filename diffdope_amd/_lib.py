"""ctypes loader of libddx.so (the C-ABI of include/ddx.h).

The product path has no CPU fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDX_LIB: a VARIANT build of the same sources (tools/build_variant.py: timing / leave-out experiments).  Never a fallback: a path that
# does not exist is an error like a missing libddx.so
LIB_PATH = os.environ.get("DDX_LIB") or os.path.join(_HERE, "libddx.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ddx.h")

_lib = None

_I, _LL, _SZ, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_void_p

COMPAT_UNCLAMPED_BARY_GRAD = 1  # ddx.h DDX_COMPAT_UNCLAMPED_BARY_GRAD (deviation D2)


class EngineDesc(ctypes.Structure):
    _fields_ = [
        ("B", ctypes.c_int32), ("B_global", ctypes.c_int32), ("V", ctypes.c_int32), ("T", ctypes.c_int32),
        ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Th", ctypes.c_int32), ("Tw", ctypes.c_int32),
        ("use_rgb", ctypes.c_int32), ("use_depth", ctypes.c_int32), ("use_mask", ctypes.c_int32),
        ("w_rgb", ctypes.c_float), ("w_depth", ctypes.c_float), ("w_mask", ctypes.c_float),
        ("optimizer", ctypes.c_int32),
        ("adam_beta1", ctypes.c_float), ("adam_beta2", ctypes.c_float), ("adam_eps", ctypes.c_float),
        ("max_iters", ctypes.c_int32), ("use_edge", ctypes.c_int32), ("w_edge", ctypes.c_float),
        ("shade_slices", ctypes.c_int32), ("edge_slices", ctypes.c_int32), ("no_backface_cull", ctypes.c_int32),
        ("compat", ctypes.c_int32), ("separate_big_pass", ctypes.c_int32), ("single_stream", ctypes.c_int32),
    ]


class EngineBuffers(ctypes.Structure):
    _fields_ = [
        ("pos", _P), ("tri", _P), ("opp", _P), ("uv", _P), ("tex", _P), ("vtx_color", _P), ("proj", _P),
        ("gt_rgb", _P), ("gt_depth", _P), ("gt_seg", _P), ("lr_mult", _P), ("lr_sched", _P),
        ("params", _P), ("loss_log", _P), ("mtx_log", _P), ("scratch", _P), ("scratch_bytes", _SZ),
    ]


_SIGNATURES = {
    "ddx_version": (_I, []),
    "ddx_last_error": (ctypes.c_char_p, []),
    "ddx_set_compat": (_I, [_I]),
    "ddx_xfm_fwd": (_I, [_P, _LL, _P, _I, _I, _I, _P, _I, _P]),
    "ddx_xfm_bwd_points": (_I, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "ddx_xfm_bwd_mtx": (_I, [_P, _LL, _I, _I, _I, _P, _P, _I, _P]),
    "ddx_xfm_bwd_full": (_I, [_P, _LL, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "ddx_pose_matrix_fwd": (_I, [_P, _P, _I, _P, _P]),
    "ddx_pose_matrix_bwd": (_I, [_P, _P, _I, _P, _P, _P]),
    "ddx_pose_pack_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "ddx_pose_pack_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "ddx_rasterize_scratch_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "ddx_rasterize_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "ddx_rasterize_fwd_rows": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P, _I, _P]),
    "ddx_rasterize_fwd_rows_clean": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P, _I, _I, _P]),
    "ddx_rasterize_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ddx_interpolate_fwd": (_I, [_P, _LL, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "ddx_interpolate_bwd": (_I, [_P, _LL, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ddx_texture_linear_fwd": (_I, [_P, _LL, _I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "ddx_texture_linear_bwd": (_I, [_P, _LL, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "ddx_topology_build": (_I, [_P, _I, _P]),
    "ddx_antialias_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "ddx_antialias_bwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ddx_gbuffer_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ddx_gbuffer_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "ddx_gbuffer_fwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "ddx_gbuffer_bwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "ddx_silhouette_fwd_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "ddx_silhouette_bwd_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ddx_silhouette_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "ddx_silhouette_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "ddx_masked_l1_fwd": (_I, [_P, _P, _P, _I, _I, _LL, _P, _P, _P]),
    "ddx_masked_l1_bwd": (_I, [_P, _P, _P, _I, _P, _I, _LL, _P, _P]),
    "ddx_masked_l1_bc3_fwd": (_I, [_P, _P, _P, _I, _LL, _P, _P, _P]),
    "ddx_masked_l1_bc3_bwd": (_I, [_P, _P, _P, _P, _I, _LL, _P, _P]),
    "ddx_masked_l1_fwd_sum": (_I, [_P, _P, _P, _I, _I, _I, _LL, _P, _P, _P, _P, _P]),
    "ddx_masked_l1_bwd_sum": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _LL, _P, _P]),
    "ddx_gbuffer_fwd_rows_c": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "ddx_silhouette_fwd_rows_c": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "ddx_silhouette_bwd_rows_c": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "ddx_engine_scratch_bytes": (_SZ, [ctypes.POINTER(EngineDesc)]),
    "ddx_engine_create": (_I, [ctypes.POINTER(EngineDesc), ctypes.POINTER(EngineBuffers), ctypes.POINTER(_P)]),
    "ddx_engine_run": (_I, [_P, _I, _I, _I, _P]),
    "ddx_engine_run_select": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "ddx_engine_run_check": (_I, [_P, _P]),
    "ddx_engine_group_run_check": (_I, [_P, _P]),
    "ddx_engine_eval": (_I, [_P, _I, _P, _P, _P]),
    "ddx_select_best": (_I, [_P, _I, _I, _P, _I, _P, _P]),
    "ddx_render_loss_fwd": (_I, [_P, _I, _P, _P]),
    "ddx_render_loss_bwd": (_I, [_P, _I, _P, _P]),
    "ddx_sgd_step": (_I, [_P, _P, ctypes.c_float, _I, _P]),
    "ddx_adam_step": (_I, [_P, _P, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "ddx_engine_status_ptr": (_P, [_P]),
    "ddx_engine_cull_sign": (_I, [_P]),
    "ddx_engine_two_chains": (_I, [_P]),
    "ddx_engine_new_observation": (_I, [_P]),
    "ddx_engine_profile": (_I, [_P, _I, _I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), _I, _P]),
    "ddx_engine_destroy": (None, [_P]),
    "ddx_engine_group_create": (_I, [ctypes.POINTER(_P), _I, ctypes.POINTER(_P)]),
    "ddx_engine_group_run": (_I, [_P, _I, _I, _P]),
    "ddx_engine_group_invalidate": (_I, [_P]),
    "ddx_engine_group_destroy": (None, [_P]),
    "ddx_engine_trace_read": (_I, [_P, _P, _I]),
}


def declared_symbols():
    """Every function include/ddx.h declares (used by the CPU test that checks the exports)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddx_[a-z_0-9]+)\s*\(", txt)))


def load():
    """dlopen libddx.so and bind every symbol of ddx.h.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python __graft_entry__.py`); diffdope_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libddx.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().ddx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")


_raw_stream = None


def stream_ptr():
    """The current HIP stream of the current device as a void pointer.  (torch.cuda.current_stream().cuda_stream builds a Stream
    object through several Python layers -- 14 us per call, seven calls per iteration of the op-by-op path, which is bound by its
    host side; the raw getter is what the object wraps.)"""
    global _raw_stream
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())

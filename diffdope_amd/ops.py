"""diffdope.ops for MI355X: xfm_points / xfm_vectors with the reference's names, argument meaning,
autograd behaviour and error behaviour (diffdope/ops.py:104-175), backed by hand-written gfx950
kernels (csrc/xfm.hip) through the C ABI of include/ddx.h instead of the JIT-compiled CUDA plugin
(ops.py:23-97).

There is no CPU fallback: non-`use_python` calls need ROCm tensors and libddx.so.
"""
import torch

from . import _lib

# 0 = MFMA (v_mfma_f32_4x4x1_16b_f32), 1 = plain VALU fma; kept switchable for A/B measurements
XFM_VARIANT = 0


def _mtx_variant():
    """Variant of the d_matrix kernels: + 2 = deterministic (one workgroup per hypothesis in a fixed order instead of fp32
    atomicAdd across workgroups, ddx.h) when torch.use_deterministic_algorithms(True) or DDX_DETERMINISTIC=1 asks for it."""
    import os

    det = torch.are_deterministic_algorithms_enabled() or os.environ.get("DDX_DETERMINISTIC", "0") not in ("", "0")
    return XFM_VARIANT | (2 if det else 0)


def _check_tensor(t, name, rank, channels):
    # mirrors CHECK_TENSOR in torch_bindings.cpp:30-31,144-145 (device, dtype, rank, channels)
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/ROCm tensor (diffdope_amd has no CPU path; use use_python=True)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != rank or t.shape[-1] != channels:
        raise RuntimeError(f"{name} must have rank {rank} and {channels} channels, got {tuple(t.shape)}")


class _Plugin:
    """Stands in for the module `renderutils_plugin` (torch_bindings.cpp:279-284): same four
    functions, same argument order, outputs allocated inside, asynchronous on the current stream."""

    def __init__(self):
        self.lib = _lib.load()

    @staticmethod
    def _prep(points, matrix):
        _check_tensor(points, "points", 3, 3)
        _check_tensor(matrix, "matrix", 3, 4)
        if matrix.shape[1] != 4:
            raise RuntimeError(f"matrix must be [B,4,4], got {tuple(matrix.shape)}")
        B = max(matrix.shape[0], points.shape[0])
        if matrix.shape[0] != B:
            raise RuntimeError("matrix must carry the full batch")  # torch_bindings.cpp:164 quirk made explicit
        if points.shape[0] not in (1, B):
            raise RuntimeError(f"points batch must be 1 or {B}, got {points.shape[0]}")
        points = points.contiguous()
        matrix = matrix.contiguous()
        pbs = 0 if (points.shape[0] == 1 and B > 1) else points.shape[1] * 3
        return points, matrix, B, points.shape[1], pbs

    def xfm_fwd(self, points, matrix, isPoints, fp16=False):
        if fp16:
            raise RuntimeError("fp16/bf16 xfm is not supported (the reference never enables it, ops.py:109)")
        points, matrix, B, N, pbs = self._prep(points, matrix)
        out = torch.empty((B, N, 4 if isPoints else 3), dtype=torch.float32, device=matrix.device)
        _lib.check(self.lib.ddx_xfm_fwd(_lib.ptr(points), pbs, _lib.ptr(matrix), B, N, int(isPoints), _lib.ptr(out),
                                         XFM_VARIANT, _lib.stream_ptr()), "ddx_xfm_fwd")
        return out

    def xfm_bwd(self, points, matrix, grad, isPoints):
        points, matrix, B, N, pbs = self._prep(points, matrix)
        grad = grad.contiguous()
        dp = torch.empty((B, N, 3), dtype=torch.float32, device=matrix.device)
        _lib.check(self.lib.ddx_xfm_bwd_points(_lib.ptr(matrix), B, N, int(isPoints), _lib.ptr(grad), _lib.ptr(dp),
                                                XFM_VARIANT, _lib.stream_ptr()), "ddx_xfm_bwd_points")
        return dp

    def xfm_bwd_mtx(self, points, matrix, grad, isPoints):
        points, matrix, B, N, pbs = self._prep(points, matrix)
        grad = grad.contiguous()
        dm = torch.empty((B, 4, 4), dtype=torch.float32, device=matrix.device)
        _lib.check(self.lib.ddx_xfm_bwd_mtx(_lib.ptr(points), pbs, B, N, int(isPoints), _lib.ptr(grad), _lib.ptr(dm),
                                             _mtx_variant(), _lib.stream_ptr()), "ddx_xfm_bwd_mtx")
        return dm

    def xfm_bwd_full(self, points, matrix, grad, isPoints):
        points, matrix, B, N, pbs = self._prep(points, matrix)
        grad = grad.contiguous()
        dp = torch.empty((B, N, 3), dtype=torch.float32, device=matrix.device)
        dm = torch.empty((B, 4, 4), dtype=torch.float32, device=matrix.device)
        variant = _mtx_variant()
        _lib.check(self.lib.ddx_xfm_bwd_full(_lib.ptr(points), pbs, _lib.ptr(matrix), B, N, int(isPoints),
                                              _lib.ptr(grad), _lib.ptr(dp), _lib.ptr(dm), variant,
                                              _lib.stream_ptr()), "ddx_xfm_bwd_full")
        return dp, dm


_cached_plugin = None


def _get_plugin():
    """Return the (cached) native plugin -- ops.py:23-97 without the JIT build."""
    global _cached_plugin
    if _cached_plugin is None:
        _cached_plugin = _Plugin()
    return _cached_plugin


class _xfm_func(torch.autograd.Function):
    # same dispatch as ops.py:104-125
    @staticmethod
    def forward(ctx, points, matrix, isPoints):
        ctx.save_for_backward(points, matrix)
        ctx.isPoints = isPoints
        return _get_plugin().xfm_fwd(points, matrix, isPoints, False)

    @staticmethod
    def backward(ctx, dout):
        points, matrix = ctx.saved_tensors
        matrix_grad = None
        points_grad = None
        need_p, need_m = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_m and need_p:
            points_grad, matrix_grad = _get_plugin().xfm_bwd_full(points, matrix, dout, ctx.isPoints)
        elif need_m and not need_p:
            matrix_grad = _get_plugin().xfm_bwd_mtx(points, matrix, dout, ctx.isPoints)
        else:
            points_grad = _get_plugin().xfm_bwd(points, matrix, dout, ctx.isPoints)
        if points_grad is not None and points.shape[0] == 1 and points_grad.shape[0] != 1:
            # the reference returns a dense [B,N,3] here, which autograd rejects for broadcast points
            # (never exercised upstream); reduce it so the op is usable.
            points_grad = points_grad.sum(dim=0, keepdim=True)
        return points_grad, matrix_grad, None


def xfm_points(points, matrix, use_python=False):
    """Transform points.
    Args:
        points: [minibatch_size, num_vertices, 3] or [1, num_vertices, 3]
        matrix: [minibatch_size, 4, 4]
        use_python: use torch.matmul (the reference's own validation path, ops.py:137-141)
    Returns:
        homogeneous 4D points [minibatch_size, num_vertices, 4].
    """
    if use_python:
        out = torch.matmul(
            torch.nn.functional.pad(points, pad=(0, 1), mode="constant", value=1.0),
            torch.transpose(matrix, 1, 2),
        )
    else:
        out = _xfm_func.apply(points, matrix, True)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of xfm_points contains inf or NaN"
    return out


def xfm_vectors(vectors, matrix, use_python=False):
    """Transform vectors with the upper-left 3x3 of `matrix` (ops.py:152-175).
    Returns [minibatch_size, num_vertices, 3]."""
    if use_python:
        out = torch.matmul(
            torch.nn.functional.pad(vectors, pad=(0, 1), mode="constant", value=0.0),
            torch.transpose(matrix, 1, 2),
        )[..., 0:3].contiguous()
    else:
        out = _xfm_func.apply(vectors, matrix, False)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of xfm_vectors contains inf or NaN"
    return out

"""Pose helpers with the reference's names (diffdope/diffdope.py:46-89)."""
import torch


class _pose_matrix_func(torch.autograd.Function):
    """ddx_pose_matrix_fwd / _bwd: one kernel each way for ROCm float32 inputs."""

    @staticmethod
    def forward(ctx, q, p):
        from . import _lib

        q, p = q.contiguous(), p.contiguous()
        B = q.shape[0]
        out = torch.empty((B, 4, 4), dtype=torch.float32, device=q.device)
        _lib.check(_lib.load().ddx_pose_matrix_fwd(_lib.ptr(q), _lib.ptr(p), B, _lib.ptr(out), _lib.stream_ptr()), "ddx_pose_matrix_fwd")
        ctx.save_for_backward(q)
        return out

    @staticmethod
    def backward(ctx, dm):
        from . import _lib

        (q,) = ctx.saved_tensors
        B = q.shape[0]
        dm = dm.contiguous()
        dq = torch.empty((B, 4), dtype=torch.float32, device=q.device)
        dp = torch.empty((B, 3), dtype=torch.float32, device=q.device)
        _lib.check(_lib.load().ddx_pose_matrix_bwd(_lib.ptr(q), _lib.ptr(dm), B, _lib.ptr(dq), _lib.ptr(dp), _lib.stream_ptr()), "ddx_pose_matrix_bwd")
        return dq, dp


def matrix_batch_44_from_position_quat(q, p):
    """(batch,4) xyzw quaternion + (batch,3) translation -> (batch,4,4), differentiable.  Same row formulas as
    diffdope.py:57-80.  ROCm float32 tensors take one kernel each way (the reference's ~30 small ops, ~60 more in the
    backward, are most of the launches of an op-by-op iteration); anything else the same formulas as torch expressions."""
    if q.is_cuda and p.is_cuda and q.dtype == torch.float32 and p.dtype == torch.float32 and q.dim() == 2 and p.dim() == 2:
        return _pose_matrix_func.apply(q, p)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    rows = [
        1.0 - 2.0 * y**2 - 2.0 * z**2, 2.0 * x * y - 2.0 * z * w, 2.0 * x * z + 2.0 * y * w, p[:, 0],
        2.0 * x * y + 2.0 * z * w, 1.0 - 2.0 * x**2 - 2.0 * z**2, 2.0 * y * z - 2.0 * x * w, p[:, 1],
        2.0 * x * z - 2.0 * y * w, 2.0 * y * z + 2.0 * x * w, 1.0 - 2.0 * x**2 - 2.0 * y**2, p[:, 2],
        zero, zero, zero, one,
    ]
    return torch.stack(rows, dim=1).reshape(-1, 4, 4)

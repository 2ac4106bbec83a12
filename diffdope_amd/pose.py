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


class _pose_pack_func(torch.autograd.Function):
    """ddx_pose_pack_fwd / _bwd: (qx, qy, qz, qw, x, y, z), each [B] -> (quat [B,4] normalised, trans [B,3])."""

    @staticmethod
    def forward(ctx, qx, qy, qz, qw, x, y, z):
        from . import _lib

        prm = [t.contiguous() for t in (qx, qy, qz, qw, x, y, z)]
        B = prm[0].shape[0]
        quat = torch.empty((B, 4), dtype=torch.float32, device=prm[0].device)
        trans = torch.empty((B, 3), dtype=torch.float32, device=prm[0].device)
        _lib.check(_lib.load().ddx_pose_pack_fwd(*[_lib.ptr(t) for t in prm], B, _lib.ptr(quat), _lib.ptr(trans), _lib.stream_ptr()), "ddx_pose_pack_fwd")
        ctx.save_for_backward(*prm[:4])
        ctx.set_materialize_grads(False)
        return quat, trans

    @staticmethod
    def backward(ctx, dquat, dtrans):
        from . import _lib

        q = ctx.saved_tensors
        B = q[0].shape[0]
        if dquat is None and dtrans is None:
            return (None,) * 7
        dquat = None if dquat is None else dquat.contiguous()
        dtrans = None if dtrans is None else dtrans.contiguous()
        d = torch.empty((7, B), dtype=torch.float32, device=q[0].device)
        _lib.check(_lib.load().ddx_pose_pack_bwd(*[_lib.ptr(t) for t in q], _lib.ptr(dquat) if dquat is not None else None,
                                                 _lib.ptr(dtrans) if dtrans is not None else None, B, _lib.ptr(d), _lib.stream_ptr()), "ddx_pose_pack_bwd")
        return tuple(d.unbind(0))


def quat_trans_from_parameters(qx, qy, qz, qw, x, y, z):
    """Object3D.forward's pose head (diffdope.py:1085-1098): quat = stack(qx, qy, qz, qw) / its norm [B,4], trans = stack(x, y, z)
    [B,3].  ROCm float32 [B] tensors take one kernel each way; anything else the torch expressions."""
    prm = (qx, qy, qz, qw, x, y, z)
    if all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 1 and t.shape == qx.shape for t in prm):
        return _pose_pack_func.apply(*prm)
    raw_q = torch.stack((qx, qy, qz, qw), dim=1)
    return raw_q / raw_q.norm(dim=1, keepdim=True), torch.stack((x, y, z), dim=1)


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

"""Pose helpers with the reference's names (diffdope/diffdope.py:46-89)."""
import torch


def matrix_batch_44_from_position_quat(q, p):
    """(batch,4) xyzw quaternion + (batch,3) translation -> (batch,4,4), autograd-traceable.
    Same row formulas as diffdope.py:57-80; built with two stacks instead of ~30 small ops and without
    the per-call host->device constant of diffdope.py:85."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    rows = [
        1.0 - 2.0 * y**2 - 2.0 * z**2, 2.0 * x * y - 2.0 * z * w, 2.0 * x * z + 2.0 * y * w, p[:, 0],
        2.0 * x * y + 2.0 * z * w, 1.0 - 2.0 * x**2 - 2.0 * z**2, 2.0 * y * z - 2.0 * x * w, p[:, 1],
        2.0 * x * z - 2.0 * y * w, 2.0 * y * z + 2.0 * x * w, 1.0 - 2.0 * x**2 - 2.0 * y**2, p[:, 2],
        zero, zero, zero, one,
    ]
    return torch.stack(rows, dim=1).reshape(-1, 4, 4)

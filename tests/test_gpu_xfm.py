"""GPU parity of the xfm kernels (the reference's own native ops, c_src/mesh.cu) against the golden
vectors captured from the reference's use_python path and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dd():
    import diffdope_amd as dd

    return dd


@pytest.mark.parametrize("variant", [0, 1])
def test_xfm_against_reference_golden_vectors(dd, golden_dir, variant):
    from diffdope_amd import ops

    g = np.load(os.path.join(golden_dir, "g1_xfm.npz"))
    ops.XFM_VARIANT = variant
    try:
        for k in range(int(g["n_cases"])):
            pre = f"c{k}_"
            isp = bool(g[pre + "is_points"])
            p = torch.tensor(g[pre + "points"], device="cuda", requires_grad=True)
            m = torch.tensor(g[pre + "matrix"], device="cuda", requires_grad=True)
            fn = dd.xfm_points if isp else dd.xfm_vectors
            out = fn(p, m)
            np.testing.assert_allclose(out.detach().cpu().numpy(), g[pre + "out"], rtol=1e-5, atol=1e-5)
            out.backward(torch.tensor(g[pre + "dout"], device="cuda"))
            np.testing.assert_allclose(p.grad.cpu().numpy(), g[pre + "dpoints"], rtol=1e-4, atol=3e-5)
            np.testing.assert_allclose(m.grad.cpu().numpy(), g[pre + "dmatrix"], rtol=1e-4, atol=1e-4)
    finally:
        ops.XFM_VARIANT = 0


def test_xfm_forward_bit_exact_vs_oracle_and_between_variants(dd):
    from diffdope_amd import ops
    from oracle import oracle as orc

    rng = np.random.RandomState(0)
    pts = rng.normal(size=(3, 1000, 3)).astype(np.float32)
    mtx = rng.normal(size=(3, 4, 4)).astype(np.float32)
    ref = orc.xfm_fwd(pts, mtx, True)
    outs = []
    for variant in (0, 1):
        ops.XFM_VARIANT = variant
        outs.append(dd.xfm_points(torch.tensor(pts, device="cuda"), torch.tensor(mtx, device="cuda")).cpu().numpy())
    ops.XFM_VARIANT = 0
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], ref)  # k-ordered fma chain == v_mfma_f32_4x4x1 accumulation


def test_xfm_backward_dispatch_matches_use_python(dd):
    torch.manual_seed(0)
    for need_p, need_m in [(True, True), (False, True), (True, False)]:
        p = torch.randn(4, 777, 3, device="cuda", requires_grad=need_p)
        m = torch.randn(4, 4, 4, device="cuda", requires_grad=need_m)
        g = torch.randn(4, 777, 4, device="cuda")
        out = dd.xfm_points(p, m)
        out.backward(g)
        p2, m2 = p.detach().clone().requires_grad_(need_p), m.detach().clone().requires_grad_(need_m)
        dd.xfm_points(p2, m2, use_python=True).backward(g)
        if need_p:
            torch.testing.assert_close(p.grad, p2.grad, rtol=1e-4, atol=1e-4)
        if need_m:
            torch.testing.assert_close(m.grad, m2.grad, rtol=1e-4, atol=2e-3)


def test_xfm_backward_deterministic_mode(dd, monkeypatch):
    """DDX_DETERMINISTIC=1 / torch.use_deterministic_algorithms(True): d_matrix from one workgroup per hypothesis in a fixed order
    (no cross-workgroup atomicAdd) -- repeated runs are bit-identical at the hot path's size (N = 640*480), and equal to the
    default (atomic) kernels to rounding."""
    torch.manual_seed(2)
    p = torch.randn(3, 640 * 480, 3, device="cuda")
    m = torch.randn(3, 4, 4, device="cuda", requires_grad=True)
    g = torch.randn(3, 640 * 480, 4, device="cuda")

    def grad_m():
        m.grad = None
        dd.xfm_points(p, m).backward(g)
        return m.grad.clone()

    base = grad_m()
    monkeypatch.setenv("DDX_DETERMINISTIC", "1")
    runs = [grad_m() for _ in range(4)]
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    torch.testing.assert_close(runs[0], base, rtol=1e-4, atol=1e-2)
    monkeypatch.delenv("DDX_DETERMINISTIC")
    torch.use_deterministic_algorithms(True)
    try:
        assert torch.equal(grad_m(), runs[0])
    finally:
        torch.use_deterministic_algorithms(False)


def test_xfm_large_n_linearity_property(dd):
    # size-independent property at the hot path's full size (N = 640*480): linearity in the matrix
    torch.manual_seed(1)
    p = torch.randn(2, 640 * 480, 3, device="cuda")
    m1, m2 = torch.randn(2, 4, 4, device="cuda"), torch.randn(2, 4, 4, device="cuda")
    a = dd.xfm_points(p, m1) + dd.xfm_points(p, m2)
    b = dd.xfm_points(p, m1 + m2)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


def test_xfm_error_behaviour(dd):
    with pytest.raises(RuntimeError):
        dd.xfm_points(torch.randn(1, 5, 3), torch.randn(1, 4, 4))  # CPU tensors: no CPU path
    with pytest.raises(RuntimeError):
        dd.xfm_points(torch.randn(1, 5, 2, device="cuda"), torch.randn(1, 4, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        dd.xfm_points(torch.randn(1, 5, 3, device="cuda").double(), torch.randn(1, 4, 4, device="cuda"))
    with torch.autograd.detect_anomaly(check_nan=False):
        with pytest.raises(AssertionError):
            dd.xfm_points(torch.full((1, 2, 3), float("nan"), device="cuda"), torch.eye(4, device="cuda")[None])

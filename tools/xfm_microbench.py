"""Microbenchmark of the op-level xfm kernels (MFMA variant 0 vs VALU variant 1): GB/s of algorithmic traffic."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6
for B, N in [(64, 10449), (128, 25921), (64, 307200)]:
    pts = torch.randn(B, N, 3, device='cuda'); M = torch.randn(B, 4, 4, device='cuda'); out = torch.empty(B, N, 4, device='cuda'); g = torch.randn(B, N, 4, device='cuda')
    dp = torch.empty(B, N, 3, device='cuda'); dm = torch.empty(B, 4, 4, device='cuda')
    for var in (0, 1):
        t1 = bench(lambda: lib.ddx_xfm_fwd(p(pts), N * 3, p(M), B, N, 1, p(out), var, s))
        t2 = bench(lambda: lib.ddx_xfm_bwd_mtx(p(pts), N * 3, B, N, 1, p(g), p(dm), var, s))
        t3 = bench(lambda: lib.ddx_xfm_bwd_full(p(pts), N * 3, p(M), B, N, 1, p(g), p(dp), p(dm), var, s))
        print(f'B={B} N={N} variant={var}: fwd {t1:.1f}us {B*N*28/t1/1e6:.2f} TB/s | bwd_mtx {t2:.1f}us {B*N*28/t2/1e6:.2f} TB/s | bwd_full {t3:.1f}us {B*N*40/t3/1e6:.2f} TB/s')
    # CPU baseline of SURVEY 8(d): the reference's use_python path (torch.matmul on the padded points, ops.py:137-141) on the host
    # cores of this box, forward + backward through autograd
    if N <= 30000:
        torch.set_num_threads(os.cpu_count())
        pc, Mc = pts.cpu().requires_grad_(True), M.cpu().requires_grad_(True)
        gc = g.cpu()
        def cpu_once():
            o = torch.matmul(torch.nn.functional.pad(pc, pad=(0, 1), mode="constant", value=1.0), torch.transpose(Mc, 1, 2))
            o.backward(gc)
            pc.grad = None; Mc.grad = None
        for _ in range(3): cpu_once()
        t = time.time(); n = 20
        for _ in range(n): cpu_once()
        tc = (time.time() - t) / n * 1e6
        print(f'B={B} N={N} CPU torch.matmul fwd+bwd ({os.cpu_count()} threads): {tc:.0f}us = {B*N*(28+40)/tc/1e3:.2f} GB/s of the same algorithmic traffic')

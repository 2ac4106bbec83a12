import ctypes, torch, time, sys
lib = ctypes.CDLL('/root/repo/diffdope_amd/libddx.so')
lib.ddx_last_error.restype = ctypes.c_char_p
P = ctypes.c_void_p
def p(t): return P(t.data_ptr())
s = P(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
ok = True
for B, N in [(1,1),(3,65),(2,257),(64,10240),(4,307200)]:
  for isp in (1,0):
    for var in (0,1):
        pts = torch.randn(B,N,3,device='cuda'); M = torch.randn(B,4,4,device='cuda')
        R = 4 if isp else 3
        out = torch.empty(B,N,R,device='cuda')
        e = lib.ddx_xfm_fwd(p(pts), ctypes.c_longlong(N*3), p(M), B, N, isp, p(out), var, s)
        assert e==0, lib.ddx_last_error()
        ref = torch.matmul(torch.nn.functional.pad(pts,(0,1),value=float(isp)).double(), M.double().transpose(1,2))[...,:R]
        err = (out.double()-ref).abs().max().item()
        g = torch.randn(B,N,R,device='cuda')
        dp = torch.empty(B,N,3,device='cuda'); dm = torch.empty(B,4,4,device='cuda'); dp2 = torch.empty(B,N,3,device='cuda'); dm2=torch.empty(B,4,4,device='cuda')
        assert lib.ddx_xfm_bwd_points(p(M),B,N,isp,p(g),p(dp),var,s)==0
        assert lib.ddx_xfm_bwd_mtx(p(pts),ctypes.c_longlong(N*3),B,N,isp,p(g),p(dm),var,s)==0
        assert lib.ddx_xfm_bwd_full(p(pts),ctypes.c_longlong(N*3),p(M),B,N,isp,p(g),p(dp2),p(dm2),var,s)==0
        gp = torch.nn.functional.pad(g,(0,4-R)).double()
        rdp = torch.matmul(gp, M.double())[...,:3]
        rdm = torch.matmul(gp.transpose(1,2), torch.nn.functional.pad(pts,(0,1),value=float(isp)).double())
        if not isp: rdm[:,:,3]=0; rdm[:,3,:]=0
        e2 = (dp.double()-rdp).abs().max().item(); e3=(dm.double()-rdm).abs().max().item()/max(1,rdm.abs().max().item())
        e4 = (dp2.double()-rdp).abs().max().item(); e5=(dm2.double()-rdm).abs().max().item()/max(1,rdm.abs().max().item())
        good = max(err,e2,e4)<1e-5 and max(e3,e5)<1e-5
        ok &= good
        print(B,N,isp,var,'fwd',err,'dp',e2,e4,'dm',e3,e5,'OK' if good else 'FAIL')
# bit-exactness of MFMA vs VALU chain
pts = torch.randn(8,5000,3,device='cuda'); M=torch.randn(8,4,4,device='cuda')
o0=torch.empty(8,5000,4,device='cuda'); o1=torch.empty_like(o0)
lib.ddx_xfm_fwd(p(pts),ctypes.c_longlong(15000),p(M),8,5000,1,p(o0),0,s); lib.ddx_xfm_fwd(p(pts),ctypes.c_longlong(15000),p(M),8,5000,1,p(o1),1,s)
print('mfma==valu bitwise:', torch.equal(o0,o1))
# timing
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e6
for B,N in [(64,10240),(128,25600),(64,307200)]:
    pts = torch.randn(B,N,3,device='cuda'); M=torch.randn(B,4,4,device='cuda'); out=torch.empty(B,N,4,device='cuda'); g=torch.randn(B,N,4,device='cuda')
    dp=torch.empty(B,N,3,device='cuda'); dm=torch.empty(B,4,4,device='cuda')
    for var in (0,1):
        t1=bench(lambda: lib.ddx_xfm_fwd(p(pts),ctypes.c_longlong(N*3),p(M),B,N,1,p(out),var,s))
        t2=bench(lambda: lib.ddx_xfm_bwd_mtx(p(pts),ctypes.c_longlong(N*3),B,N,1,p(g),p(dm),var,s))
        t3=bench(lambda: lib.ddx_xfm_bwd_full(p(pts),ctypes.c_longlong(N*3),p(M),B,N,1,p(g),p(dp),p(dm),var,s))
        by1=B*N*28; by2=B*N*28; by3=B*N*40
        print(f'B={B} N={N} var={var} fwd {t1:.1f}us {by1/t1/1e6:.2f}TB/s | bwd_mtx {t2:.1f}us {by2/t2/1e6:.2f}TB/s | bwd_full {t3:.1f}us {by3/t3/1e6:.2f}TB/s')
print('ALL OK' if ok else 'SOME FAIL')

"""Op-by-op (materialising) path: render_texture_batch + torch L1 losses + autograd backward + SGD step on a BASELINE
workload -- the compatibility path user loss functions take (every image materialised in HBM), next to the fused engine."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
from diffdope_amd.render import RasterizeContext, render_texture_batch

cfg = next((a for a in sys.argv[1:] if not a.startswith('--')), 'cfg2')
dev = torch.device('cuda:0')
w = wl.build(cfg, dev)
B, H, W = w['B'], w['H'], w['W']
params = w['params0'].clone().requires_grad_(True)
ctx = RasterizeContext()
ex = lambda t: t[None].expand(B, *t.shape)
kw = dict(uv=ex(w['uv']), uv_idx=ex(w['tri']), tex=ex(w['tex'])) if w['uv'] is not None else dict(vtx_color=ex(w['vtx_color']))
gt = {k: v[None] for k, v in w['gt'].items()}
lr_mult = w['lr_mult']
wt = w['weights']

from diffdope_amd.render import masked_l1_mean
FUSED_LOSS = '--torch-losses' not in sys.argv  # the built-in loss functions use render.masked_l1_mean; --torch-losses: the plain expressions

# (what the loss terms read: without a colour term the colour image is not rendered -- as DiffDope._loop_outputs decides for the built-in losses)
OUTPUTS = None if wt.get('rgb') is not None or '--all-outputs' in sys.argv else tuple(k for k in ('depth', 'mask') if wt.get(k) is not None)

def step():
    q = params[:4].T / torch.norm(params[:4].T, dim=1, keepdim=True)
    mtx = dd.matrix_batch_44_from_position_quat(q=q, p=params[4:].T)
    r = render_texture_batch(ctx, ex(w['proj']), mtx, ex(w['pos']), ex(w['tri']), [H, W], outputs=OUTPUTS, **kw)
    loss = 0
    if FUSED_LOSS:
        if wt.get('rgb') is not None:
            loss = loss + (masked_l1_mean(r['rgb'], gt['rgb'], gt['segmentation']) * lr_mult).mean() * wt['rgb']
        if wt.get('depth') is not None:
            loss = loss + (masked_l1_mean(r['depth'], gt['depth'], gt['segmentation'], mask_channel0=True) * lr_mult).mean() * wt['depth']
        if wt.get('mask') is not None:
            loss = loss + (masked_l1_mean(r['mask'], gt['segmentation']) * lr_mult).mean() * wt['mask']
    else:
        if wt.get('rgb') is not None:
            loss = loss + (torch.abs((r['rgb'] - gt['rgb']) * gt['segmentation']).mean((1, 2, 3)) * lr_mult).mean() * wt['rgb']
        if wt.get('depth') is not None:
            loss = loss + (torch.abs((r['depth'] - gt['depth']) * gt['segmentation'][..., 0]).mean((1, 2)) * lr_mult).mean() * wt['depth']
        if wt.get('mask') is not None:
            loss = loss + (torch.abs(r['mask'] - gt['segmentation']).mean((1, 2, 3)) * lr_mult).mean() * wt['mask']
    g, = torch.autograd.grad(loss, params)
    with torch.no_grad():
        params.sub_(1e-3 * g)

for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'{cfg} ({"fused masked-L1 losses" if FUSED_LOSS else "torch loss expressions"}): op-by-op path {dt*1e3:.2f} ms/iteration = {1/dt:.0f} it/s  (peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB)')

if '--graph' in sys.argv:
    # the same iteration captured ONCE into a hipGraph (torch.cuda.graph: the library's launches go to torch's current stream, so
    # they are captured like torch's own) and replayed: what the ~60 framework launches per iteration cost on the host
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f'{cfg}: the same iteration replayed from one captured graph: {dt*1e3:.2f} ms/iteration = {1/dt:.0f} it/s')

if '--api' in sys.argv:
    # the same workload through DiffDope.run_optimization(fused=False): Object3D / Mesh modules, the built-in loss functions
    # with their per-iteration logs, torch SGD -- what a user loss function forces (api._run_autograd)
    import numpy as np
    nb = 40
    tex_kw = dict(uv=w['uv'].cpu().numpy(), tex=w['tex'].cpu().numpy()) if w['uv'] is not None else dict(vtx_color=w['vtx_color'].cpu().numpy())
    mesh = dd.Mesh.from_arrays(w['pos'].cpu().numpy(), w['tri'].cpu().numpy(), **tex_kw)
    p0 = w['params0'][:, 0].cpu().numpy()
    obj = dd.Object3D(position=list(p0[4:]), rotation=list(p0[:4] / np.linalg.norm(p0[:4])), batchsize=B, opencv2opengl=False, scale=1, mesh=mesh)
    g = {k: v.cpu() for k, v in w['gt'].items()}
    scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=g['rgb']), tensor_depth=dd.Image(img_tensor=g['depth']),
                     tensor_segmentation=dd.Image(img_tensor=g['segmentation']))
    cam = dd.Camera(fx=1, fy=1, cx=0, cy=0, im_width=W, im_height=H)
    cam.cam_proj = w['proj'].double().cpu()
    cfg_d = dict(losses=dict(l1_rgb_with_mask=wt.get('rgb') is not None, weight_rgb=wt.get('rgb') or 1.0,
                             l1_depth_with_mask=wt.get('depth') is not None, weight_depth=wt.get('depth') or 1.0,
                             l1_mask=wt.get('mask') is not None, weight_mask=wt.get('mask') or 1.0),
                 hyperparameters=dict(nb_iterations=nb, batchsize=B, base_lr=1e-3, learning_rates_bound=[0.5, 2.0], learning_rate_base=1,
                                      lr_decay=0.1, seed=3))
    d = dd.DiffDope(cfg=cfg_d, camera=cam, object3d=obj, scene=scene)
    if '--fused' in sys.argv:  # the default path of the same object: whole calls incl. engine construction and result collection
        d.cfg.hyperparameters.nb_iterations = 100
        for _ in range(2): d.run_optimization(fused=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): d.run_optimization(fused=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'{cfg}: DiffDope.run_optimization(fused=True), 100 iterations: {dt*1e3:.2f} ms per call')
        sys.exit(0)
    nb = int(os.environ.get('DDX_API_NB', nb))
    d.cfg.hyperparameters.nb_iterations = nb
    for label, kw_ in (('eager', dict(graph=False)), ('captured iteration', dict(graph=True))):
        d.run_optimization(fused=False, **kw_)
        for rep in ('second call', 'third call'):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d.run_optimization(fused=False, **kw_)
            best = int(d.get_argmin())  # (reads the logs: the deferred host copies happen here)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (nb + 1)
            print(f'{cfg}: DiffDope.run_optimization(fused=False), {label}, {nb + 1} iterations, {rep}: {dt*1e3:.2f} ms/iteration = {1/dt:.0f} it/s (arg-min hypothesis {best})')

"""Python face of the fused refinement engine (csrc/engine.hip, include/ddx.h ddx_engine_*).

`RefineEngine` runs the body of DiffDope.run_optimization (diffdope/diffdope.py:1656-1714) for the
built-in losses entirely on the device.  All buffers are torch tensors owned here; the native side
only borrows pointers.
"""
import ctypes
import weakref

import torch

from . import _lib
from .render import build_topology

KERNEL_NAMES_MAX = 16


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, eng):
        if params.data_ptr() != eng.params.data_ptr():
            eng.params.copy_(params.detach())
        losses, grad = eng.loss_and_grad()
        ctx.save_for_backward(grad)
        return (losses * eng.lr_mult[None]).sum() / float(eng.desc.B_global)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class RefineEngine:
    """
    Args (torch tensors on one ROCm device, fp32 / int32):
        pos [V,3], tri [T,3], proj [4,4]; uv [V,2] + tex [Th,Tw,3]  or  vtx_color [V,3]
        gt: dict with 'segmentation' [H,W,3] and optionally 'rgb' [H,W,3], 'depth' [H,W]
            (bottom-up rows, as diffdope.py:1131 holds them)
        params [7,B]: qx,qy,qz,qw,x,y,z (diffdope.py:1019-1026) -- updated in place
        lr_mult [B]: per-hypothesis loss multipliers (diffdope.py:1368-1375)
        lr_sched: list/1-D tensor of optimiser learning rates, one per iteration (diffdope.py:1657-1661)
        weights: dict(rgb=, depth=, mask=, edge=) -- None/absent disables the term (cfg.losses); "edge" is this
            build's extension (Sobel-gradient L1 of the luminance, no reference counterpart: oracle orc_loss_edge)
        global_batch: batch size of the whole job when hypotheses are sharded over GPUs
        shade_slices / edge_slices: workgroups per hypothesis of the shading / edge launches (0 = from B).  A shard
            that must reproduce the unsharded run bit for bit passes the unsharded engine's `slices` (ddx.h)
        compat: None = this build's documented arithmetic; "nvdiffrast" = where a deviation from nvdiffrast's published behaviour is
            switchable, nvdiffrast's (D2: the rasterize backward differentiates the unclamped barycentrics; ddx.h DDX_COMPAT_*)
        cull_backfaces: False (default) = both faces of every triangle are drawn, as dr.rasterize does (diffdope.py:198-200).  True =
            deviation D5: the back faces of a CLOSED mesh are skipped while a hypothesis lies inside the view volume (ddx.h
            no_backface_cull = 0) -- identical visibility in exact arithmetic, ~8 % faster, but in float32 a back-facing sliver on the
            silhouette can win a pixel (1 351 of 8 000 random hypotheses differ in some bit: profiles/r5c_cull_sweep.json).
        separate_big_pass: True = the tile pass for large / near-clipped triangles always as its own launch (ddx.h separate_big_pass);
            default: the set-up decides from the expected triangle size (no launch where no large triangle is expected; same results).
        single_stream: True = every launch of a run on the caller's stream (ddx.h single_stream); default: a run of 16 or more
            iterations goes out as two half-batch chains, one of them on a stream the engine owns (same results, bit for bit).
    """

    def __init__(self, pos, tri, proj, resolution, gt, params, lr_mult, lr_sched, weights, uv=None, tex=None,
                 vtx_color=None, optimizer="sgd", adam=(0.9, 0.999, 1e-8), global_batch=None, log_mtx=True, shade_slices=0,
                 edge_slices=0, cull_backfaces=False, compat=None, separate_big_pass=False, single_stream=False):
        self.lib = _lib.load()
        dev = pos.device
        if dev.type != "cuda":
            raise RuntimeError("RefineEngine needs ROCm tensors (diffdope_amd has no CPU path)")
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
        self.pos, self.proj = f32(pos), f32(proj)
        self.tri = tri.to(device=dev, dtype=torch.int32).contiguous()
        self.opp = build_topology(self.tri, cached=False)  # (once per engine: no need for the op-level cache)
        self.uv, self.tex, self.vtx_color = f32(uv), f32(tex), f32(vtx_color)
        if self.tex is not None and self.tex.dim() == 4:
            self.tex = self.tex[0].contiguous()
        H, W = int(resolution[0]), int(resolution[1])
        self.H, self.W = H, W
        # (private copies: new_observation() overwrites these buffers in place, which must not reach back into the caller's images)
        own = lambda t: None if t is None else (f32(t).clone() if f32(t).data_ptr() == t.data_ptr() else f32(t))
        self.gt_seg = own(gt["segmentation"])
        self.gt_rgb = own(gt.get("rgb"))
        self.gt_depth = own(gt.get("depth"))
        assert tuple(self.gt_seg.shape) == (H, W, 3), f"segmentation must be [H,W,3], got {tuple(self.gt_seg.shape)}"
        self.params = params
        assert params.is_cuda and params.dtype == torch.float32 and params.is_contiguous() and params.shape[0] == 7
        B = params.shape[1]
        self.B = B
        self.lr_mult = own(lr_mult)
        self.lr_sched = torch.as_tensor(lr_sched, dtype=torch.float64).to(torch.float32).to(dev).contiguous()
        n_it = self.lr_sched.numel()
        self.max_iters = n_it
        self.loss_log = torch.zeros((n_it, 4, B), dtype=torch.float32, device=dev)
        self.mtx_log = torch.zeros((n_it, B, 16), dtype=torch.float32, device=dev) if log_mtx else None
        w = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
        d = _lib.EngineDesc()
        d.B, d.B_global = B, int(global_batch or B)
        d.V, d.T, d.H, d.W = self.pos.shape[0], self.tri.shape[0], H, W
        d.Th, d.Tw = (self.tex.shape[0], self.tex.shape[1]) if (self.tex is not None and self.vtx_color is None) else (0, 0)
        d.use_rgb, d.use_depth, d.use_mask = int(w["rgb"] is not None), int(w["depth"] is not None), int(w["mask"] is not None)
        d.w_rgb, d.w_depth, d.w_mask = float(w["rgb"] or 0), float(w["depth"] or 0), float(w["mask"] or 0)
        d.use_edge, d.w_edge = int(w["edge"] is not None), float(w["edge"] or 0)
        d.optimizer = {"sgd": 0, "adam": 1}[optimizer]
        d.adam_beta1, d.adam_beta2, d.adam_eps = adam
        d.max_iters = n_it
        d.shade_slices, d.edge_slices = int(shade_slices), int(edge_slices)
        d.no_backface_cull = int(not cull_backfaces)
        d.separate_big_pass = int(bool(separate_big_pass))
        d.single_stream = int(bool(single_stream))
        d.compat = {None: 0, "nvdiffrast": _lib.COMPAT_UNCLAMPED_BARY_GRAD}[compat]
        self.desc = d
        nbytes = self.lib.ddx_engine_scratch_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError("ddx_engine_scratch_bytes: " + self.lib.ddx_last_error().decode())
        self._scratch_raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-self._scratch_raw.data_ptr()) % 256
        self.scratch = self._scratch_raw[off:off + nbytes]
        b = _lib.EngineBuffers()
        P = lambda t: None if t is None else t.data_ptr()
        b.pos, b.tri, b.opp, b.uv, b.tex, b.vtx_color, b.proj = P(self.pos), P(self.tri), P(self.opp), P(self.uv), P(self.tex), P(self.vtx_color), P(self.proj)
        b.gt_rgb, b.gt_depth, b.gt_seg = P(self.gt_rgb), P(self.gt_depth), P(self.gt_seg)
        b.lr_mult, b.lr_sched, b.params = P(self.lr_mult), P(self.lr_sched), P(self.params)
        b.loss_log, b.mtx_log = P(self.loss_log), P(self.mtx_log)
        b.scratch, b.scratch_bytes = P(self.scratch), nbytes
        self.bufs = b
        h = ctypes.c_void_p()
        _lib.check(self.lib.ddx_engine_create(ctypes.byref(d), ctypes.byref(b), ctypes.byref(h)), "ddx_engine_create")
        self.handle = h
        self.it = 0
        self._unchecked = False  # a run has been enqueued that ddx_engine_run_check has not seen yet
        self._group = None       # weak reference to the RefineEngineGroup whose run this engine's unchecked run was part of
        self.repeated_runs = 0   # runs ddx_engine_run_check had to repeat (the in-launch tile pass timed out: ddx.h)

    def _run_check(self):
        """ddx_engine_run_check: synchronises the current stream; True when the last run was void and has been repeated.  A run
        that was part of a group's run is checked -- and, if void, repeated -- by the GROUP (ddx_engine_group_run_check: the
        members ran in shared launches, one of them alone cannot be repeated)."""
        grp = self._group() if self._group is not None else None
        if grp is not None and grp._unchecked:
            rep = grp._run_check()
            self.repeated_runs += int(rep)
            return rep
        self._group = None
        rc = self.lib.ddx_engine_run_check(self.handle, _lib.stream_ptr())
        if rc not in (0, 1):
            _lib.check(rc, "ddx_engine_run_check")
        self._unchecked = False
        self.repeated_runs += int(rc == 1)
        return rc == 1

    def run(self, n=None, use_graph=False):
        """Run n iterations (default: all remaining) asynchronously on the current stream.
        use_graph=k (or True = 1) replays a captured hipGraph of k iterations; with 4 launches per iteration plain
        stream launches measured 9 % faster than k = 1 and equal to k = 20 on MI355X, so streams are the default.
        Asynchronous with one exception: a run enqueued BEHIND a run that nothing has checked yet (no finish() / check() / losses()
        in between) first has ddx_engine_run_check look at that run -- a stream synchronise and a 4-byte copy -- because a run whose
        in-launch tile pass timed out is void and the next one would start from its parameters (ddx.h).  Loops that chain
        run(k); run(k) without reading anything in between pay that synchronise per call."""
        n = self.max_iters - self.it if n is None else n
        if self._unchecked:  # (a run behind a void one would start from void parameters: ddx.h ddx_engine_run_check)
            self._run_check()
        _lib.check(self.lib.ddx_engine_run(self.handle, self.it, n, int(use_graph), _lib.stream_ptr()), "ddx_engine_run")
        self.it += n
        self._unchecked = n > 0

    def run_select(self, out18, n=None, lo=0, use_graph=False):
        """run(n) with the arg-min selection of its last iteration folded into the run's last kernel (ddx_engine_run_select):
        `out18` -- an 18-float tensor in device memory or PINNED host memory -- receives (mean loss of the best local
        hypothesis over the enabled terms, lo + its index, its 4x4 pose), as ddx_select_best would write it."""
        n = self.max_iters - self.it if n is None else n
        assert out18.dtype == torch.float32 and out18.numel() >= 18 and out18.is_contiguous() and (out18.is_cuda or out18.is_pinned())
        if self._unchecked:
            self._run_check()
        _lib.check(self.lib.ddx_engine_run_select(self.handle, self.it, n, int(use_graph), int(lo), out18.data_ptr(), _lib.stream_ptr()),
                   "ddx_engine_run_select")
        self.it += n
        self._unchecked = True  # (cleared without a further call by a reader that finds out18[0] finite: dist.run_and_select)

    def new_observation(self, gt=None, params=None, lr_mult=None, lr_sched=None):
        """The same object in a new frame (ddx_engine_new_observation): copies the given observed images (dict with the keys
        of the constructor's `gt`), initial parameters [7,B], multipliers and / or schedule (same length) into the engine's
        buffers and restarts at iteration 0 with a fresh optimiser state.  The mesh half of the set-up (sorted copies, triangle
        and texel records, closedness analysis) is kept; mesh, texture, projection, resolution, batch size and loss set must be
        unchanged.  `params` is copied INTO the tensor the engine was built on (self.params), which keeps receiving the result."""
        def put(dst, src, what):
            if src is None:
                return
            if dst is None or tuple(dst.shape) != tuple(src.shape):
                raise ValueError(f"new_observation: {what} must keep its shape {None if dst is None else tuple(dst.shape)}")
            dst.copy_(src.to(device=dst.device, dtype=dst.dtype))
        if gt is not None:
            put(self.gt_seg, gt.get("segmentation"), "segmentation")
            if self.gt_rgb is not None:
                put(self.gt_rgb, gt.get("rgb"), "rgb")
            if self.gt_depth is not None:
                put(self.gt_depth, gt.get("depth"), "depth")
        put(self.params, params, "params")
        put(self.lr_mult, lr_mult, "lr_mult")
        if lr_sched is not None:
            put(self.lr_sched, torch.as_tensor(lr_sched, dtype=torch.float64).to(torch.float32), "lr_sched")
        _lib.check(self.lib.ddx_engine_new_observation(self.handle), "ddx_engine_new_observation")
        self.it = 0
        self._unchecked = False

    def finish(self):
        """Wait until everything run() enqueued on the current stream has executed (run() itself is asynchronous) and make sure
        it is valid (ddx_engine_run_check: a run whose in-launch tile pass timed out is repeated here, with the separate launch).
        Returns True when that happened."""
        if self._unchecked:
            return self._run_check()
        torch.cuda.current_stream().synchronize()
        return False

    def loss_and_grad(self):
        """One evaluation pass at the CURRENT contents of `params`, no optimiser step: returns (losses [4,B] weighted,
        un-LR'd per hypothesis (rgb, depth, mask, edge), grad [7,B] = d loss / d params with
        loss = sum_k sum_b lr_mult[b] * losses[k,b] / global_batch).  For callers that bring their own optimiser."""
        B = self.B
        if self._unchecked:
            self._run_check()
        grad = torch.empty((7, B), dtype=torch.float32, device=self.params.device)
        losses = torch.empty((4, B), dtype=torch.float32, device=self.params.device)
        it = min(self.it, self.max_iters - 1)
        _lib.check(self.lib.ddx_engine_eval(self.handle, it, grad.data_ptr(), losses.data_ptr(), _lib.stream_ptr()), "ddx_engine_eval")
        return losses, grad

    def loss(self, params=None):
        """Scalar total loss with autograd through the fused engine: `params` [7,B] (default: the engine's own tensor)
        gets its gradient from the analytic backward of the kernels, so any torch optimiser can drive the fused path:
            opt = torch.optim.Adam([p], lr=1e-2);  loss = eng.loss(p);  loss.backward();  opt.step()"""
        return _FusedLoss.apply(self.params if params is None else params, self)

    @property
    def slices(self):
        """(shade_slices, edge_slices) this engine runs with: what a shard of this batch passes to reproduce it bitwise."""
        auto = lambda grid: max(1, min(64, grid // self.B))
        d = self.desc
        n_roles = int(bool(d.use_rgb or d.use_depth or d.use_edge)) + int(bool(d.use_mask))
        return (d.shade_slices or auto(512 if n_roles == 2 else 1024), d.edge_slices or auto(1792))

    @property
    def cull_sign(self):
        """0 = both faces drawn; +-1 = back faces (snapped area of that sign) culled (decided by the first run / eval)."""
        return int(self.lib.ddx_engine_cull_sign(self.handle))

    @property
    def two_chains(self):
        """1 = long runs go out as two half-batch chains (the set-up's probe found a stream that runs beside the caller's), 0 = probed
        and refused (one chain), -1 = not eligible / not probed yet (ddx.h ddx_engine_two_chains)."""
        return int(self.lib.ddx_engine_two_chains(self.handle))

    def rewind(self, it=0):
        self.it = it

    def status(self):
        """dict(overflow (always 0), big_triangles (0/1: the tile pass ran), active_tiles, it (the last iteration drawn), n_seg,
        outside_view_volume (hypotheses of the last iteration whose bounding box had a corner at w <= 0 or |z| > w: near-plane
        clipping and two-sided drawing for those), flags (ddx.h status word 7; 0 after the check this call makes),
        repeated_runs) -- synchronises."""
        if self._unchecked:
            self._run_check()
        p = self.lib.ddx_engine_status_ptr(self.handle)
        off = p - self.scratch.data_ptr()
        st = self.scratch[off:off + 32].view(torch.int32).cpu().tolist()
        return dict(overflow=st[0], big_triangles=st[1], active_tiles=st[2], it=st[5] - 1, n_seg=st[4], outside_view_volume=st[6],
                    flags=st[7], repeated_runs=self.repeated_runs)

    def check(self):
        """status() with its invariants enforced: no overflow, and no void run left standing -- status() has already let
        ddx_engine_run_check act on a run whose bounded in-kernel wait ran out, so a flag that is STILL set means the results in
        the buffers are void (a run of a group that nobody checked, a raw-handle caller): raise rather than hand them out."""
        st = self.status()
        if st["overflow"]:
            raise RuntimeError("engine reported an internal overflow")
        if st["flags"]:
            raise RuntimeError(f"engine status flags {st['flags']}: the last run is void (a bounded in-kernel wait ran out) and has not been "
                               "repeated -- call finish() on the engine or on the group it ran in")
        return st

    def profile(self, it0=0, iters=5):
        """Per-kernel average launch duration in ms (hipEvents on the current stream).  Mutates params."""
        ms = (ctypes.c_float * KERNEL_NAMES_MAX)()
        names = (ctypes.c_char_p * KERNEL_NAMES_MAX)()
        if self._unchecked:
            self._run_check()
        k = self.lib.ddx_engine_profile(self.handle, it0, iters, ms, names, KERNEL_NAMES_MAX, _lib.stream_ptr())
        if k <= 0:
            _lib.check(k if k else -99, "ddx_engine_profile")
        if self.lib.ddx_engine_run_check(self.handle, _lib.stream_ptr()) == 1:  # (a profile is not repeated: measure again)
            raise RuntimeError("profile: the in-launch tile pass timed out; the engine now uses the separate launch -- profile again")
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    def losses(self):
        """[iters_done, 4, B] weighted un-LR'd per-hypothesis losses (rgb, depth, mask, edge)."""
        if self._unchecked:
            self._run_check()
        return self.loss_log[: self.it]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ddx_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class RefineEngineGroup:
    """Several RefineEngines advanced in lock step, ONE launch of each kernel per iteration for all of them (ddx_engine_group_*):
    the objects of one frame (BASELINE config 5: 4 objects x 64 hypotheses per GPU).  Every member ends with bit for bit the
    result it would get from its own run(); what changes is that the launches are shared, so the chip runs in its large-batch
    regime instead of one latency-bound 64-hypothesis launch after the other.  Members must have the same number of
    iterations (lr_sched length) and sit at the same iteration."""

    def __init__(self, engines):
        self.lib = _lib.load()
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("RefineEngineGroup needs at least one engine")
        if len({e.max_iters for e in self.engines}) != 1 or len({e.it for e in self.engines}) != 1:
            raise ValueError("the members of a group must share the schedule length and the current iteration")
        arr = (ctypes.c_void_p * len(self.engines))(*[e.handle for e in self.engines])
        h = ctypes.c_void_p()
        _lib.check(self.lib.ddx_engine_group_create(arr, len(self.engines), ctypes.byref(h)), "ddx_engine_group_create")
        self.handle = h
        self._unchecked = False
        self.repeated_runs = 0

    def _run_check(self):
        rc = self.lib.ddx_engine_group_run_check(self.handle, _lib.stream_ptr())
        if rc not in (0, 1):
            _lib.check(rc, "ddx_engine_group_run_check")
        self._unchecked = False
        for e in self.engines:  # (the members' runs were this run: checked with it)
            if e._group is not None and e._group() is self:
                e._unchecked = False
                e._group = None
        self.repeated_runs += int(rc == 1)
        return rc == 1

    @property
    def it(self):
        return self.engines[0].it

    def run(self, n=None):
        """n iterations (default: all remaining) of every member, asynchronously on the current stream."""
        e0 = self.engines[0]
        if len({e.it for e in self.engines}) != 1:  # (a member was run, rewound or given a new observation on its own)
            raise ValueError("the members of a group must sit at the same iteration: " + str([e.it for e in self.engines]))
        n = e0.max_iters - e0.it if n is None else n
        if self._unchecked:
            self._run_check()
        for e in self.engines:  # (a member with an unchecked run of its own: validated before the group builds on it)
            if e._unchecked:
                e._run_check()
        _lib.check(self.lib.ddx_engine_group_run(self.handle, e0.it, n, _lib.stream_ptr()), "ddx_engine_group_run")
        for e in self.engines:
            e.it += n
            if n > 0:  # whoever synchronises a member next (finish / status / losses / a further run) checks the GROUP's run
                e._unchecked = True
                e._group = weakref.ref(self)
        self._unchecked = n > 0

    def invalidate(self):
        """A member was re-created in place (same handle, other buffers): its table row is uploaded again by the next run.  Not
        needed after new_observation(): the native side notices a member's new set-up by itself."""
        _lib.check(self.lib.ddx_engine_group_invalidate(self.handle), "ddx_engine_group_invalidate")

    def finish(self):
        """Synchronise and validate (ddx_engine_group_run_check); True when the group's last run had to be repeated."""
        if self._unchecked:
            return self._run_check()
        torch.cuda.current_stream().synchronize()
        return False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ddx_engine_group_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

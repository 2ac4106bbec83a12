"""Run-to-run determinism of the engine GROUP launches (`-m gpu`).  Round 3 saw a build of shade_group_kernel give different losses
from run to run (the inline tile pass); nothing in the suite would have caught a latent race in the shipped
kernels, so this repeats ddx_engine_group_run of four unlike members 50 times per shading-grid size and compares every bit --
parameters, loss log, pose log -- with the first repetition and with each member's own run."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shade_grid", [256, 512, 1024])
def test_fifty_group_runs_give_the_same_bits(shade_grid):
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl
    from tests.test_gpu_group import _members

    dev = torch.device("cuda")
    n_it, reps = 5, 50
    old = os.environ.get("DDX_SHADE_GRID")
    os.environ["DDX_SHADE_GRID"] = str(shade_grid)  # (read by the native side when an engine is created and when it launches)
    try:
        specs = _members(dev, n_it, "adam")
        engs = [wl.engine_for(w, lrs, optimizer="adam") for w, lrs in specs]
        assert engs[0][0].lib.ddx_engine_scratch_bytes is not None
        grp = dd.RefineEngineGroup([e for e, _ in engs])
        first = None
        for rep in range(reps):
            if rep:
                for (e, _), (w, _) in zip(engs, specs):
                    e.new_observation(params=w["params0"])
            grp.run()
            grp.finish()
            snap = [(p.clone(), e.losses().clone(), e.mtx_log.clone()) for e, p in engs]
            for e, _ in engs:
                e.check()
            if first is None:
                first = snap
                continue
            for k, (a, b) in enumerate(zip(first, snap)):
                assert all(torch.equal(x, y) for x, y in zip(a, b)), f"repetition {rep}: member {k} differs from the first run (shade grid {shade_grid})"
        # ... and the bits are those of each member's own run with the same grid
        for k, (w, lrs) in enumerate(specs):
            e, p = wl.engine_for(w, lrs, optimizer="adam")
            e.run()
            e.finish()
            assert torch.equal(p, first[k][0]) and torch.equal(e.losses(), first[k][1]) and torch.equal(e.mtx_log, first[k][2]), f"member {k} alone"
    finally:
        if old is None:
            os.environ.pop("DDX_SHADE_GRID", None)
        else:
            os.environ["DDX_SHADE_GRID"] = old


@pytest.mark.parametrize("name,B,dist,must", [("lowpoly", 16, None, True), ("hugetri", 8, None, True), ("midpoly", 24, 2.5, False), ("cfg2", 16, 0.9, False)])
def test_tile_pass_inside_the_shading_launch_equals_the_separate_launch(name, B, dist, must):
    """Round 4: where no large triangle is expected the tile-pass launch is dropped and worker workgroups in the first slab of the
    shading launch run the pass for a hypothesis that has some after all (engine.hip big_worker_wg).  Forced on for meshes that
    are ALL large triangles, a mix, and a dense mesh with the camera almost inside it (near-clipped triangles): parameters, loss
    log, pose log and status equal the separate launch bit for bit, alone and as an engine group."""
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    n_it = 6
    w = wl.build(name, dev, B=B, distance=dist)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    old = os.environ.get("DDX_BIG_INLINE")
    out = {}
    try:
        for mode in ("0", "1"):
            os.environ["DDX_BIG_INLINE"] = mode  # (read when an engine is created)
            e, p = wl.engine_for(w, lrs, optimizer="adam")
            e.run(2)
            e.run()
            e.finish()
            st = e.check()
            e2, p2 = wl.engine_for(w, lrs, optimizer="adam")
            e3, p3 = wl.engine_for(w, lrs, optimizer="adam")
            g = dd.RefineEngineGroup([e2, e3])
            g.run()
            g.finish()
            out[mode] = (p.clone(), e.losses().clone(), e.mtx_log.clone(), st, p2.clone(), e2.losses().clone(), p3.clone())
    finally:
        if old is None:
            os.environ.pop("DDX_BIG_INLINE", None)
        else:
            os.environ["DDX_BIG_INLINE"] = old
    a, b = out["0"], out["1"]
    print(name, "large triangles in the last iteration:", a[3]["big_triangles"], "outside the view volume:", a[3]["outside_view_volume"])
    assert a[3]["big_triangles"] == 1 or not must, "the case must take the tile pass"
    assert a[3] == b[3]
    for x, y in zip(a[:3] + a[4:], b[:3] + b[4:]):
        assert torch.equal(x, y)
    assert torch.equal(b[0], b[4]) and torch.equal(b[0], b[6]) and torch.equal(b[1], b[5])  # group members == the engine alone


def test_bounded_wait_of_the_in_launch_tile_pass_falls_back_to_the_separate_launch():
    """Round 5: nothing about the tile pass inside the shading launch may depend on dispatch order or placement.  With the debug
    switch DDX_DEBUG_REVERSE_SLABS the worker slab gets the LARGEST block ids: the 64 x 8 x 2 = 1024 shading workgroups of a batch in
    which every hypothesis has large triangles fill the chip and wait for workers that cannot start.  The wait is bounded
    (DDX_BIG_WAIT_US), sets status word 7, every kernel still terminates, and ddx_engine_run_check -- through RefineEngine.finish(),
    through the NaN loss dist.run_and_select finds in the selection row, and for an engine group -- restores the run's start,
    switches the engine to the separate launch and repeats the run: parameters, loss log, pose log and the selected hypothesis
    equal an engine that used the separate launch from the start, bit for bit."""
    import diffdope_amd as dd
    from diffdope_amd import dist as ddist, workloads as wl

    dev = torch.device("cuda")
    n_it = 6
    w = wl.build("hugetri", dev, B=64)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    keys = ("DDX_BIG_INLINE", "DDX_DEBUG_REVERSE_SLABS", "DDX_BIG_WAIT_US")
    old = {k: os.environ.get(k) for k in keys}

    def scenario():
        e, p = wl.engine_for(w, lrs, optimizer="adam")
        e.run(2)
        e.run()
        rep = e.finish()
        st = e.check()
        e1, p1 = wl.engine_for(w, lrs, optimizer="adam")
        best = ddist.run_and_select(e1, n_it, lo=5)
        e2, p2 = wl.engine_for(w, lrs, optimizer="adam")
        e3, p3 = wl.engine_for(w, lrs, optimizer="adam")
        g = dd.RefineEngineGroup([e2, e3])
        g.run()
        grep = g.finish()
        return dict(p=p.clone(), ll=e.losses().clone(), ml=e.mtx_log.clone(), st=st, rep=e.repeated_runs, rep_now=rep, best=best, p1=p1.clone(),
                    rep1=e1.repeated_runs, p2=p2.clone(), l2=e2.losses().clone(), p3=p3.clone(), grep=grep, flags2=e2.status()["flags"])

    try:
        os.environ.update(DDX_BIG_INLINE="0")
        os.environ.pop("DDX_DEBUG_REVERSE_SLABS", None)
        ref = scenario()
        os.environ.update(DDX_BIG_INLINE="1", DDX_DEBUG_REVERSE_SLABS="1", DDX_BIG_WAIT_US="300")
        t0 = time.time()
        got = scenario()
        took = time.time() - t0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ref["st"]["big_triangles"] == 1 and ref["rep"] == 0 and ref["rep1"] == 0 and not ref["grep"]
    assert got["rep"] == 1 and got["rep1"] == 1 and got["grep"], "the reversed slab order must run into the bounded wait"
    assert not got["rep_now"]  # (the first run() was repeated by the check in front of the second; nothing left for finish())
    assert got["st"]["flags"] == 0 and got["flags2"] == 0 and took < 120
    for k in ("p", "ll", "ml", "p1", "p2", "l2", "p3"):
        assert torch.equal(ref[k], got[k]), k
    assert ref["best"][:2] == got["best"][:2] and torch.equal(ref["best"][2], got["best"][2])
    assert {k: v for k, v in ref["st"].items() if k != "repeated_runs"} == {k: v for k, v in got["st"].items() if k != "repeated_runs"}


@pytest.mark.parametrize("name,B", [("cfg2", 64), ("cfg4", 32), ("cfg3", 48), ("cfg50k64", 64)])
def test_long_runs_on_two_streams_equal_the_single_chain(name, B):
    """Round 4: a run of 16+ iterations goes out as two chains of half-batch launches, one on a stream the engine owns, forked
    from and joined to the caller's stream by events (engine.hip engine_run_impl, ddx.h single_stream).  Every hypothesis runs
    the slots, slices and sums of the full launches: parameters, loss log, pose log and status equal the single chain bit for
    bit -- on the caller's stream WITHOUT a host synchronisation before the comparison (the join orders the copy after both
    chains), run after run, on a side stream of the caller's, and through the fused selection."""
    from diffdope_amd import dist as ddist, workloads as wl

    dev = torch.device("cuda")
    n_it = 120
    w = wl.build(name, dev, B=B)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    out = {}
    for single in (True, False):
        e, p = wl.engine_for(w, lrs, optimizer="adam", single_stream=single)
        e.run(50)
        snap = p.clone()  # (enqueued on the caller's stream right behind the run: sees both chains' parameters)
        e.run(10)         # (short: one chain)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            e.run(50)
            snap2 = p.clone()
        torch.cuda.current_stream().wait_stream(st)
        best = ddist.run_and_select(e, 10, lo=3)
        e.finish()
        out[single] = (snap, snap2, p.clone(), e.losses().clone(), e.mtx_log.clone(), best, e.check())
    a, b = out[True], out[False]
    assert a[6] == b[6]
    assert a[5][:2] == b[5][:2] and torch.equal(a[5][2], b[5][2])
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)


def test_selection_inside_the_last_kernel_equals_the_selection_kernel():
    """ddx_engine_run_select (round 4): the arg-min over the local hypotheses folded into finish_kernel -- atomicMin of (loss bits,
    index) + an arrival count -- against ddx_select_best on the same iteration's rows: same index (ties to the LOWEST index: two
    hypotheses are exact duplicates), same loss bits, same pose; into pinned host memory and into a device row; twice in a row
    (the words are re-armed by the launch that used them); through dist.run_and_select."""
    from diffdope_amd import dist as ddist, workloads as wl

    dev = torch.device("cuda")
    for name, B in (("cfg2", 24), ("cfg4", 40)):
        w = wl.build(name, dev, B=B)
        p0 = w["params0"].clone()
        p0[:, 7] = p0[:, 3]      # exact duplicates: they tie
        p0[:, B - 1] = p0[:, 3]
        lrm = w["lr_mult"].clone()
        lrm[7] = lrm[3]
        lrm[B - 1] = lrm[3]
        w = dict(w, params0=p0, lr_mult=lrm)
        n_it = 9
        eng, params = wl.engine_for(w, wl.bench_lr_schedule(n_it, "adam"), optimizer="adam")
        used = [i for i, k in enumerate(("rgb", "depth", "mask", "edge")) if w["weights"].get(k) is not None]
        mask = sum(1 << i for i in used)
        pinned = torch.empty((1, 18), dtype=torch.float32, pin_memory=True)
        row = torch.zeros(18, dtype=torch.float32, device=dev)
        done = 0
        for n, out in ((3, pinned[0]), (2, row), (1, pinned[0])):
            eng.run_select(out, n, lo=100)
            eng.finish()
            done += n
            ref = ddist.global_argmin_fused(eng.loss_log[done - 1], mask, eng.mtx_log[done - 1], lo=100)
            got = out.cpu().numpy()
            assert int(got[1]) == ref[0] and float(got[0]) == ref[1], (name, n, got[:2], ref[:2])
            assert torch.equal(torch.from_numpy(got[2:].copy()).reshape(4, 4), ref[2])
        assert torch.equal(params[:, 3], params[:, 7])  # (the duplicates stayed duplicates, so the tie is real)
        gi, gl, gm = ddist.run_and_select(eng, 2, lo=5)
        ref = ddist.global_argmin_fused(eng.loss_log[eng.it - 1], mask, eng.mtx_log[eng.it - 1], lo=5)
        assert (gi, gl) == (ref[0], ref[1]) and torch.equal(gm, ref[2])
        eng.check()


def test_row_restricted_materialising_path_equals_the_full_frame_one():
    """render_texture_batch(fused=True): the passes restricted to the rows each hypothesis draws into (ddx_*_rows, round 4) against
    the same passes over the whole frame: rgb, depth and mask bit for bit, the gradient with respect to the pose matrices up to the order of its atomic additions --
    hypotheses in the middle of the frame, at its upper and lower border, one that draws nothing at all, and one covering
    every row."""
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl
    from diffdope_amd.render import RasterizeContext, render_texture_batch

    dev = torch.device("cuda")
    for name in ("cfg2", "cfg4"):
        w = wl.build(name, dev, B=6)
        p = w["params0"].clone()
        p[5, 1] += 2.2    # towards the upper border
        p[5, 2] -= 2.4    # the lower border
        p[4, 3] += 40.0   # out of the frame: draws nothing
        p[6, 4] = -1.6    # close: every row
        B, H, W = 6, w["H"], w["W"]
        ex = lambda t: t[None].expand(B, *t.shape)
        kw = dict(uv=ex(w["uv"]), uv_idx=ex(w["tri"]), tex=ex(w["tex"])) if w["uv"] is not None else dict(vtx_color=ex(w["vtx_color"]))
        outs = []
        for restrict in (False, True):
            q = p[:4].T / torch.norm(p[:4].T, dim=1, keepdim=True)
            mtx = dd.matrix_batch_44_from_position_quat(q=q, p=p[4:].T).detach().requires_grad_(True)
            r = render_texture_batch(RasterizeContext(), ex(w["proj"]), mtx, ex(w["pos"]), ex(w["tri"]), [H, W], fused=True, restrict_rows=restrict, **kw)
            g = torch.Generator(device="cpu").manual_seed(7)
            wr, wd, wm = (torch.rand(r[k].shape, generator=g).to(dev) for k in ("rgb", "depth", "mask"))
            loss = (r["rgb"] * wr).sum() + (r["depth"] * wd).sum() + (r["mask"] * wm).sum()
            (grad,) = torch.autograd.grad(loss, mtx)
            outs.append((r["rgb"].detach().clone(), r["depth"].detach().clone(), r["mask"].detach().clone(), grad.clone()))
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(a, b)
        # (the backward passes accumulate with floating-point atomics: equal up to the order of the additions)
        ga, gb = outs[0][3], outs[1][3]
        assert torch.allclose(ga, gb, rtol=2e-4, atol=2e-4 * float(ga.abs().max()))
        assert float(outs[0][2][1].sum()) > 0 and float(outs[0][2][3].sum()) == 0  # (the shifted one draws, the far one does not)

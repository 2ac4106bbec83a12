"""The one-launch form of a run (`-m gpu`; engine.hip run_kernel, ddx.h one_launch_run; round 5).

ddx_engine_run as ONE kernel: every hypothesis is advanced through all its iterations (diffdope/diffdope.py:1656-1714) by a team of
workgroups that meet at two team barriers per iteration.  It must give the bits of the launch form whatever the dispatcher does:
teams are formed per XCD at run time, every phase works for a team of any size, every wait is bounded, and a run that cannot be
completed in this form (a wait ran out, a hypothesis met a large triangle) is repeated as launches by ddx_engine_run_check."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)  # (read by the native side when an engine is created / a run is launched)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _scenario(w, lrs, n_it, **kw):
    """A run in pieces, one of them on a side stream, the last one through the fused selection."""
    from diffdope_amd import dist as ddist, workloads as wl

    e, p = wl.engine_for(w, lrs, optimizer="adam", **kw)
    e.run(7)
    snap = p.clone()
    e.run(1)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        e.run(n_it - 12)
    torch.cuda.current_stream().wait_stream(st)
    forms = [e.run_form]
    best = ddist.run_and_select(e, 4, lo=3)
    forms.append(e.run_form)
    e.finish()
    return dict(snap=snap, p=p.clone(), ll=e.losses().clone(), ml=e.mtx_log.clone(), best=best, st=e.check(), forms=forms, rep=e.repeated_runs, eng=e)


def _same(a, b):
    for k in ("snap", "p", "ll", "ml"):
        assert torch.equal(a[k], b[k]), k
    assert a["best"][:2] == b["best"][:2] and torch.equal(a["best"][2], b["best"][2])
    drop = lambda d: {k: v for k, v in d.items() if k != "repeated_runs"}
    assert drop(a["st"]) == drop(b["st"])


@pytest.mark.parametrize("name,B", [("cfg2", 64), ("cfg2", 40), ("cfg4", 32), ("cfg3ref", 128), ("cfg50k64", 64), ("cfg4", 256)])
def test_one_launch_run_equals_the_launches(name, B):
    """Parameters, loss log, pose log, status words and the selected hypothesis of a run in pieces: bit for bit those of the launch
    form -- 64 teams of 16, 40 hypotheses (a batch that is no power of two), depth + rgb, a mesh that takes
    the compacting rasteriser variant as launches (the plain one here), and more hypotheses than teams of four fit (B = 256)."""
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    n_it = 20
    w = wl.build(name, dev, B=B)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    ref = _scenario(w, lrs, n_it)
    got = _scenario(w, lrs, n_it, one_launch_run=True)
    assert ref["forms"] == [0, 0] and got["forms"] == [1, 1] and got["rep"] == 0
    _same(ref, got)


@pytest.mark.parametrize("env", [dict(DDX_DEBUG_RUN=1), dict(DDX_RUN_TEAM=32, DDX_RUN_GRID=1000), dict(DDX_RUN_GRID=2500), dict(DDX_RUN_TEAM=4, DDX_RUN_GRID=72),
                                 dict(DDX_RUN_TEAM=1, DDX_RUN_GRID=9)])
def test_teams_of_any_size_give_the_same_bits(env):
    """Nothing may depend on how many workgroups a team gets or where they run: every team closed by its first member at once (sizes
    1..16 as the arrivals fall), teams of 32 from a grid that is not a multiple of anything (the last team of each XCD closes
    undersized after DDX_RUN_CLOSE_US), a grid of 2 500 workgroups where about 1 024 are resident (the late ones find the queue
    empty and leave; the control block is re-armed by the last), 18 teams of four and nine lone workgroups working through the
    queue of 64 hypotheses -- the same bits as the launches, twice in a row on the same engine."""
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    n_it = 20
    w = wl.build("cfg2", dev, B=64)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    ref = _scenario(w, lrs, n_it)
    with _env(DDX_RUN_CLOSE_US=30, **env):
        got = _scenario(w, lrs, n_it, one_launch_run=True)
        assert got["forms"] == [1, 1] and got["rep"] == 0 and got["st"]["flags"] == 0
        _same(ref, got)
        e = got["eng"]
        e.new_observation(params=w["params0"])  # (the same engine again: its control block must have been left all zero)
        e.run()
        e.finish()
        assert e.run_form == 1 and torch.equal(e.params, ref["p"]) and e.check()["flags"] == 0


def test_a_wait_that_runs_out_falls_back_to_the_launches():
    """DDX_DEBUG_RUN=2: the first member of a team never publishes its size, so every other member's (bounded) wait runs out.  The
    launch terminates, status word 7 has bit 1, the selection row says NaN, and ddx_engine_run_check -- through finish() and
    through dist.run_and_select -- puts the run's start back, repeats it as launches and keeps the engine there: the same bits as
    an engine that ran launches from the start."""
    from diffdope_amd import dist as ddist, workloads as wl

    dev = torch.device("cuda")
    n_it = 12
    w = wl.build("cfg2", dev, B=64)
    lrs = wl.bench_lr_schedule(n_it, "adam")

    def scenario(**kw):
        e, p = wl.engine_for(w, lrs, optimizer="adam", **kw)
        e.run(5)
        rep = e.finish()
        form = e.run_form
        best = ddist.run_and_select(e, 7, lo=2)
        e1, p1 = wl.engine_for(w, lrs, optimizer="adam", **kw)
        best1 = ddist.run_and_select(e1, n_it, lo=2)
        return dict(p=p.clone(), ll=e.losses().clone(), best=best, st=e.check(), rep=rep, form=form, reps=e.repeated_runs, p1=p1.clone(), best1=best1,
                    reps1=e1.repeated_runs, form1=e1.run_form)

    ref = scenario()
    with _env(DDX_DEBUG_RUN=2, DDX_BIG_WAIT_US=300):
        t0 = time.time()
        got = scenario(one_launch_run=True)
        took = time.time() - t0
    assert got["rep"] and got["reps"] == 1 and got["reps1"] == 1 and got["form"] == 0 and got["form1"] == 0 and took < 120
    assert ref["reps"] == 0 and got["st"]["flags"] == 0
    for k in ("p", "ll", "p1"):
        assert torch.equal(ref[k], got[k]), k
    for k in ("best", "best1"):
        assert ref[k][:2] == got[k][:2] and torch.equal(ref[k][2], got[k][2])


def test_a_large_triangle_falls_back_to_the_launches():
    """The teams do not run the tile pass: the one-launch form is chosen where the set-up expects no large triangle.  Forced onto a
    dense mesh with the camera almost inside it (near-clipped triangles take the tile pass): status word 7 gets bit 2, and the run is
    repeated as launches with the bits of an engine that never tried."""
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    n_it = 6
    w = wl.build("cfg2", dev, B=32, distance=0.9)
    lrs = wl.bench_lr_schedule(n_it, "adam")
    out = {}
    with _env(DDX_BIG_INLINE=1, DDX_SCATTER_MODE=0):  # (a close-up takes the exchange variant of the rasteriser, which has no one-launch form)
        for one in (False, True):
            e, p = wl.engine_for(w, lrs, optimizer="adam", one_launch_run=one)
            e.run()
            rep = e.finish()
            out[one] = (p.clone(), e.losses().clone(), e.mtx_log.clone(), e.check(), rep, e.run_form)
    a, b = out[False], out[True]
    assert a[3]["big_triangles"] >= 1, "the case must take the tile pass"
    assert not a[4] and b[4] and b[5] == 0
    assert all(torch.equal(x, y) for x, y in zip(a[:3], b[:3]))
    drop = lambda d: {k: v for k, v in d.items() if k != "repeated_runs"}
    assert drop(a[3]) == drop(b[3])

"""Engine groups (ddx_engine_group_*, RefineEngineGroup): several engines -- the objects of one frame, BASELINE config 5 -- advanced
with one launch of each kernel per iteration.  Every member must end with bit for bit the result of its own run()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _members(dev, n_it, optimizer):
    """Four unlike engines: dense textured meshes with different loss sets and batch sizes, an untextured one with vertex colours,
    a low-polygon mesh (large triangles: the tile pass, and the 64-thread step_kernel variant) on another frame size."""
    from diffdope_amd import workloads as wl

    specs = [("cfg2", 24, dict(rgb=0.7, mask=1.0), None), ("cfg50k64", 16, dict(rgb=0.7, depth=1.0, edge=1.0), None),
             ("cfg4", 40, dict(depth=1.0, mask=1.0), 5.0), ("lowpoly", 8, dict(depth=1.0, mask=1.0), None)]
    out = []
    for name, B, weights, dist in specs:
        w = wl.build(name, dev, B=B, distance=dist)
        w = dict(w, weights=weights)
        out.append((w, wl.bench_lr_schedule(n_it, optimizer)))
    return out


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_group_members_match_their_own_runs_bit_for_bit(optimizer):
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    n_it = 7
    specs = _members(dev, n_it, optimizer)
    alone = []
    for w, lrs in specs:
        e, p = wl.engine_for(w, lrs, optimizer=optimizer)
        e.run(3)
        e.run()  # (two calls: a run that continues where the last one stopped)
        e.finish()
        e.check()
        alone.append((p.clone(), e.losses().clone(), e.mtx_log.clone(), e.status()))
    engs = [wl.engine_for(w, lrs, optimizer=optimizer) for w, lrs in specs]
    grp = dd.RefineEngineGroup([e for e, _ in engs])
    grp.run(3)
    grp.run()
    grp.finish()
    for (e, p), (p0, l0, m0, st0) in zip(engs, alone):
        st = e.check()
        assert e.it == n_it and st["active_tiles"] == st0["active_tiles"] and st["big_triangles"] == st0["big_triangles"]
        assert torch.equal(p, p0) and torch.equal(e.losses(), l0) and torch.equal(e.mtx_log, m0)
    assert alone[3][3]["big_triangles"] == 1 and alone[0][3]["big_triangles"] == 0  # (the low-polygon member took the tile pass)
    # a member keeps working on its own after the group, and a new observation of one member is picked up by the next group run
    e0, p0 = engs[0]
    e0.new_observation(params=specs[0][0]["params0"])
    for e, _ in engs[1:]:
        e.new_observation()
    grp.run(2)
    grp.finish()
    ref, pr = wl.engine_for(specs[0][0], specs[0][1], optimizer=optimizer)
    ref.run(2)
    ref.finish()
    assert torch.equal(p0, pr) and torch.equal(e0.losses(), ref.losses())


def test_group_refuses_members_out_of_step():
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    w = wl.build("tiny", dev)
    a, _ = wl.engine_for(w, [0.1] * 4)
    b, _ = wl.engine_for(w, [0.1] * 5)
    with pytest.raises(ValueError):
        dd.RefineEngineGroup([a, b])
    c, _ = wl.engine_for(w, [0.1] * 4)
    c.run(1)
    with pytest.raises(ValueError):
        dd.RefineEngineGroup([a, c])
    with pytest.raises(RuntimeError):
        dd.RefineEngineGroup([a, a])

"""gloo tests (CPU, world size 2 and 8) of the only exchange step of the path: the global arg-min over the
hypothesis shards (diffdope_amd/dist.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, losses, mtx, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffdope_amd.dist import global_argmin, shard_range

    res = []
    for case in range(losses.shape[0]):
        lo, hi = shard_range(losses.shape[1], rank, world)
        gi, gl, gm = global_argmin(torch.tensor(losses[case, lo:hi]), torch.tensor(mtx[case, lo:hi]), lo=lo)
        res.append((gi, gl, gm.numpy().copy()))
    # multi-object table: object i is owned by rank i % world
    from diffdope_amd.bop import owner_of
    from diffdope_amd.dist import merge_object_tables

    n_obj = min(32, losses.shape[1]) if world == 8 else 6
    full = torch.tensor(mtx[0].reshape(losses.shape[1], 16)[:n_obj, :])
    tab = torch.zeros(n_obj, 18)
    for i in range(n_obj):
        if owner_of(i, world) == rank:
            tab[i, 0], tab[i, 1], tab[i, 2:] = float(losses[0, i]), i, full[i]
    res.append(merge_object_tables(tab).numpy().copy())
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_global_argmin_two_ranks_gloo():
    rng = np.random.RandomState(0)
    B, cases = 10, 4
    losses = rng.uniform(size=(cases, B)).astype(np.float32)
    losses[1, 7] = losses[1, 2] = losses[1].min() - 0.1  # a tie across the two shards -> lowest global index
    losses[2, 9] = -1.0  # winner on the last rank
    losses[3, 0] = -1.0  # winner on rank 0
    mtx = rng.normal(size=(cases, B, 4, 4)).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, losses, mtx, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        tab = got[rank][cases]
        np.testing.assert_allclose(tab[:, 0], losses[0, :6], rtol=1e-6)
        np.testing.assert_allclose(tab[:, 2:], mtx[0].reshape(B, 16)[:6], rtol=1e-6)
        assert list(tab[:, 1]) == list(range(6))
    for case in range(cases):
        expect = int(np.argmin(losses[case]))
        for rank in (0, 1):
            gi, gl, gm = got[rank][case]
            assert gi == expect
            assert abs(gl - float(losses[case, expect])) < 1e-6
            np.testing.assert_allclose(gm, mtx[case, expect], rtol=1e-6)


def test_eight_ranks_gloo_config4_shards_and_config5_object_table():
    """The node the driver measures on has 8 ranks: BASELINE configs[3] (512 hypotheses, 64 per rank) and configs[4] (32 objects,
    4 per rank) through the same code at world size 8 -- shard ranges, the [8,18] table, ties ACROSS ranks (lowest global index
    wins, as torch.argmin over the whole batch), winners on the first / a middle / the last rank, and the 32-row object table."""
    from diffdope_amd.bop import owner_of
    from diffdope_amd.dist import shard_range

    world, B = 8, 512
    # the shards of the fixed job: contiguous, complete, 64 each; an uneven job differs by at most one
    assert [shard_range(B, r, world) for r in range(world)] == [(64 * r, 64 * r + 64) for r in range(world)]
    cover = [shard_range(515, r, world) for r in range(world)]
    assert cover[0][0] == 0 and cover[-1][1] == 515 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert {hi - lo for lo, hi in cover} == {64, 65}
    assert [sum(owner_of(i, world) == r for i in range(32)) for r in range(world)] == [4] * 8
    rng = np.random.RandomState(1)
    cases = 5
    losses = rng.uniform(0.5, 1.5, size=(cases, B)).astype(np.float32)
    losses[1, [70, 200, 449]] = 0.25   # a three-way tie on ranks 1, 3 and 7 -> global index 70
    losses[2, 511] = 0.125             # winner = the last hypothesis of the last rank
    losses[3, 0] = 0.125               # winner = the first hypothesis of rank 0
    losses[4, [320, 319]] = 0.0625     # a tie across the boundary of ranks 4 / 5 -> 319
    mtx = rng.normal(size=(cases, B, 4, 4)).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, losses, mtx, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [int(np.argmin(losses[c])) for c in range(cases)]
    assert expect[1] == 70 and expect[2] == 511 and expect[3] == 0 and expect[4] == 319
    for rank in range(world):
        for case in range(cases):
            gi, gl, gm = got[rank][case]
            assert gi == expect[case], (rank, case, gi)
            assert abs(gl - float(losses[case, expect[case]])) < 1e-6
            np.testing.assert_allclose(gm, mtx[case, expect[case]], rtol=1e-6)
        tab = got[rank][cases]  # 32 objects, 4 per rank, one all_reduce
        assert tab.shape == (32, 18) and list(tab[:, 1]) == list(range(32))
        np.testing.assert_allclose(tab[:, 0], losses[0, :32], rtol=1e-6)
        np.testing.assert_allclose(tab[:, 2:], mtx[0].reshape(B, 16)[:32], rtol=1e-6)


def test_global_argmin_single_process_matches_reference_get_argmin(golden_dir):
    """World size 1 (no process group): same selection as DiffDope.get_argmin on the golden losses."""
    from diffdope_amd.dist import global_argmin

    g = np.load(os.path.join(golden_dir, "g6_argmin.npz"))
    stacked = np.stack([g[f"v_{k}"][-1] for k in ("rgb", "depth", "mask_selection")]).mean(0)
    gi, gl, _ = global_argmin(torch.tensor(stacked), torch.zeros(stacked.shape[0], 4, 4))
    assert gi == int(g["argmin"])

"""Hypothesis sharding over the GPUs of one node.

Hypotheses are independent (per-hypothesis parameters diffdope.py:1019-1026 and losses :534-544), so
rank r simply owns a contiguous slice of the batch; mesh, texture and observed images are replicated.
The only exchange is the final arg-min over all hypotheses (get_argmin / get_pose semantics,
diffdope.py:1488-1513,1618-1632): every rank puts (loss, global index, 4x4 pose) of its local best into
its row of a zeroed [world,18] buffer, ONE all_reduce(SUM) over RCCL/xGMI (576 B at 8 GPUs: latency
bound) makes the table identical everywhere, and each rank takes the row-arg-min.
"""
import torch


def shard_range(total, rank, world):
    """Contiguous slice [lo,hi) of `total` hypotheses owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_best(per_hyp_loss, mtx):
    """per_hyp_loss [B_local], mtx [B_local,4,4] -> (loss scalar tensor, local index tensor, mtx [16])."""
    idx = torch.argmin(per_hyp_loss)
    return per_hyp_loss[idx], idx, mtx[idx].reshape(16)


def global_argmin(per_hyp_loss, mtx, lo=0, group=None):
    """Arg-min over the hypotheses of ALL ranks with a single all_reduce.

    per_hyp_loss [B_local] (mean over loss keys of the last-step losses, diffdope.py:1505-1511),
    mtx [B_local,4,4]; lo = global index of this rank's first hypothesis.
    Returns (global_index:int, loss:float, pose [4,4] tensor), identical on every rank.
    Ties resolve to the lowest global index (what torch.argmin over the concatenated batch gives).
    """
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    loss, idx, m = local_best(per_hyp_loss, mtx)
    table = torch.zeros((world, 18), dtype=torch.float64 if per_hyp_loss.device.type == "cpu" else torch.float32,
                        device=per_hyp_loss.device)
    table[rank, 0] = loss
    table[rank, 1] = (idx + lo).to(table.dtype)
    table[rank, 2:] = m.to(table.dtype)
    if dist.is_initialized():  # (also with one rank: the collective path is the same code at every world size)
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    # row arg-min, ties -> lowest global index
    losses, gidx = table[:, 0], table[:, 1]
    best = torch.min(losses)
    cand = torch.where(losses == best, gidx, torch.full_like(gidx, float("inf")))
    row = int(torch.argmin(cand))
    return int(table[row, 1].item()), float(table[row, 0].item()), table[row, 2:].reshape(4, 4).to(mtx.dtype)


def global_argmin_fused(loss_rows, row_mask, mtx, lo=0, group=None):
    """global_argmin with the local selection done by one device kernel (ddx_select_best) and ONE host
    synchronisation: loss_rows [4,B] (one iteration's row block of RefineEngine.loss_log), row_mask = bit r set when
    loss row r takes part in the mean, mtx [B,16] or [B,4,4].  Same result as
    global_argmin(loss_rows[used].mean(0), mtx, lo), with the pose returned as a HOST tensor at every world size."""
    import torch.distributed as dist

    from . import _lib

    lib = _lib.load()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = loss_rows.shape[1]
    if not dist.is_initialized():
        # one process: the kernel stores its 18 floats straight into pinned host memory (mapped into the GPU's address space):
        # no device -> host copy to launch, the stream synchronisation is the only wait
        with _PINNED_LOCK:  # (one pinned row per process: held until its 18 floats have been read)
            table = _pinned_row()
            _lib.check(lib.ddx_select_best(loss_rows.data_ptr(), int(row_mask), B, mtx.data_ptr(), int(lo), table.data_ptr(),
                                           _lib.stream_ptr()), "ddx_select_best")
            torch.cuda.current_stream().synchronize()
            t = table.numpy()
            return int(t[0, 1]), float(t[0, 0]), table[0, 2:].clone().reshape(4, 4)
    table = torch.zeros((world, 18), dtype=torch.float32, device=loss_rows.device)
    _lib.check(lib.ddx_select_best(loss_rows.data_ptr(), int(row_mask), B, mtx.data_ptr(), int(lo), table[rank].data_ptr(),
                                   _lib.stream_ptr()), "ddx_select_best")
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    t = table.cpu().numpy()  # the one synchronisation (device -> host copy of world x 18 floats)
    losses, gidx = t[:, 0], t[:, 1]
    cand = [(gidx[r], r) for r in range(world) if losses[r] == losses.min()]  # ties -> lowest global index
    row = min(cand)[1]
    return int(t[row, 1]), float(t[row, 0]), torch.from_numpy(t[row, 2:].copy()).reshape(4, 4)  # (the pose on the HOST, as in the one-process path)


def run_and_select(eng, n, lo=0, group=None, use_graph=False):
    """eng.run(n) + global_argmin_fused of its last iteration with the local selection folded into the run's LAST kernel
    (ddx_engine_run_select): no selection launch, and in one process the 18 floats land in pinned host memory, so the end of a
    run is one kernel and one synchronisation.  Returns (global index, loss, pose [4,4] on the host), identical on every rank.
    A NaN loss in a row says that rank's run was void (its in-launch tile pass timed out, ddx.h ddx_engine_run_check): the owner
    repeats the run -- which rewrites its row -- and the exchange is made once more."""
    import math

    import torch.distributed as dist

    if not dist.is_initialized():
        with _PINNED_LOCK:
            table = _pinned_row()
            eng.run_select(table, n, lo=lo, use_graph=use_graph)
            torch.cuda.current_stream().synchronize()
            if math.isnan(float(table[0, 0])):
                eng.finish()  # (repeats the run with the separate tile-pass launch; the same pinned row receives the result)
            else:
                eng._unchecked = False
            t = table.numpy()
            return int(t[0, 1]), float(t[0, 0]), table[0, 2:].clone().reshape(4, 4)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # (the exchange table is kept per (device, group, stream) and zeroed again right AFTER it has been read -- off the path of the
    # next call, which would otherwise begin with a fill launch in front of its first kernel; the lock keeps two threads of a
    # process from summing each other's rows)
    key = (eng.params.device, world, id(group), torch.cuda.current_stream().cuda_stream)
    with _TABLES_LOCK:
        table = _TABLES.get(key)
        if table is None:
            table = _TABLES[key] = torch.zeros((world, 18), dtype=torch.float32, device=eng.params.device)
        try:
            eng.run_select(table[rank], n, lo=lo, use_graph=use_graph)
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
            t = table.cpu().numpy()
            if any(math.isnan(float(v)) for v in t[:, 0]):
                # some rank's run was void: every rank keeps its own row (the void one's is rewritten by the repeated run)
                own = torch.from_numpy(t[rank].copy())
                table.zero_()
                if math.isnan(float(t[rank, 0])):
                    eng.finish()
                else:
                    table[rank].copy_(own)
                    eng._unchecked = False
                dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
                t = table.cpu().numpy()
            else:
                eng._unchecked = False
        finally:
            table.zero_()
    losses, gidx = t[:, 0], t[:, 1]
    cand = [(gidx[r], r) for r in range(world) if losses[r] == losses.min()]
    row = min(cand)[1]
    return int(t[row, 1]), float(t[row, 0]), torch.from_numpy(t[row, 2:].copy()).reshape(4, 4)


import threading

_PINNED = None
_TABLES = {}
_PINNED_LOCK = threading.Lock()
_TABLES_LOCK = threading.Lock()


def _pinned_row():
    global _PINNED
    if _PINNED is None:
        _PINNED = torch.empty((1, 18), dtype=torch.float32, pin_memory=True)
    return _PINNED


def merge_object_tables(table, group=None):
    """Multi-object jobs (bop.refine_frame): row i of `table` [n_obj, 18] is filled by the one rank that owns
    object i and zero elsewhere, so a single all_reduce(SUM) gives every rank the complete table."""
    import torch.distributed as dist

    if dist.is_initialized():
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return table

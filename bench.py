#!/usr/bin/env python
"""bench.py -- render+backward iterations/sec of the fused refinement engine on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...` (one rank per GPU over RCCL); started by a launcher (RANK / WORLD_SIZE in the
environment) it just runs its rank.

One "step" = one full optimiser iteration (pose -> matrices -> vertex transform -> binning/raster ->
shade + losses + analytic backward -> d loss/d(q,t) -> optimiser step) over the 64 pose hypotheses a GPU
owns, on BASELINE.json configs[1] ("cfg2": 20 480-triangle textured mesh, 640x480, rgb+mask loss),
synthetic inputs resident in HBM.  Multi-GPU = hypothesis sharding (weak scaling: 64 hypotheses per
GPU, global batch 64*N in the batch-mean factor) with ONE all_reduce for the global arg-min pose.
Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import gc
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: 8 TB/s HBM3E spec (6.3 TB/s achievable copy rate); 256 CUs x 4 SIMD-32 at 2.4 GHz, one wave64 VALU
# instruction issues over 2 cycles => 256 * 4 * 2.4e9 / 2 wave-instructions per second
HBM_PEAK_GBS = 8000.0
LINE_RATE_PEAK_G = 52.7   # random 64-byte lines per second, whole chip, in units of 1e9 (tools/ubench/gather_rate.hip)
STORE_LINE_PEAK_G = 104.2  # coalesced 16 B + 8 B per-lane record stores as 64-byte lines per second: 6.67 TB/s on a 257 MB footprint (tools/ubench/store_rate.hip,
                           # profiles/r4b_ubench_store_rate.jsonl; 33.5 G lines/s on cfg2's own 16 MB, where the launch is the limit)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # 1228.8 G wave-instructions / s


def algorithmic_bytes(V, T, HW, B):
    """SURVEY.md section 8(d), fp32/int32 'visibility-buffer model', bytes per LAUNCH (all B hypotheses).  Kept as
    `roofline.model_8d` for continuity: it counts full-frame G-buffer streams this engine never moves."""
    return {
        "step_kernel": (12.0 * V + 16.0 * V) * B + 212.0 * B + (16.0 * V + 12.0 * T + 16.0 * HW) * B,  # xfm + raster rows
        "shade_kernel": (32.0 * HW + 60.0 * V) * B + 40.0 * HW,
        "iteration": (104.0 * V + 12.0 * T + (48.0 + 40.0 / B) * HW) * B,
    }


def compulsory_bytes(V, T, HW, B, active_tiles, covered_px, textured, n_roles, shade_slices, uses):
    """Work-proportional HBM byte model of THIS engine, per launch (all B hypotheses): every stream the kernels must move
    once, at the 64-byte granularity of the memory-side requests, nothing for what stays in L2 between kernels being
    optimistic.  active_tiles = 16x16 tiles with any coverage summed over the hypotheses (engine status), covered_px =
    covered pixels summed over the hypotheses (coverage x HW x B).
      step    : partial rows read (every workgroup of a hypothesis reads them: L2 hits after the first), the meshlet tables
                (16 B per vertex slot ~ 1.2 V, 8 B per triangle; static, shared by the hypotheses), clip 16 B + snap 8 B per
                vertex and hypothesis WRITTEN (read back by the mask role only for silhouette triangles: not counted), zbuf
                atomics 8 B x 2 fragments per covered pixel (front + back faces) as 64-B sectors of the 4x4-pixel blocks
                => active_tiles x 256 x 8 B read-modify-write, zbuf re-arm 8 B x 256 per active tile written
      shade   : the tile-flag rows (NT bytes per hypothesis), zbuf 8 B x 256 per active tile (read once; both roles hit the
                same lines), triangle records 64 B x T (static table, shared), observed rgb + seg 24 B/px of the tiles one
                hypothesis touches (shared), texels: ONE 64-byte record per covered pixel (the engine's texq layout: the whole
                2x2 bilinear footprint in one sector; no reuse across hypotheses: 3 500 samples spread over a 268 MB table),
                partial rows written
    """
    tiles_one = active_tiles / max(B, 1)
    part = B * shade_slices * n_roles * 96.0
    tex = 64.0 * covered_px if (textured and (uses["rgb"] or uses["edge"])) else 0.0
    gt = tiles_one * 256 * (12.0 + (12.0 if uses["rgb"] else 0.0) + (4.0 if uses["depth"] else 0.0))
    out = {
        "step_kernel": part + 16.0 * 1.2 * V + 8.0 * T + (16.0 + 8.0) * V * B + 3 * active_tiles * 256 * 8.0,
        "shade_kernel": B * HW / 256.0 + active_tiles * 256 * 8.0 + (64.0 if textured else 80.0) * T + gt + tex + part,
    }
    out["iteration"] = sum(out.values())
    return out


def cpu_baseline(w, budget_s=12.0):
    """The oracle (CPU port of the same iteration, op by op like the reference) timed on this host's cores: whole iterations
    of the workload's full batch, one hypothesis per task on a thread pool over ALL host cores (the C calls and numpy's
    large-array ops release the GIL), until ~budget_s seconds have been spent; plus the same on ONE core for a 2-hypothesis
    sample, and BASELINE configs[0] (1 hypothesis, 160x120, 13 860 triangles) end to end on one core."""
    import concurrent.futures as cf
    import platform

    import numpy as np

    from oracle import oracle as orc  # cpu_baseline leg only

    npy = lambda t: None if t is None else t.detach().cpu().numpy()

    def oracle_of(w):
        kw = dict(uv=npy(w["uv"]), tex=npy(w["tex"])) if w["tex"] is not None else dict(vtx_color=npy(w["vtx_color"]))
        wts = {k: w["weights"].get(k) for k in ("rgb", "depth", "mask", "edge")}
        return orc.RenderOracle(npy(w["pos"]), npy(w["tri"]), npy(w["proj"]), w["H"], w["W"], {k: npy(v) for k, v in w["gt"].items()}, wts,
                                dtype=np.float32, cull_backfaces=False, **kw)  # (both faces: dr.rasterize's rule, as `value`)

    R = oracle_of(w)
    B = w["B"]
    params, lrm = npy(w["params0"]).copy(), npy(w["lr_mult"]).copy()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    model = platform.processor() or ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    # ---- one core, 2-hypothesis sample (the round-1 figure)
    R.loss_and_grad(params[:, :2], lrm[:2], global_B=B)  # warm caches
    n1, t0 = 0, time.perf_counter()
    while True:
        R.loss_and_grad(params[:, :2], lrm[:2], global_B=B)
        n1 += 1
        el1 = time.perf_counter() - t0
        if el1 > budget_s / 3 or n1 >= 200:
            break
    s_hyp_1core = el1 / (n1 * 2)
    # ---- all cores: whole B-hypothesis iterations
    workers = max(1, min(B, cores))
    chunks = [(i * B // workers, (i + 1) * B // workers) for i in range(workers)]
    task = lambda c: R.loss_and_grad(params[:, c[0]:c[1]], lrm[c[0]:c[1]], global_B=B)[2]
    with cf.ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(task, chunks))  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            list(ex.map(task, chunks))
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 400:
                break
    out = {
        "value": n / el, "unit": "iters/s", "cores": workers, "kind": "port", "host_cores": cores, "cpu_model": model,
        "sample": f"{n} whole iterations of the same workload ({B} hypotheses each, one hypothesis-chunk per thread, {workers} threads, {el:.1f} s)",
        "one_core": {"value": 1.0 / (s_hyp_1core * B), "cores": 1,
                     "sample": f"{n1} iterations x 2 hypotheses ({el1:.1f} s), scaled to {B} hypotheses/iter",
                     "s_per_hypothesis_iteration": s_hyp_1core},
    }
    # ---- BASELINE configs[0]: 1 hypothesis, 160x120, 13 860 triangles, 61 iterations end to end, one core
    try:
        from diffdope_amd import workloads as wl

        w1 = wl.build("cfg1", w["pos"].device)
        R1 = oracle_of(w1)
        lrs = wl.bench_lr_schedule(61, "sgd")
        t0 = time.perf_counter()
        R1.optimise(npy(w1["params0"]), npy(w1["lr_mult"]), lrs)
        el = time.perf_counter() - t0
        out["cfg1_it_s"] = 61 / el
        out["cfg1"] = {"value": 61 / el, "unit": "iters/s", "cores": 1, "sample": f"61 SGD iterations end to end, 1 hypothesis, 160x120, T={w1['T']} ({el:.2f} s)"}
    except Exception as e:  # the headline baseline above stands on its own
        out["cfg1"] = {"error": repr(e)}
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def csrc_sha16():
    """Hash of the kernel sources: counters from a PMC pass of other sources are stale (profiles/summarize_sq.py records it)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "diffdope_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "diffdope_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _flush_c_stdio():
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _latest_profile(suffix):
    d = os.path.join(ROOT, "profiles")
    try:
        files = sorted(f for f in os.listdir(d) if f.endswith(suffix))
        return (json.load(open(os.path.join(d, files[-1]))), "profiles/" + files[-1]) if files else (None, None)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--optimizer", default="adam", choices=["adam", "sgd"],
                    help="adam = north_star's outer loop (default); sgd = the reference's optimiser (diffdope.py:1642-1644); the default run reports both")
    ap.add_argument("--distance", type=float, default=None, help="camera distance in scene units (default: the config's, 7.5 = the example's 747 mm); "
                    "smaller = larger object in the frame -- for coverage-sensitivity sweeps, not the headline line")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: a fixed job of this many hypotheses sharded over the ranks (BASELINE "
                    "configs[3]: --config cfg4 --global-batch 512 --gpus 8); default 0 = weak scaling, the config's batch on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true", help="skip the untimed 200-iteration convergence leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (reference SGD, close-up at distance 3.75, repeats)")
    ap.add_argument("--repeats", type=int, default=5, help="extra timed windows of --steps iterations after the contract's one (median reported beside it)")
    ap.add_argument("--closing-barrier", action="store_true",
                    help="N > 1: a torch.distributed.barrier() behind the job's all_reduce inside the timed window (default: the all_reduce "
                         "itself is the closing barrier, followed by torch.cuda.synchronize())")
    ap.add_argument("--settle-ms", type=float, default=15.0,
                    help="before the W warm-up iterations: run the engine for this long, untimed, then restore the initial poses and a fresh optimiser "
                         "state (0 = off).  The board's power management needs ~7 ms of the engine's own load to settle: an identical 20-iteration "
                         "window takes 44.6 us per iteration 13 ms into a process and 40.0 us after 7 ms more of the same work, and 43.6 us again "
                         "after 50 ms of idling (tools/ramp_probe.py, profiles/r4c_ramp_probe.log); other device work does not substitute")
    ap.add_argument("--graph", type=int, nargs="?", const=1, default=0, help="replay captured hipGraphs of K iterations (default 1; measured 9 %% slower than plain stream launches at K=1, equal at K=20)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain form `python bench.py --gpus N`: become the launcher (one rank per GPU, RCCL over xGMI)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # DDX_BENCH_SHARE_GPU (tests only): every rank on device 0 with the gloo backend -- RCCL refuses two ranks on one GPU, and
    # the 1-GPU test box is where the world-size-2 code path of this file can be exercised on the device at all
    share = bool(os.environ.get("DDX_BENCH_SHARE_GPU"))
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or bool(os.environ.get("DDX_FORCE_DIST"))  # DDX_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))  # (only reached without a launcher: DDX_FORCE_DIST on one rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from diffdope_amd import dist as ddist
    from diffdope_amd import workloads as wl

    strong = args.global_batch > 0
    if strong:  # a fixed job sharded over the ranks (rank r owns a contiguous slice)
        B_job = args.global_batch
        lo, hi = ddist.shard_range(B_job, rank, world)
        Bl = hi - lo
        if Bl < 1:
            raise SystemExit(f"--global-batch {B_job} leaves rank {rank} of {world} without a hypothesis")
    else:       # weak scaling: the config's batch on every rank
        Bl = wl.CONFIGS[args.config]["B"]
        B_job, lo = Bl * world, rank * Bl
    n_it = args.warmup + args.steps
    dist_info = {"backend": torch.distributed.get_backend() if use_dist else None, "world_size": world, "device_ids": [dev_index]}
    if use_dist:
        ids = torch.zeros(world, dtype=torch.int32, device=dev)
        ids[rank] = dev_index + 1
        torch.distributed.all_reduce(ids)  # (also the first collective: RCCL channel set-up happens here, outside every timed window)
        dist_info["device_ids"] = [int(x) - 1 for x in ids.cpu().tolist()]
        rng = torch.zeros((world, 2), dtype=torch.int32, device=dev)  # which hypotheses of the job each rank owns: [lo, hi) per rank
        rng[rank, 0], rng[rank, 1] = lo, lo + Bl
        torch.distributed.all_reduce(rng)
        dist_info["shard_ranges"] = [[int(a), int(b)] for a, b in rng.cpu().tolist()]
        _flush_c_stdio()  # (RCCL's banner, if any, now and not at exit)
        dist_info["hsa_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(w, optimizer, repeats=0, lo_w=None, global_w=None, steps=None, warmup=None, settle=True, **eng_kw):
        """The contract's measurement on workload w: W warm-up iterations, then EXACTLY K iterations + the arg-min selection
        (incl. the one all_reduce) between barrier + synchronize, max over ranks.  Optionally `repeats` more windows of the
        same engine, each restarted from the initial poses with a fresh optimiser state and the same W warm-up iterations
        (the same rows of the schedule, the same work), for a median."""
        # (the interpreter's cyclic collector stays off from here to the end of the measurement, as in timeit: a generation-2
        # collection -- 1.3 ms with torch loaded -- landed in the first window of some workloads and not of others, by allocation
        # count; and collecting right before a window leaves the GPU idle for milliseconds, after which the window itself runs
        # 8-10 us per iteration slower.  tools/first_window.py)
        gc.disable()
        try:
            return _timed(w, optimizer, repeats, lo_w, global_w, args.steps if steps is None else steps, args.warmup if warmup is None else warmup,
                          settle, eng_kw)
        finally:
            gc.enable()

    def _timed(w, optimizer, repeats, lo_w, global_w, K, W_, settle, eng_kw):
        n_it = W_ + K
        lrs = wl.bench_lr_schedule(n_it, optimizer)
        lo_w = lo if lo_w is None else lo_w
        eng, params = wl.engine_for(w, lrs, optimizer=optimizer, global_batch=global_w or w["global_B"], **eng_kw)
        used = [i for i, k in enumerate(("rgb", "depth", "mask", "edge")) if w["weights"].get(k) is not None]
        row_mask = sum(1 << i for i in used)
        # the one collective of the job: global arg-min hypothesis + its pose (diffdope.py:1488-1513,1618-1632);
        # local selection by one device kernel, one all_reduce of the [world,18] table, one host synchronisation
        # (round 4: the local selection rides on the run's last kernel -- ddx_engine_run_select -- instead of its own launch)
        def window():
            barrier()
            t0 = time.perf_counter()
            best = ddist.run_and_select(eng, K, lo=lo_w, use_graph=args.graph)
            # the closing barrier: with more than one rank the job's own all_reduce IS one -- no rank holds the global result
            # before every rank has finished its K iterations and contributed its row -- and run_and_select has already copied
            # that result to the host; a second collective behind it (torch.distributed.barrier() = another all_reduce, 24-27 us
            # with one RCCL rank) would only time itself.  --closing-barrier adds it back.
            if use_dist and not args.closing_barrier:
                torch.cuda.synchronize()
            else:
                barrier()
            el = time.perf_counter() - t0
            if use_dist:
                tmax = torch.tensor([el], dtype=torch.float64, device=dev)
                torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
                el = float(tmax.item())
            return el, best

        cold = None
        if args.settle_ms > 0 and settle:
            # what the same W + K read BEFORE the device has settled (reported beside the contract's window, never as `value`)
            if W_ > 0:
                ddist.run_and_select(eng, W_, lo=lo_w, use_graph=args.graph)
            cold = window()[0]
            # steady state of the device before anything is timed (--settle-ms): the same engine, the same iterations; afterwards
            # the engine is put back exactly as the repeat windows below put it back
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < args.settle_ms * 1e-3:
                eng.rewind(0)
                eng.run(n_it)
                torch.cuda.synchronize()
            eng.new_observation(params=w["params0"])
        if W_ > 0:
            ddist.run_and_select(eng, W_, lo=lo_w, use_graph=args.graph)  # (warms the selection path too: pinned row, RCCL channels)
        elapsed, best = window()
        if os.environ.get("DDX_BENCH_CHECK_SELECT"):  # (tests: the fused selection against the stand-alone kernel on the same rows)
            ref = ddist.global_argmin_fused(eng.loss_log[n_it - 1], row_mask, eng.mtx_log[n_it - 1], lo=lo_w)
            assert ref[0] == best[0] and ref[1] == best[1] and torch.equal(ref[2], best[2]), (ref, best)
        st = eng.check()
        final = params.clone()
        per_hyp = eng.loss_log[n_it - 1][used].mean(0).clone()
        extra = []
        for _ in range(repeats):
            eng.new_observation(params=w["params0"])  # initial poses, zero moments, iteration 0
            if W_ > 0:
                eng.run(W_, use_graph=args.graph)
                eng.finish()  # (synchronises and validates the warm-up here, not inside the window: ddx_engine_run_check)
            extra.append(window()[0])
        return dict(elapsed=elapsed, best=best, status=st, params=final, per_hyp=per_hyp, lrs=lrs, repeats=extra, eng=eng, cold=cold)

    w = wl.build(args.config, dev, B=Bl, global_lo=lo, global_B=B_job, distance=args.distance)
    extras_on = not args.no_extras and world == 1
    r = timed(w, args.optimizer, repeats=args.repeats if extras_on else 0)
    elapsed = r["elapsed"]
    # did this rank's engine find a stream that runs BESIDE the caller's (two half-batch chains) or fall back to one chain?  (-1: the
    # engine never asked -- its launches fill the chip, or the run was too short)
    two = torch.full((world,), 0, dtype=torch.int32, device=dev)
    two[rank] = r["eng"].two_chains + 2
    if use_dist:
        torch.distributed.all_reduce(two)
    dist_info["two_chains"] = [int(x) - 2 for x in two.cpu().tolist()]
    gidx, gloss, gpose = r["best"]

    if rank == 0:
        rot, tr = wl.pose_errors(r["params"], w["q_gt"], w["t_gt"])
        add = wl.add_error(r["params"], w["pos"], w["q_gt"], w["t_gt"])
        lbest = int(np.argmin(r["per_hyp"].cpu().numpy()))
        V, T, HW = w["V"], w["T"], w["H"] * w["W"]
        alg = algorithmic_bytes(V, T, HW, Bl)
        # per-kernel launch durations, live, HIP events on the launch stream (a second engine: profiling mutates poses)
        # (one chain of full-batch launches: what a launch duration means.  The timed region above runs 16+ iterations as two
        # half-batch chains on two streams -- their kernels overlap each other --; the same window as ONE chain is timed here,
        # untimed for the contract, and the per-kernel shares below are shares of THAT iteration)
        eng2, _ = wl.engine_for(w, r["lrs"], optimizer=args.optimizer, global_batch=B_job, single_stream=True)
        wu2 = min(args.warmup, n_it - 1)
        eng2.run(wu2)
        torch.cuda.synchronize()
        kms_ev = eng2.profile(it0=wu2, iters=max(1, min(20, n_it - args.warmup)))
        one_chain = []
        for _ in range(3):
            eng2.new_observation(params=w["params0"])
            eng2.run(wu2)
            torch.cuda.synchronize()
            t1c = time.perf_counter()
            eng2.run(n_it - wu2)
            torch.cuda.synchronize()
            one_chain.append((time.perf_counter() - t1c) / (n_it - wu2) * 1e3)
        ms_one_chain = sorted(one_chain)[1]
        ms_per_step = elapsed / args.steps * 1e3
        # An event between two kernels of the stream costs ~3-4 us per kernel (the launches are no longer back to back), so
        # the event-bracketed durations sum to more than the iteration itself.  The one-chain window IS the kernels back to
        # back (rocprofv3: they sum to the iteration within 1 us, profiles/), so each kernel's share of that iteration is its
        # event-measured share: kernel_ms = kernel_ms_events * ms_one_chain / sum(kernel_ms_events).
        per_it = [k for k in kms_ev if k != "finish_kernel"]  # (finish_kernel runs once per run, not per iteration)
        ev_scale = min(1.0, ms_one_chain / max(sum(kms_ev[k] for k in per_it), 1e-9))
        kms = {k: v * (ev_scale if k in per_it else 1.0) for k, v in kms_ev.items()}
        groups = {"step_stage": kms["step_kernel"] + kms["big_pass_kernel"], "shade_stage": kms["shade_kernel"] + kms["edge_kernel"]}
        dom = max(("shade_kernel", "step_kernel"), key=lambda k: kms[k])
        dom_s = kms[dom] * 1e-3
        # ---- counters of the committed rocprofv3 --pmc passes for THIS workload (profiles/summarize_sq.py; separate passes,
        # kernel trace only): wave-level VALU instructions and HBM-side bytes per launch.  null if no pass matches.
        wkey = f"{args.config}_d{w['distance']:g}"
        pmc, pmc_src = _latest_profile(f"_pmc_sq_{wkey}.json")
        kp = None
        if pmc:
            # (the full-batch launches: the half-batch launches of a two-stream run carry a last template argument `true`)
            kp = next((v for k, v in pmc["kernels"].items() if dom in k and not k.rstrip().endswith("true>") and v.get("launches", 0) > 4), None)
        # counters measured on other kernel sources than the ones running now are not this build's: reported as stale, never as frac
        stale = bool(pmc) and pmc.get("csrc_sha16") != csrc_sha16()
        valu_insts = kp.get("SQ_INSTS_VALU") if kp else None
        valu_active_q = kp.get("SQ_ACTIVE_INST_VALU") if kp else None                # quad-cycles the waves spent executing VALU instructions
        traffic = kp.get("hbm_bytes_per_launch") if kp else None                      # FETCH_SIZE + WRITE_SIZE as reported
        traffic_x2 = kp.get("hbm_bytes_per_launch_fetch_doubled") if kp else None     # with the guide's x2 on FETCH_SIZE
        fetch_cal = (pmc or {}).get("fetch_size_calibration")                         # tools/ubench/gather64.hip under --pmc FETCH_SIZE
        uses = {k: w["weights"].get(k) is not None for k in ("rgb", "depth", "mask", "edge")}
        n_roles = int(uses["rgb"] or uses["depth"] or uses["edge"]) + int(uses["mask"])
        comp = compulsory_bytes(V, T, HW, Bl, r["status"]["active_tiles"], w["coverage"] * HW * Bl, w["tex"] is not None,
                                max(n_roles, 1), r["eng"].slices[0], uses)
        model_bytes = alg[dom]
        # ---- the roofline block (round 4: says what the measurements support).  Top level = the HBM byte view of the dominant
        # kernel: achieved = the work-proportional COMPULSORY bytes of this engine per launch / the live launch duration, against
        # 8 TB/s; `traffic` = what the PMC passes counted on the memory side for that launch, next to it their ratio.  The
        # SURVEY 8(d) model (full-frame G-buffer streams this engine never moves) stays as model_8d for continuity.  `valu` is the
        # issue view (wave-level VALU instructions per launch / duration against 1228.8 G/s).  Neither is near its ceiling: these
        # launches are bound by dependent memory levels and kernel boundaries (what_binds).
        valu_ach = (valu_insts / dom_s / 1e9) if (valu_insts and not stale) else None
        hbm_ach = comp[dom] / dom_s / 1e9
        sh_s = kms["shade_kernel"] * 1e-3
        st_s = kms["step_kernel"] * 1e-3
        texel_lines = (w["coverage"] * HW * Bl) if (w["tex"] is not None and (uses["rgb"] or uses["edge"])) else 0.0
        store_lines = 24.0 * V * Bl / 64.0
        roof = {
            "kernel": dom, "bound": "hbm", "achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS,
            "compulsory_bytes_per_launch": comp[dom],
            "traffic": None if stale else traffic, "traffic_over_compulsory": (traffic / comp[dom]) if (traffic and not stale) else None,
            "traffic_frac_of_peak": (traffic / dom_s / 1e9 / HBM_PEAK_GBS) if (traffic and not stale) else None,
            "traffic_fetch_doubled": None if stale else traffic_x2, "fetch_size_calibration": fetch_cal,
            "avg_launch_ms": kms[dom], "counters_source": pmc_src, "counters_stale": stale, "csrc_sha16": csrc_sha16(),
            "valu": {"achieved": valu_ach, "peak": VALU_PEAK_GINST, "unit": "Ginst/s", "frac": (valu_ach / VALU_PEAK_GINST) if valu_ach else None,
                     "valu_wave_insts_per_launch": valu_insts,
                     "salu_wave_insts_per_launch": (kp.get("SQ_INSTS_SALU") if kp else None),  # (round 4: 7.23 M VALU + 3.59 M SALU on cfg2)
                     "wait_frac_of_wave_cycles": ((kp["SQ_WAIT_ANY"] / kp["SQ_WAVE_CYCLES"]) if (kp and kp.get("SQ_WAVE_CYCLES") and not stale) else None),
                     # the same pipe seen by SQ_ACTIVE_INST_VALU (quad-cycles waves spent executing VALU instructions, summed over the
                     # chip) x 4 / (1024 SIMDs x 2.4 GHz x duration): ~4 cycles per instruction on these integer-heavy kernels, so
                     # about twice `frac`; the larger of the two is the honest "how busy is the VALU" (0.72 at saturation)
                     "busy_frac": (valu_active_q * 4.0 / (1024 * 2.4e9 * dom_s)) if (valu_active_q and not stale) else None},
            "iteration": {"compulsory_bytes": comp["iteration"], "achieved_GBps": comp["iteration"] / (ms_per_step * 1e-3) / 1e9,
                          "frac": comp["iteration"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          # the timed window's iteration, and the same window as one chain of full-batch launches (median of 3, no
                          # selection): the launch durations above are shares of the latter
                          "ms": ms_per_step, "ms_one_chain": ms_one_chain},
            # the request view, per kind of line: lines NO lane shares (one 64-byte texel record per covered pixel: the colour role's
            # gather, the one load of the shading launch whose removal shortens it) against the measured rate of random 64-byte
            # records (tools/ubench/gather_rate.hip); COALESCED store lines of step_kernel (16 B clip + 8 B snap per vertex, whole
            # lines since round 4) against the measured rate of such stores (tools/ubench/store_rate.hip).  No line-rate claim is
            # made for step_kernel as a whole: leaving its stores out does not shorten the launch.
            "lines": {"shade_texel_record_lines_per_launch": texel_lines, "gather_ceiling_Glines_s": LINE_RATE_PEAK_G,
                      "shade_texel_gather_frac": (texel_lines / sh_s / 1e9 / LINE_RATE_PEAK_G) if sh_s > 0 else None,
                      "step_coalesced_store_lines_per_launch": store_lines, "store_ceiling_Glines_s": STORE_LINE_PEAK_G,
                      "step_store_frac": (store_lines / st_s / 1e9 / STORE_LINE_PEAK_G) if st_s > 0 else None,
                      "sources": "tools/ubench/gather_rate.hip, tools/ubench/store_rate.hip (profiles/r3c_ubench_gather_rate.jsonl, profiles/r4b_ubench_store_rate.jsonl)"},
            "what_binds": "at 64 hypotheses no unit is saturated (profiles/r6a_bottleneck.md: every busy fraction 0.3-0.65): step_kernel is a ~9-us "
                          "floor (launch start, partial rows, the optimiser head's one-wave tail) plus four stages -- transform, triangle predicates, "
                          "coverage, fragments -- of 3.5-5 us each whose cost follows their instruction count (ablation builds, "
                          "profiles/r6e_ablate_table.md), and ends with its slowest workgroup (median 15.7, worst 20.4 us); a kernel boundary costs "
                          "~2.4 us; at 512 hypotheses per GPU the same kernels run at 27 us per 64 hypotheses with the VALU 0.66 and TD / TCP 0.76 busy",
            "model_8d": {"algorithmic_bytes_per_launch": model_bytes, "achieved_GBps": model_bytes / dom_s / 1e9,
                         "ratio_to_hbm_peak": model_bytes / dom_s / 1e9 / HBM_PEAK_GBS,
                         "iteration_ratio_to_hbm_peak": alg["iteration"] * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8(d) visibility-buffer model (full-frame G-buffer streams); this engine touches active tiles "
                                 "only, so the ratio exceeds 1 -- continuity with round 1, not a roofline"},
            "note": "frac = compulsory HBM bytes of the dominant kernel / its live HIP-event launch duration / 8 TB/s; traffic = FETCH_SIZE + WRITE_SIZE "
                    "of the committed PMC passes for this build (null when the kernel sources changed since); see DESIGN.md section 6",
        }
        tex_hw = (int(w["tex"].shape[0]), int(w["tex"].shape[1])) if w["tex"] is not None else (0, 0)
        job_iters = args.steps / elapsed  # iterations of the whole job per second
        out = {
            "metric": f"render+backward iters/sec at {w['W']}x{w['H']}, {Bl if not strong else B_job} hypotheses per iteration",
            # weak scaling: every rank runs its own 64-hypothesis iterations (value = N x the per-rank rate); strong scaling
            # (--global-batch): one iteration covers the whole fixed job
            "value": job_iters if strong else world * job_iters,
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # (kept under 128 characters: the driver's parser cuts the string there)
            "config": {"workload": f"{args.config}: blob mesh T={T} V={V}, tex {tex_hw[0]}^2, {w['W']}x{w['H']}, {Bl} hyps/GPU, "
                                   f"{'+'.join(sorted(w['weights']))}, {args.optimizer}, both faces, cov {100 * w['coverage']:.2f}%",
                       "rasteriser_rule": "both faces of every triangle drawn, as dr.rasterize (diffdope.py:198-200); also.cfg2_cull = deviation D5",
                       "hypotheses_per_gpu": Bl, "global_hypotheses": B_job, "parallelism": f"hyp-shard x{world}",
                       "hipgraph": bool(args.graph), "settle_ms": args.settle_ms,
                       "closing_barrier": ("torch.distributed.barrier()" if args.closing_barrier else "the job's all_reduce") if use_dist else "synchronize (one rank)"},
            "dist": dist_info,
            "hypothesis_iters_per_s": B_job * job_iters,
            "roofline": roof,
            "kernel_ms": kms, "kernel_ms_events": kms_ev, "stage_ms": groups,
            # engine memory that scales with the texture: one 64-byte footprint record per texel (4x the [Th,Tw,3] fp32 texture)
            "texq_bytes": tex_hw[0] * tex_hw[1] * 64, "engine_scratch_bytes": int(r["eng"].scratch.numel()),
            "final_pose": {"what": f"SNAPSHOT after the {n_it} iterations of this run (warm-up + timed), not a converged result when that is few: see `convergence`",
                           "argmin_global_index": gidx, "argmin_loss": gloss,
                           "rot_err_rad_best": float(rot[lbest]), "trans_err_m_best": float(tr[lbest]), "add_m_best": float(add[lbest]),
                           "rot_err_rad_median": float(np.median(rot)), "trans_err_m_median": float(np.median(tr))},
            "engine_status": r["status"],
        }
        if not args.no_convergence:
            # the second half of the metric ("final ADD err"): the full schedule on the SAME workload, outside every timed window
            n_cv = 200
            lr_cv = wl.bench_lr_schedule(n_cv, args.optimizer)
            eng_cv, p_cv = wl.engine_for(w, lr_cv, optimizer=args.optimizer, global_batch=B_job)
            eng_cv.run(n_cv)
            eng_cv.finish()
            used_cv = [i for i, k in enumerate(("rgb", "depth", "mask", "edge")) if w["weights"].get(k) is not None]
            bcv = int(torch.argmin(eng_cv.loss_log[n_cv - 1][used_cv].mean(0)))
            rot_c, tr_c = wl.pose_errors(p_cv, w["q_gt"], w["t_gt"])
            add_c = wl.add_error(p_cv, w["pos"], w["q_gt"], w["t_gt"])
            out["convergence"] = {"iterations": n_cv, "optimizer": args.optimizer, "argmin_local_index": bcv,
                                  "rot_err_rad_best": float(rot_c[bcv]), "trans_err_m_best": float(tr_c[bcv]), "add_m_best": float(add_c[bcv]),
                                  "rot_err_rad_median": float(np.median(rot_c)), "trans_err_m_median": float(np.median(tr_c)),
                                  "within_north_star_tolerance": bool(rot_c[bcv] < 1e-3 and tr_c[bcv] < 1e-3),
                                  "what": "same workload and engine, full 200-iteration schedule, untimed; errors against the generating pose"}
        if r["cold"] is not None:
            # the reading WITHOUT the settle phase, at top level beside `value` (ADVICE r4 / VERDICT r4 item 4): the contract's W + K as the
            # first thing the engine does in the process -- the number to compare with rounds 1-3, which had no settle phase
            out["value_no_settle"] = (job_iters if strong else world * job_iters) * elapsed / r["cold"]
            out["cold_window"] = {"ms_per_step": r["cold"] / args.steps * 1e3, "iters_per_s": (job_iters if strong else world * job_iters) * elapsed / r["cold"],
                                  "what": "the same W warm-up + K timed iterations + selection as the first thing the engine does in this process, "
                                          "before the --settle-ms phase (the board's power management still ramping: DESIGN.md section 6)"}
        if r["repeats"]:
            allw = sorted([elapsed] + r["repeats"])
            med = allw[len(allw) // 2]
            out["repeat_windows"] = {"n": len(allw), "ms_per_step_median": med / args.steps * 1e3, "iters_per_s_median": args.steps / med,
                                     "ms_per_step_all": [x / args.steps * 1e3 for x in [elapsed] + r["repeats"]]}
        if extras_on:
            # secondary lines on the same GPU (never `value`): the reference's own optimiser, and the close-up regime where
            # scatter and shade are throughput (VALU) bound instead of latency bound
            other = "sgd" if args.optimizer == "adam" else "adam"
            r2 = timed(w, other)
            out["also"] = {f"{args.config}_{other}": {"iters_per_s": args.steps / r2["elapsed"], "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                                                      "what": ("the reference's SGD" if other == "sgd" else "Adam") + ", same workload"}}
            if args.distance is None and args.config == "cfg2":
                # `value` follows the reference's rasteriser rule: dr.rasterize (diffdope.py:198-200) draws BOTH faces of every triangle
                # (workloads.engine_for: cull_backfaces=False).  Deviation D5 as an option: the back faces of a closed mesh inside the
                # view volume skipped (same pixels in exact arithmetic; tools/cull_sweep.py counts how often float arithmetic differs).
                # Timed exactly like `value`.
                rn = timed(w, args.optimizer, cull_backfaces=True)
                out["also"]["cfg2_cull"] = {"iters_per_s": args.steps / rn["elapsed"], "ms_per_step": rn["elapsed"] / args.steps * 1e3,
                                            "what": "RefineEngine(cull_backfaces=True), deviation D5: back faces of the closed mesh skipped; same window as `value`"}
                # BASELINE.md's own form of the metric: 200 iterations after 20 warm-up iterations, same workload and engine settings
                rc = timed(w, args.optimizer, steps=200, warmup=20)
                out["also"]["cfg2_contract200"] = {"iters_per_s": 200 / rc["elapsed"], "ms_per_step": rc["elapsed"] / 200 * 1e3, "steps": 200, "warmup": 20,
                                                   "what": "BASELINE.md's window: 200 timed iterations after 20 warm-up (+ selection), behind the settle phase",
                                                   "no_settle_iters_per_s": (200 / rc["cold"]) if rc["cold"] else None}
                del rn, rc
                wc = wl.build(args.config, dev, B=Bl, distance=3.75)
                r3 = timed(wc, args.optimizer)
                out["also"]["cfg2_d3.75"] = {"iters_per_s": args.steps / r3["elapsed"], "ms_per_step": r3["elapsed"] / args.steps * 1e3,
                                             "coverage_pct": 100 * wc["coverage"], "what": "same mesh and losses at half the distance (object 4x the area)"}
                for name, what in (("cfg50k64", "north_star target sentence: 64 hypotheses of a 51 200-triangle textured mesh at 640x480, rgb+mask"),
                                   ("cfg3", "BASELINE configs[2]: 128 hypotheses, 51 200 triangles, rgb+depth+edge (edge = this build's extension)")):
                    wx = wl.build(name, dev)
                    rx = timed(wx, args.optimizer, lo_w=0)
                    out["also"][name] = {"iters_per_s": args.steps / rx["elapsed"], "ms_per_step": rx["elapsed"] / args.steps * 1e3,
                                         "hypotheses": wx["B"], "hypothesis_iters_per_s": wx["B"] * args.steps / rx["elapsed"], "what": what}
                    del wx, rx
        if not args.no_cpu_baseline and world == 1:  # (the CPU leg is timed on rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(w)
    # rank 0's JSON line is the LAST thing on the job's stdout: whatever the native libraries still hold in C stdio buffers (with
    # NCCL_DEBUG=VERSION, as on the GPU boxes, RCCL's version banner would otherwise be flushed at process exit, behind the line)
    # goes out first on every rank, and the line is printed behind the last collective
    _flush_c_stdio()
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        _flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

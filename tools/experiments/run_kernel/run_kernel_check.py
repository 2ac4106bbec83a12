"""The one-launch form of a run (engine.hip run_kernel) against the launch form: same bits, and what it costs.

    python tools/run_kernel_check.py [--configs cfg2,cfg4,cfg50k64] [--iters 20] [--time-iters 200] [--batch B]

For every workload: two engines on the same inputs, one with one_launch_run=True; n iterations (Adam); parameters, loss log,
pose log, status words and the row of ddx_engine_run_select compared bit for bit.  Then steady-state windows of both forms."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl  # noqa: E402


def run_once(w, lrs, n, optimizer, **kw):
    eng, p = wl.engine_for(w, lrs, optimizer=optimizer, **kw)
    out = torch.zeros(18, dtype=torch.float32, device=p.device)
    eng.run_select(out, n=n)
    eng.finish()
    st = eng.status()
    return eng, dict(params=p.clone(), loss=eng.loss_log[:n].clone(), mtx=eng.mtx_log[:n].clone(), sel=out.clone(), st=st, form=eng.run_form)


def window(eng, p0, n, reps=5):
    """median us per iteration of `reps` windows of n iterations (rewound, same parameters)"""
    ts = []
    out = torch.zeros(18, dtype=torch.float32, device=p0.device)
    for _ in range(reps + 2):
        eng.params.copy_(p0)
        eng.rewind(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_select(out, n=n)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / n)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,cfg4,cfg50k64")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--time-iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--optimizer", default="adam")
    ap.add_argument("--no-time", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ok_all = True
    for name in args.configs.split(","):
        w = wl.build(name, dev, B=args.batch)
        n = args.iters
        lrs = wl.bench_lr_schedule(max(n, args.time_iters), args.optimizer)
        e_l, a = run_once(w, lrs, n, args.optimizer)
        e_r, b = run_once(w, lrs, n, args.optimizer, one_launch_run=True)
        same = {k: bool(torch.equal(a[k], b[k])) for k in ("params", "loss", "mtx", "sel")}
        st_keys = ("big_triangles", "active_tiles", "it", "outside_view_volume", "flags")
        same["status"] = all(a["st"][k] == b["st"][k] for k in st_keys)
        rec = dict(config=name, B=w["B"], iters=n, form_launches=a["form"], form_run=b["form"], same=same, repeated=b["st"]["repeated_runs"],
                   status_launches={k: a["st"][k] for k in st_keys}, status_run={k: b["st"][k] for k in st_keys})
        if not all(same.values()):
            ok_all = False
            d = (a["params"] - b["params"]).abs().max().item()
            rec["max_abs_param_diff"] = d
            rec["sel_l"] = a["sel"].cpu().tolist()[:2]
            rec["sel_r"] = b["sel"].cpu().tolist()[:2]
        if not args.no_time:
            p0 = w["params0"].clone()
            for tn in sorted({20, args.time_iters}):
                m_l, b_l = window(e_l, p0, tn)
                m_r, b_r = window(e_r, p0, tn)
                rec[f"us_per_iter_{tn}"] = dict(launches=round(m_l, 2), run_kernel=round(m_r, 2), best_launches=round(b_l, 2), best_run_kernel=round(b_r, 2),
                                                two_chains=e_l.two_chains)
        print(json.dumps(rec), flush=True)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())

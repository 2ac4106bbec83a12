#!/usr/bin/env python
"""Summarise tools/ablate_step.sh:   python tools/ablate_table.py <tag> [B ...]   (the driver's batch sizes, in its order)
Per build and batch size: step_kernel's un-profiled launch duration and its SQ counters per launch and per hypothesis; the
differences between consecutive builds are the stages' shares."""
import collections
import csv
import glob
import os
import statistics
import sys

tag = sys.argv[1]
BS = [int(a) for a in sys.argv[2:]] or [64, 512]
N_IT = int(os.environ.get("BN_ITERS", "24"))
root = f"gpurun_out/ab_{tag}"
KERNELS = {"step": "step_kernel", "shade": "shade_kernel"}


def rows_by_batch(path, kname):
    rows = [r for r in csv.DictReader(open(path)) if kname in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    rank = {d: i for i, d in enumerate(ids)}
    out = collections.defaultdict(list)
    for r in rows:
        i = rank[int(r["Dispatch_Id"])]
        if i // N_IT < len(BS) and i % N_IT >= 2:
            out[BS[i // N_IT]].append(r)
    return out


libs = [v for v in ("product", "a1", "a2", "a3", "a4") if os.path.isdir(f"{root}/{v}")] + sorted(
    os.path.basename(p) for p in glob.glob(f"{root}/*") if os.path.isdir(p) and os.path.basename(p) not in ("product", "a1", "a2", "a3", "a4"))
WHAT = {"product": "everything", "a1": "- fragments (depth, atomicMin)", "a2": "- coverage mask, tile flags", "a3": "- triangle predicates, compaction",
        "a4": "- meshlet loop (transform)"}
for kk, kname in KERNELS.items():
    print(f"\n## {kname}\n")
    for B in BS:
        print(f"### {B} hypotheses\n")
        print("| build | duration (us) | VALU / launch | VALU / hypothesis | SALU / hyp | LDS / hyp | VMEM_WR / hyp | VALU busy | wait |")
        print("|---|---|---|---|---|---|---|---|---|")
        prev = None
        for v in libs:
            d = []
            for p in glob.glob(f"{root}/{v}/trace/**/*kernel_trace.csv", recursive=True):
                d += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows_by_batch(p, kname).get(B, [])]
            c = collections.defaultdict(list)
            for p in glob.glob(f"{root}/{v}/pmc/**/*counter_collection.csv", recursive=True):
                for r in rows_by_batch(p, kname).get(B, []):
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if not d or not c:
                continue
            m = {k: statistics.median(x) for k, x in c.items()}
            du = statistics.mean(d) / 1e3
            busy = m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e3 * du) if "SQ_ACTIVE_INST_VALU" in m else float("nan")
            wait = m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
            print(f"| {v}: {WHAT.get(v, '')} | {du:.1f} | {m['SQ_INSTS_VALU']:,.0f} | {m['SQ_INSTS_VALU'] / B:,.0f} | {m['SQ_INSTS_SALU'] / B:,.0f} | "
                  f"{m.get('SQ_INSTS_LDS', 0) / B:,.0f} | {m.get('SQ_INSTS_VMEM_WR', 0) / B:,.0f} | {busy:.2f} | {wait:.2f} |")
        print()

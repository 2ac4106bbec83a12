#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE needs
3 TCC slots, WRITE_SIZE 2) into per-kernel HBM-side bytes per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_<tag>_FETCH_SIZE -o pmc --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_<tag>_WRITE_SIZE -o pmc --output-format csv -- python bench.py ... (same)
    python profiles/summarize_pmc.py <tag> > profiles/<tag>_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (TCC_EA request counts x 64 B / 1024).  Calibration notes of the guide: on
gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x; the engine's reads are gathers and small
tiles, for which no calibration exists, so the raw value is reported (`fetch_doubled` gives the upper reading).
"""
import collections
import csv
import json
import statistics
import sys

tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/pmc_{tag}_{c}/pmc_counter_collection.csv")):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(s in k for s in ("scatter", "compact", "shade", "update", "pose_xfm")):
            out.setdefault(k, {})[c + "_KiB_median_per_launch"] = statistics.median(v)
            out[k]["launches"] = len(v)
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_KiB_median_per_launch", 0.0), v.get("WRITE_SIZE_KiB_median_per_launch", 0.0)
    v["hbm_bytes_per_launch"] = (f + w) * 1024.0
    v["hbm_bytes_per_launch_fetch_doubled"] = (2 * f + w) * 1024.0
print(json.dumps({"tag": tag, "workload": "cfg2 (64 hyps, 640x480, T=20480)", "kernels": out}, indent=1))

#!/usr/bin/env python
"""Summarise the rocprofv3 passes of tools/pmc_passes.sh into one JSON per workload:

    python profiles/summarize_sq.py <tag> <workload key>  >  profiles/<tag>_pmc_sq_<key>.json

Per kernel of the iteration: median counter value per launch of every collected counter, the average launch
duration of the un-profiled kernel trace, and the derived figures a reader can recompute from them:

  valu_issue_frac   = SQ_INSTS_VALU * 2 cycles / (1024 SIMD-32 * 2.4 GHz * duration)      (MI355X_MICROARCH.md: a wave64
                      VALU instruction issues over 2 cycles on a SIMD-32; 256 CUs x 4 SIMDs)
  valu_per_wave     = SQ_INSTS_VALU / SQ_WAVES
  hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) KiB * 1024   (raw; `_fetch_doubled` applies the guide's x2 correction for
                      wide coalesced reads, an upper reading for this engine's gathers)
  hbm_frac_of_peak  = hbm_bytes_per_launch / duration / 8 TB/s
  l2_hit_rate       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
SQ_* cycle counters are summed over the chip's SQs (quad-cycle units for WAVE_CYCLES / WAIT_* / ACTIVE_INST_*, see the guide).
"""
import collections
import csv
import glob
import json
import statistics
import sys

tag, key = sys.argv[1], sys.argv[2]
root = f"gpurun_out/pmc_{tag}_{key}"
KEEP = ("step_kernel", "big_pass_kernel", "shade_kernel", "edge_kernel", "finish_kernel")
short = lambda name: name.split("(")[0].replace("void ", "").strip()
out = collections.defaultdict(dict)
for path in glob.glob(f"{root}/*/**/pmc_counter_collection.csv", recursive=True) + glob.glob(f"{root}/*/pmc_counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if any(s in k for s in KEEP):
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            out[k]["vgpr"] = int(r["VGPR_Count"]); out[k]["sgpr"] = int(r["SGPR_Count"]); out[k]["lds_bytes"] = int(r["LDS_Block_Size"])
            out[k]["grid"] = int(r["Grid_Size"]); out[k]["workgroup"] = int(r["Workgroup_Size"])
    for (k, c), v in acc.items():
        out[k][c] = statistics.median(v)
        out[k]["launches"] = max(out[k].get("launches", 0), len(v))
for path in glob.glob(f"{root}/trace/**/t_kernel_stats.csv", recursive=True) + glob.glob(f"{root}/trace/t_kernel_stats.csv"):
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        if k in out:
            out[k]["avg_ns_unprofiled"] = float(r["AverageNs"])
            out[k]["calls_unprofiled"] = int(r["Calls"])
for k, v in out.items():
    dur = v.get("avg_ns_unprofiled")
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is not None and w is not None:
        v["hbm_bytes_per_launch"] = (f + w) * 1024.0
        v["hbm_bytes_per_launch_fetch_doubled"] = (2 * f + w) * 1024.0
        if dur:
            v["hbm_frac_of_peak"] = v["hbm_bytes_per_launch"] / (dur * 1e-9) / 8e12
    if "SQ_INSTS_VALU" in v:
        if v.get("SQ_WAVES"):
            v["valu_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        if dur:
            v["valu_issue_frac"] = v["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9 * dur * 1e-9)
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]) > 0:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
bench = None
try:
    for line in open(f"{root}/trace.log"):
        if line.startswith("{"):
            bench = json.loads(line)
except Exception:
    pass
sha = None
try:
    sys.path.insert(0, ".")
    import bench as _bench_py

    sha = _bench_py.csrc_sha16()  # (bench.py refuses counters of other kernel sources: roofline.counters_stale)
except Exception:
    pass
cal = None
try:
    cal = json.load(open(f"{root}/fetch_calibration.json"))
except Exception:
    pass
print(json.dumps({"tag": tag, "workload_key": key, "csrc_sha16": sha, "fetch_size_calibration": cal, "workload": bench["config"]["workload"] if bench else None,
                  "bench_under_kernel_trace": {k: bench[k] for k in ("value", "ms_per_step", "kernel_ms", "engine_status")} if bench else None,
                  "kernels": out}, indent=1))

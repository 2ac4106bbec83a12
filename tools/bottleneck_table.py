#!/usr/bin/env python
"""Summarise tools/bottleneck_passes.sh:   python tools/bottleneck_table.py <tag>  >  profiles/<tag>_bottleneck.md

Per kernel (step / shade) and batch size (told apart by the launch's grid): the un-profiled average launch duration, and for
every counter collected its median per launch, then the BUSY FRACTIONS a reader can recompute:

  unit busy fraction = <unit>_BUSY_sum / (instances x 2.4e9 Hz x duration)       instances: TA / TD / TCP 256 (one per CU), TCC 128
                       (16 channels x 8 XCDs), TCA 8 (or from <unit>_CYCLE_sum when the block counts its own cycles)
  GRBM_*_BUSY / GRBM_GUI_ACTIVE = fraction of the launch in which ANY instance of the unit was busy
  SQ_* are summed over the chip's SQs in quad-cycles (MI355X_MICROARCH.md)

The question: does some unit's busy fraction rise towards 1 as the load on the shared units doubles?"""
import collections
import csv
import glob
import statistics
import sys

tag = sys.argv[1]
BS = [int(a) for a in sys.argv[2:]] or [64, 128, 256, 512]   # the driver's batch sizes, in its order
N_IT = 24                                                      # ... and its iterations per engine (BN_ITERS)
root = f"gpurun_out/bn_{tag}"
KEEP = ("step_kernel", "shade_kernel")
CLK = 2.4e9


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return "step" if "step_kernel" in n else ("shade" if "shade_kernel" in n else n)


def rows_by_batch(path):
    """(kernel, B) -> rows: the driver runs its engines one after the other, N_IT launches of each kernel per engine, so the k-th
    launch of a kernel (in dispatch order) belongs to batch size BS[k // N_IT] (grids cannot tell: 20 slots x 64 hypotheses and
    10 x 128 are both 1 280 workgroups)."""
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in KEEP):
            per[short(r["Kernel_Name"])].append(r)
    out = collections.defaultdict(list)
    for k, rows in per.items():
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        rank = {d: i for i, d in enumerate(ids)}
        for r in rows:
            i = rank[int(r["Dispatch_Id"])]
            if i // N_IT < len(BS) and i % N_IT >= 2:  # (the first launches of an engine -- first iteration, calibration -- left out)
                out[(k, BS[i // N_IT])].append(r)
    return out


# ---- durations from the un-profiled trace, per (kernel, grid); the first launches of an engine (set-up, first iteration) dropped
dur = collections.defaultdict(list)
for path in glob.glob(f"{root}/trace/**/*kernel_trace.csv", recursive=True):
    for kb, rows in rows_by_batch(path).items():
        dur[kb] += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
dur = {k: statistics.mean(v) for k, v in dur.items()}
# ---- counters
val = collections.defaultdict(dict)
for path in glob.glob(f"{root}/*/**/pmc_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for (k, g), rows in rows_by_batch(path).items():
        for r in rows:
            acc[(k, g, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, g, c), v in acc.items():
        val[(k, g)][c] = statistics.median(v)

keys = sorted(val.keys() | dur.keys(), key=lambda kg: (kg[0] != "step", kg[1]))
grids = {}
for k in ("step", "shade"):
    gs = sorted(g for (kk, g) in keys if kk == k)
    grids[k] = gs
print(f"# Which unit saturates?  (tools/bottleneck_passes.sh {tag}; cfg2 mesh and frame, both faces, one chain of full-batch launches)\n")
print("Batch sizes are told apart by the order of the launches; under `rocprofv3 --pmc` kernels run one at a time, so the load of a second "
      "chain beside the first is reproduced by doubling the hypotheses of ONE launch (the same load on every shared unit).\n")
try:
    print("```\n" + open(f"{root}/trace.log").read().strip()[-600:] + "\n```\n")
except Exception:
    pass
INST = {"TA": 256, "TD": 256, "TCP": 256, "TCC": 128, "TCA": 8}


def frac_rows(k):
    rows = []
    gs = grids[k]
    allc = sorted({c for g in gs for c in val.get((k, g), {})})

    def per(c):
        return [val.get((k, g), {}).get(c) for g in gs]

    def add(name, fn):
        out = []
        for g in gs:
            try:
                out.append(fn(val.get((k, g), {}), dur.get((k, g))))
            except Exception:
                out.append(None)
        if any(x is not None for x in out):
            rows.append((name, out))

    add("launch duration, un-profiled (us)", lambda v, d: d / 1e3)
    add("workgroups", lambda v, d: None)
    add("GRBM_TA_BUSY / GUI_ACTIVE", lambda v, d: v["GRBM_TA_BUSY"] / v["GRBM_GUI_ACTIVE"])
    add("GRBM_TC_BUSY / GUI_ACTIVE", lambda v, d: v["GRBM_TC_BUSY"] / v["GRBM_GUI_ACTIVE"])
    add("GRBM_EA_BUSY / GUI_ACTIVE", lambda v, d: v["GRBM_EA_BUSY"] / v["GRBM_GUI_ACTIVE"])
    add("GRBM_SPI_BUSY / GUI_ACTIVE", lambda v, d: v["GRBM_SPI_BUSY"] / v["GRBM_GUI_ACTIVE"])
    add("GRBM_UTCL2_BUSY / GUI_ACTIVE", lambda v, d: v["GRBM_UTCL2_BUSY"] / v["GRBM_GUI_ACTIVE"])
    add("TA busy (TA_TA_BUSY_sum / 256 TAs x cycles)", lambda v, d: v["TA_TA_BUSY_sum"] / (INST["TA"] * CLK * d * 1e-9))
    add("TA_BUSY_avr / cycles", lambda v, d: v["TA_BUSY_avr"] / (CLK * d * 1e-9))
    add("TA addr stalled by TC / TA busy", lambda v, d: v["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / v["TA_TA_BUSY_sum"])
    add("TA data stalled by TC / TA busy", lambda v, d: v["TA_DATA_STALLED_BY_TC_CYCLES_sum"] / v["TA_TA_BUSY_sum"])
    add("TA addr stalled by TD / TA busy", lambda v, d: v["TA_ADDR_STALLED_BY_TD_CYCLES_sum"] / v["TA_TA_BUSY_sum"])
    add("TD busy (TD_TD_BUSY_sum / 256 x cycles)", lambda v, d: v["TD_TD_BUSY_sum"] / (INST["TD"] * CLK * d * 1e-9))
    add("TD stalled by TC / TD busy", lambda v, d: v["TD_TC_STALL_sum"] / v["TD_TD_BUSY_sum"])
    add("TCP busy (TCP_GATE_EN2_sum / 256 x cycles)", lambda v, d: v["TCP_GATE_EN2_sum"] / (INST["TCP"] * CLK * d * 1e-9))
    add("TCP pending-request stall / (256 x cycles)", lambda v, d: v["TCP_PENDING_STALL_CYCLES_sum"] / (INST["TCP"] * CLK * d * 1e-9))
    add("TCP stalled by TCR (L2 return) / (256 x cycles)", lambda v, d: v["TCP_TCR_TCP_STALL_CYCLES_sum"] / (INST["TCP"] * CLK * d * 1e-9))
    add("TCP atomic tag-conflict stall / (256 x cycles)", lambda v, d: v["TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum"] / (INST["TCP"] * CLK * d * 1e-9))
    add("TCP read tag-conflict stall / (256 x cycles)", lambda v, d: v["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / (INST["TCP"] * CLK * d * 1e-9))
    add("TCP->TCC read latency (cycles per request)", lambda v, d: v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"])
    add("TCP->TCC atomics with return per launch", lambda v, d: v["TCP_TCC_ATOMIC_WITH_RET_REQ_sum"])
    add("TCP->TCC atomics without return per launch", lambda v, d: v["TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"])
    add("TCP->TCC read requests per launch", lambda v, d: v["TCP_TCC_READ_REQ_sum"])
    add("TCP->TCC write requests per launch", lambda v, d: v["TCP_TCC_WRITE_REQ_sum"])
    add("TCC busy (TCC_BUSY_sum / TCC_CYCLE_sum)", lambda v, d: v["TCC_BUSY_sum"] / v["TCC_CYCLE_sum"])
    add("TCC busy (TCC_BUSY_sum / 128 channels x cycles)", lambda v, d: v["TCC_BUSY_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC requests per channel-cycle (TCC_REQ / 128 x cycles)", lambda v, d: v["TCC_REQ_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC atomics per channel-cycle", lambda v, d: v["TCC_ATOMIC_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC atomics / requests", lambda v, d: v["TCC_ATOMIC_sum"] / v["TCC_REQ_sum"])
    add("TCC tag stall / (128 x cycles)", lambda v, d: v["TCC_TAG_STALL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC EA write-request stall / (128 x cycles)", lambda v, d: v["TCC_EA0_WRREQ_STALL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC too-many-EA-writes stall / (128 x cycles)", lambda v, d: v["TCC_TOO_MANY_EA_WRREQS_STALL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC source FIFO full / (128 x cycles)", lambda v, d: v["TCC_SRC_FIFO_FULL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC latency FIFO full / (128 x cycles)", lambda v, d: v["TCC_LATENCY_FIFO_FULL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC input-buffer stall / (128 x cycles)", lambda v, d: v["TCC_IB_STALL_sum"] / (INST["TCC"] * CLK * d * 1e-9))
    add("TCC hit rate", lambda v, d: v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]))
    add("TCC EA atomics per launch (to memory)", lambda v, d: v["TCC_EA0_ATOMIC_sum"])
    add("TCC EA write requests per launch", lambda v, d: v["TCC_EA0_WRREQ_sum"])
    add("TCC EA read requests per launch", lambda v, d: v["TCC_EA0_RDREQ_sum"])
    add("TCA busy (TCA_BUSY_sum / TCA_CYCLE_sum)", lambda v, d: v["TCA_BUSY_sum"] / v["TCA_CYCLE_sum"])
    add("SQ: waves waiting (WAIT_ANY / WAVE_CYCLES)", lambda v, d: v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"])
    add("SQ: issue stalls (WAIT_INST_ANY / WAVE_CYCLES)", lambda v, d: v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"])
    add("SQ: LDS issue stalls (WAIT_INST_LDS / WAVE_CYCLES)", lambda v, d: v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"])
    add("SQ: VALU busy (ACTIVE_INST_VALU x 4 / 1024 SIMDs x cycles)", lambda v, d: v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * CLK * d * 1e-9))
    add("SQ: VMEM issue busy (ACTIVE_INST_VMEM x 4 / 1024 x cycles)", lambda v, d: v["SQ_ACTIVE_INST_VMEM"] * 4 / (1024 * CLK * d * 1e-9))
    add("SQ: scalar busy (ACTIVE_INST_SCA x 4 / 1024 x cycles)", lambda v, d: v["SQ_ACTIVE_INST_SCA"] * 4 / (1024 * CLK * d * 1e-9))
    add("SQ: LDS busy (ACTIVE_INST_LDS x 4 / 1024 x cycles)", lambda v, d: v["SQ_ACTIVE_INST_LDS"] * 4 / (1024 * CLK * d * 1e-9))
    add("SQ: VMEM instructions in flight per CU (INST_LEVEL_VMEM / BUSY_CU_CYCLES)", lambda v, d: v["SQ_INST_LEVEL_VMEM"] / v["SQ_BUSY_CU_CYCLES"])
    add("SQ: SMEM instructions in flight per CU", lambda v, d: v["SQ_INST_LEVEL_SMEM"] / v["SQ_BUSY_CU_CYCLES"])
    add("SQ: CU-time with a wave resident (BUSY_CU_CYCLES x 4 / 256 CUs x cycles)", lambda v, d: v["SQ_BUSY_CU_CYCLES"] * 4 / (256 * CLK * d * 1e-9))
    add("SQ: instruction fetches per launch", lambda v, d: v["SQ_IFETCH"])
    add("SQ_INSTS_VALU per launch", lambda v, d: v["SQ_INSTS_VALU"])
    add("SQ_INSTS_SALU per launch", lambda v, d: v["SQ_INSTS_SALU"])
    add("SQ_INSTS_SMEM per launch", lambda v, d: v["SQ_INSTS_SMEM"])
    add("SQ_INSTS_VMEM_RD per launch", lambda v, d: v["SQ_INSTS_VMEM_RD"])
    add("SQ_INSTS_VMEM_WR per launch", lambda v, d: v["SQ_INSTS_VMEM_WR"])
    add("SQ_INSTS_LDS per launch", lambda v, d: v["SQ_INSTS_LDS"])
    add("scalar data cache hit rate", lambda v, d: v["SQC_DCACHE_HITS"] / max(v["SQC_DCACHE_REQ"], 1))
    add("scalar data cache busy (SQC_DCACHE_BUSY_CYCLES / 64 SQCs x cycles)", lambda v, d: v["SQC_DCACHE_BUSY_CYCLES"] / (64 * CLK * d * 1e-9))
    add("instruction cache hit rate", lambda v, d: v["SQC_ICACHE_HITS"] / max(v["SQC_ICACHE_REQ"], 1))
    add("instruction cache busy (SQC_ICACHE_BUSY_CYCLES / 64 x cycles)", lambda v, d: v["SQC_ICACHE_BUSY_CYCLES"] / (64 * CLK * d * 1e-9))
    add("SQC->TC stall / (64 x cycles)", lambda v, d: v["SQC_TC_STALL"] / (64 * CLK * d * 1e-9))
    return rows, allc


for k in ("step", "shade"):
    gs = grids[k]
    if not gs:
        continue
    print(f"## {k}_kernel\n")
    print("| | " + " | ".join(f"{g} hypotheses" for g in gs) + " |")
    print("|---|" + "---|" * len(gs))
    rows, allc = frac_rows(k)
    for name, out in rows:
        if name == "workgroups":
            continue
        cells = ["" if x is None else (f"{x:.3g}" if abs(x) < 1000 else f"{x:,.0f}") for x in out]
        print(f"| {name} | " + " | ".join(cells) + " |")
    print()
    print("<details><summary>raw medians per launch</summary>\n")
    print("| counter | " + " | ".join(str(g) for g in gs) + " |")
    print("|---|" + "---|" * len(gs))
    for c in allc:
        print(f"| {c} | " + " | ".join("" if val.get((k, g), {}).get(c) is None else f"{val[(k, g)][c]:,.0f}" for g in gs) + " |")
    print("\n</details>\n")
try:
    print("failed passes:\n```\n" + open(f"{root}/failed.log").read() + "```")
except Exception:
    print("failed passes: none")

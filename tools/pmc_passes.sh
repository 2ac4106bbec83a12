#!/bin/bash
# tools/pmc_passes.sh <tag> <workload key> [bench.py args...]
# rocprofv3 counter passes for one workload, on the GPU box (run through gpurun): one un-profiled kernel trace (durations)
# and separate --pmc passes (kernel trace only, as the node policy and MI355X_MICROARCH.md prescribe: SQ has 8 slots per pass,
# FETCH_SIZE needs 3 of the 4 TCC slots and WRITE_SIZE 2, so they get their own passes).  Output: gpurun_out/pmc_<tag>_<key>/<pass>/.
# Summarise with:  python profiles/summarize_sq.py <tag> <key>  >  profiles/<tag>_pmc_sq_<key>.json
set -u
tag=$1; key=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
export DDX_TWO_STREAMS=0  # (counters are per full-batch launch; under --pmc the kernels run one at a time anyway)
out=gpurun_out/pmc_${tag}_${key}
mkdir -p "$out"
BENCH="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- $BENCH > "$out/trace.log" 2>&1
declare -A P
P[sq_insts]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT"
P[sq_cycles]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P[sq_valu]="SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32"
P[sq_lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
P[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
P[fetch]="FETCH_SIZE"
P[write]="WRITE_SIZE"
for name in sq_insts sq_cycles sq_valu sq_lds tcc fetch write; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${P[$name]} -d "$out/$name" -o pmc --output-format csv -- $BENCH > "$out/$name.log" 2>&1 \
        || echo "pass $name failed (rc $?)" >> "$out/failed.log"
done
ls -R "$out" | head -60
# ---- FETCH_SIZE calibration on a known count of 64-byte record gathers (and a wide coalesced stream of the same size)
if [ ! -f "$out/fetch_calibration.json" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather64 tools/ubench/gather64.hip 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/fetchcal" -o g --output-format csv -- /tmp/gather64 > "$out/fetchcal.log" 2>&1
  python - "$out" <<'PY'
import csv, glob, json, statistics, sys
out = sys.argv[1]
info = None
for line in open(f"{out}/fetchcal.log"):
    if line.startswith("{"):
        info = json.loads(line)
vals = {"gather64_kernel": [], "stream_kernel": []}
for path in glob.glob(f"{out}/fetchcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        for k in vals:
            if k in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                vals[k].append(float(r["Counter_Value"]))
if info and vals["gather64_kernel"]:
    g = statistics.median(vals["gather64_kernel"]) * 1024.0
    s = statistics.median(vals["stream_kernel"]) * 1024.0 if vals["stream_kernel"] else None
    cal = {"what": "tools/ubench/gather64.hip under rocprofv3 --pmc FETCH_SIZE: 262 144 random 64-byte records (3 x 16-byte loads each) from a 512 MiB "
                   "table per launch, and the same bytes as a wide coalesced stream",
           "gather_true_bytes_64B_sectors": info["bytes_per_gather_launch_64B_records"], "gather_FETCH_SIZE_bytes": g,
           "gather_true_over_reported": info["bytes_per_gather_launch_64B_records"] / g,
           "stream_true_bytes": info["bytes_per_stream_launch"], "stream_FETCH_SIZE_bytes": s,
           "stream_true_over_reported": (info["bytes_per_stream_launch"] / s) if s else None}
    json.dump(cal, open(f"{out}/fetch_calibration.json", "w"))
    print(json.dumps(cal))
PY
fi

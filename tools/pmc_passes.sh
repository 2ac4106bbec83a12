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

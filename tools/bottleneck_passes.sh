#!/bin/bash
# tools/bottleneck_passes.sh <tag> [B ...]
# Which shared unit saturates?  rocprofv3 counter passes (kernel trace + --pmc only, one small counter set per pass: the per-block
# slot counts of TA / TD / TCP / TCC / SQC on gfx950 are not documented, so a pass that asks for too much fails alone and is listed
# in failed.log) over tools/bottleneck_driver.py: cfg2's launches at 64 / 128 / 256 / 512 hypotheses.  One un-profiled kernel trace
# gives the durations.  Output gpurun_out/bn_<tag>/<pass>/;  summarise with  python tools/bottleneck_table.py <tag> > profiles/<tag>_bottleneck.md
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
export DDX_TWO_STREAMS=0
out=gpurun_out/bn_${tag}
mkdir -p "$out"
DRV="python tools/bottleneck_driver.py $*"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- $DRV > "$out/trace.log" 2>&1
declare -A P
P[grbm1]="GRBM_GUI_ACTIVE GRBM_TA_BUSY"
P[grbm2]="GRBM_TC_BUSY GRBM_EA_BUSY"
P[grbm3]="GRBM_SPI_BUSY GRBM_UTCL2_BUSY"
P[sq_a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P[sq_b]="SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT"
P[sq_c]="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_WAVES"
P[sqc_d]="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES"
P[sqc_i]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES"
P[sqc_t]="SQC_TC_REQ SQC_TC_STALL SQC_TC_DATA_READ_REQ SQC_TC_INST_REQ"
P[ta1]="TA_TA_BUSY_sum TA_BUSY_avr"
P[ta2]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P[ta3]="TA_FLAT_ATOMIC_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"
P[ta4]="TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
P[ta5]="TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum"
P[td1]="TD_TD_BUSY_sum TD_TC_STALL_sum"
P[td2]="TD_ATOMIC_WAVEFRONT_sum TD_LOAD_WAVEFRONT_sum"
P[tcp1]="TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
P[tcp2]="TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"
P[tcp3]="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
P[tcp4]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
P[tcp5]="TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
P[tcp6]="TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
P[tcp7]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
P[tcp8]="TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum"
P[tcp9]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
P[tcc1]="TCC_BUSY_sum TCC_CYCLE_sum"
P[tcc2]="TCC_ATOMIC_sum TCC_REQ_sum"
P[tcc3]="TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
P[tcc4]="TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum"
P[tcc5]="TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum"
P[tcc6]="TCC_EA0_RDREQ_sum TCC_IB_STALL_sum"
P[tcc7]="TCC_HIT_sum TCC_MISS_sum"
P[tcc8]="TCC_READ_sum TCC_WRITE_sum"
P[tcc9]="TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_ATOMIC_LEVEL_sum"
P[tcc10]="TCC_BUBBLE_sum TCC_NORMAL_WRITEBACK_sum"
P[tca]="TCA_BUSY_sum TCA_CYCLE_sum"
names=${BN_PASSES:-"grbm1 grbm2 grbm3 sq_a sq_b sq_c sqc_d sqc_i sqc_t ta1 ta2 ta3 ta4 ta5 td1 td2 tcp1 tcp2 tcp3 tcp4 tcp5 tcp6 tcp7 tcp8 tcp9 tcc1 tcc2 tcc3 tcc4 tcc5 tcc6 tcc7 tcc8 tcc9 tcc10 tca"}
for name in $names; do
    timeout 240 rocprofv3 --kernel-trace --pmc ${P[$name]} -d "$out/$name" -o pmc --output-format csv -- $DRV > "$out/$name.log" 2>&1 \
        || echo "pass $name failed (rc $?)" >> "$out/failed.log"
done
ls "$out"
cat "$out/failed.log" 2>/dev/null

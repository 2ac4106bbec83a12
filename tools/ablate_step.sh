#!/bin/bash
# tools/ablate_step.sh <tag> [B ...]
# Where do step_kernel's instructions go?  The product and the DDX_ABLATE = 1..4 builds (python tools/build_variant.py a<n> -DDDX_ABLATE=<n>,
# built beforehand: raster_dev.h says what each leaves out) over tools/ablate_driver.py: one un-profiled kernel trace (durations) and
# one SQ counter pass each.  Output gpurun_out/ab_<tag>/<lib>/;  summarise with  python tools/ablate_table.py <tag> [B ...]
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
out=gpurun_out/ab_${tag}
mkdir -p "$out"
DRV="python tools/ablate_driver.py $*"
for v in ${AB_LIBS:-product a1 a2 a3 a4}; do
    if [ "$v" = product ]; then unset DDX_LIB; else export DDX_LIB=$PWD/tools/_variants/libddx_$v.so; fi
    timeout 300 rocprofv3 --kernel-trace -d "$out/$v/trace" -o t --output-format csv -- $DRV > "$out/$v.trace.log" 2>&1 || echo "trace $v failed" >> "$out/failed.log"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR \
        -d "$out/$v/pmc" -o pmc --output-format csv -- $DRV > "$out/$v.pmc.log" 2>&1 || echo "pmc $v failed" >> "$out/failed.log"
done
ls "$out"; cat "$out/failed.log" 2>/dev/null

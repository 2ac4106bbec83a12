#!/bin/bash
# tools/kres.sh <file.hip> -- registers / spills / LDS / occupancy of every kernel of a source file (no GPU needed)
f=${1:-diffdope_amd/csrc/engine.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage ${DDX_CXXFLAGS} -c "$f" -o /dev/null 2>&1 \
 | grep remark | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk -F': ' '/Function Name/{name=$2} /TotalSGPRs/{sg=$2} /^VGPRs:/{v=$2} /AGPRs/{a=$2} /ScratchSize/{sc=$2} /Occupancy/{o=$2} /^VGPRs Spill/{sp=$2} /LDS Size/{printf "%-60s vgpr %3s agpr %2s sgpr %3s scratch %4s spill %3s occ %s lds %s\n", name, v, a, sg, sc, sp, o, $2}' | c++filt | sort -u

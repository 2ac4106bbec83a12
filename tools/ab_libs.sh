#!/bin/bash
# tools/ab_libs.sh <out file> <lib> [lib ...]     A/B of builds of libddx.so on ONE box: tools/ab_engine.py's windows with each library in turn
# ("product" = diffdope_amd/libddx.so, anything else = tools/_variants/libddx_<name>.so), twice over so that drift shows.
out=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
: > "$out"
for rep in 1 2; do
    for L in "$@"; do
        if [ "$L" = product ]; then unset DDX_LIB; else export DDX_LIB=$PWD/tools/_variants/libddx_$L.so; fi
        python tools/ab_engine.py "$PWD" ${AB_REPS:-5} 2>&1 | grep "us/it" | sed "s/^repo/$L/" >> "$out"
    done
done

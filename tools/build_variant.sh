#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags ...]
# A second libddx.so built from the working tree with extra flags (measurement builds: -DDDX_ABLATE=n, -DDDX_TRACE ...), written to
# tools/_variants/libddx_<name>.so (git-ignored; travels to the GPU box); load it with DDX_LIB=tools/_variants/libddx_<name>.so.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
out=tools/_variants/obj_$name
mkdir -p "$out"
FLAGS="-I${DDX_INC:-include} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
pids=()
SRC=${DDX_SRC:-diffdope_amd/csrc}
for s in $SRC/*.hip; do
    /opt/rocm/bin/hipcc $FLAGS "$@" -c "$s" -o "$out/$(basename "$s").o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_variants/libddx_$name.so "$out"/*.o -ldl
echo "built tools/_variants/libddx_$name.so"

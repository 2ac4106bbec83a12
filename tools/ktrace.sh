#!/bin/bash
# tools/ktrace.sh <tag> [bench.py args...] -- on the GPU box: rocprofv3 kernel trace of a short bench run, summary of the engine's kernels
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" > $out/bench.log 2>&1
f=$(find $out -name "t_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(s in r["Name"] for s in ("step_kernel", "big_pass", "shade_kernel", "edge_kernel", "finish_kernel"))]
tot = 0.0
for r in keep:
    n = int(r["Calls"])
    print(f'{r["Name"][:48]:50s} calls {n:4d} avg {float(r["AverageNs"])/1000:7.2f} us  min {float(r["MinNs"])/1000:7.2f}')
    if n > 50: tot += float(r["AverageNs"]) / 1000
print(f"sum of per-iteration kernels: {tot:.2f} us")
PY
grep -o '"value": [0-9.]*' $out/bench.log | head -1

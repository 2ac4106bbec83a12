#!/bin/bash
# tools/sweep.sh <tag> -- on the GPU box: one bench line per workload into gpurun_out/<tag>_bench_<workload>.json, summary on stdout
tag=$1; shift
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$name.json
  python - gpurun_out/${tag}_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f'{sys.argv[2]:12s} {d["value"]:9.0f} it/s  {d["ms_per_step"]*1000:7.2f} us  ', {k[:6]: round(v * 1000, 1) for k, v in d["kernel_ms"].items()}, f'rot {d["final_pose"]["rot_err_rad_best"]:.1e}')
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run cfg2 "$@"
run cfg2_steps20 --steps 20 --warmup 5 "$@"
for d in 5 3.75 2.5 1.8; do run cfg2_d$d --distance $d "$@"; done
for c in cfg1 cfg3 cfg3ref cfg4 cfg5 cfg50k64 lowpoly midpoly hugetri; do run $c --config $c "$@"; done

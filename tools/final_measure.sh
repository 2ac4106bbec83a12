#!/bin/bash
# final measurement session of the round: everything profiles/r5c_* is made of (run on the GPU box through gpurun)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r6d}
O=gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/${T}_pytest_gpu.log 2>&1; tail -3 $O/${T}_pytest_gpu.log
timeout 900 bash tools/pmc_passes.sh $T cfg2_d7.5 > $O/${T}_pmc_cfg2.log 2>&1
timeout 900 bash tools/pmc_passes.sh $T cfg2_d3.75 --distance 3.75 > $O/${T}_pmc_cfg2d.log 2>&1
timeout 900 bash tools/pmc_passes.sh $T cfg3_d7.5 --config cfg3 > $O/${T}_pmc_cfg3.log 2>&1
timeout 900 bash tools/pmc_passes.sh $T cfg4_gb512 --config cfg4 --global-batch 512 > $O/${T}_pmc_cfg4sat.log 2>&1
for k in cfg2_d7.5 cfg2_d3.75 cfg3_d7.5 cfg4_gb512; do python profiles/summarize_sq.py $T $k > $O/${T}_pmc_sq_$k.json 2>>$O/${T}_summ.err; done
cp $O/${T}_pmc_sq_*.json profiles/  # (bench.py reads its counters from profiles/: those of THIS build, made just above)
timeout 900 python bench.py > $O/${T}_bench.json 2>$O/${T}_bench.err; tail -c 400 $O/${T}_bench.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json 2>/dev/null; tail -c 300 $O/${T}_bench_steps20.json
timeout 1200 bash tools/sweep.sh $T > $O/${T}_sweep.log 2>&1; cat $O/${T}_sweep.log
timeout 300 bash tools/ktrace.sh $T > $O/${T}_ktrace_cfg2.log 2>&1; cat $O/${T}_ktrace_cfg2.log; cp $(find $O/trace_$T -name "t_kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
timeout 300 bash tools/ktrace.sh ${T}cfg3 --config cfg3 > $O/${T}_ktrace_cfg3.log 2>&1; cat $O/${T}_ktrace_cfg3.log; cp $(find $O/trace_${T}cfg3 -name "t_kernel_stats.csv" | head -1) $O/${T}_kernel_stats_cfg3.csv
T=$T python - <<'PY'
import json, os
T = os.environ["T"]
for k in ("cfg2_d7.5", "cfg2_d3.75", "cfg3_d7.5", "cfg4_gb512"):
    try:
        d = json.load(open(f"gpurun_out/{T}_pmc_sq_{k}.json"))
        for n, v in d["kernels"].items():
            print(k, n[:40], {x: (round(v[x], 3) if isinstance(v.get(x), float) else v.get(x)) for x in ("avg_ns_unprofiled", "valu_issue_frac", "hbm_bytes_per_launch", "hbm_frac_of_peak", "l2_hit_rate", "SQ_WAVES")},
                  "wait", round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3), "ldsconf", round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3))
    except Exception as e:
        print(k, "ERR", e)
PY
timeout 600 bash tools/mfma_pass.sh $T > $O/${T}_mfma_util.json 2>$O/${T}_mfma.err
timeout 120 python tools/launch_sync_cost.py 2>/dev/null | tail -6 > $O/${T}_launch_sync_cost.txt; cat $O/${T}_launch_sync_cost.txt
timeout 120 python tools/fixed_cost.py 2>/dev/null | tail -9 > $O/${T}_fixed_cost.txt; cat $O/${T}_fixed_cost.txt
DDX_TRACE=1 DDX_TWO_STREAMS=0 timeout 120 python tools/trace_kernels.py > $O/${T}_trace_cfg2.log 2>&1; cat $O/${T}_trace_cfg2.log
timeout 300 python tools/large_batch.py > $O/${T}_large_batch.log 2>&1; cat $O/${T}_large_batch.log
timeout 400 python tools/ab_engine.py . 2>/dev/null > $O/${T}_windows.log; cat $O/${T}_windows.log
timeout 300 python tools/multi_object_streams.py cfg5 > $O/${T}_multi_object_cfg5.log 2>&1; cat $O/${T}_multi_object_cfg5.log
DDX_API_NB=100 timeout 300 python tools/bench_opbyop.py cfg2 --api --graph > $O/${T}_opbyop.txt 2>&1; cat $O/${T}_opbyop.txt
timeout 900 python tools/cull_sweep.py 1000 0 > $O/${T}_cull_sweep.json 2>$O/${T}_cull_sweep.err; cat $O/${T}_cull_sweep.json
timeout 1800 bash tools/final_fuzz.sh $T 20 > /dev/null 2>&1; cat $O/${T}_fuzz_final.log
FUZZ_CULL=1 timeout 900 bash tools/final_fuzz.sh ${T}_cull 8 > /dev/null 2>&1; cat $O/${T}_cull_fuzz_final.log
echo FINAL DONE

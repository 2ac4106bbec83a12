cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2k
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc" gpurun_out/r2k/pytest.log | head
bash tools/pmc_passes.sh r2b cfg2_d7.5 > gpurun_out/r2k/pmc1.log 2>&1
bash tools/pmc_passes.sh r2b cfg2_d3.75 --distance 3.75 > gpurun_out/r2k/pmc2.log 2>&1
bash tools/pmc_passes.sh r2b cfg2_d1.8 --distance 1.8 > gpurun_out/r2k/pmc3.log 2>&1
bash tools/pmc_passes.sh r2b cfg3_d7.5 --config cfg3 > gpurun_out/r2k/pmc4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r2k/stats_cfg2 -o t --output-format csv -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2k/bench_traced.json 2> gpurun_out/r2k/bench_traced.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2k/stats_cfg3 -o t --output-format csv -- python bench.py --config cfg3 --no-cpu-baseline --no-extras > gpurun_out/r2k/bench_traced_cfg3.json 2>> gpurun_out/r2k/bench_traced.err
timeout 900 python bench.py > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err; echo "bench rc=$?"
for c in cfg3 cfg3ref cfg50k64 cfg1 cfg4 cfg5 lowpoly midpoly hugetri; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras > gpurun_out/r2k/bench_$c.json 2>> gpurun_out/r2k/bench.err; done
for d in 5 3.75 2.5 1.8; do timeout 300 python bench.py --distance $d --no-cpu-baseline --no-extras > gpurun_out/r2k/bench_cfg2_d$d.json 2>> gpurun_out/r2k/bench.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2k/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), {k[:7]:round(v*1e3,1) for k,v in d['kernel_ms'].items() if v>0})
    except Exception as e: print(f,'ERR',e)
PY

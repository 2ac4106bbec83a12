set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_20.json 2>> gpurun_out/r2a/bench.err
for c in cfg3 cfg50k64 cfg1 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras > gpurun_out/r2a/bench_$c.json 2>> gpurun_out/r2a/bench.err; done
bash tools/pmc_passes.sh r2a cfg2_d7.5 > gpurun_out/r2a/pmc1.log 2>&1
bash tools/pmc_passes.sh r2a cfg2_d3.75 --distance 3.75 > gpurun_out/r2a/pmc2.log 2>&1
bash tools/pmc_passes.sh r2a cfg2_d1.8 --distance 1.8 > gpurun_out/r2a/pmc3.log 2>&1
bash tools/pmc_passes.sh r2a cfg3_d7.5 --config cfg3 > gpurun_out/r2a/pmc4.log 2>&1
cat gpurun_out/r2a/bench.json | head -c 3000

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc\|^E  " gpurun_out/r2n/pytest.log | head -40
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
for c in lowpoly hugetri; do python bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200

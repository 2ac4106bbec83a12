cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 1200 python -m pytest tests -m gpu -q -k "render_texture_batch or api or bop or raster or example" > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc\|^E  " gpurun_out/r2l/pytest.log | head -30
python tools/bench_opbyop.py 2>&1 | tail -1
python tools/bench_opbyop.py cfg2 --torch-losses 2>&1 | tail -1
python tools/bench_opbyop.py cfg3ref 2>&1 | tail -1

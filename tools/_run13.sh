cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests -m gpu -q -k "raster or engine" -x > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc\|^E  " gpurun_out/r2m/pytest.log | head
bash tools/ab.sh "nofast4 fast4" "cfg2 cfg2@3.75 cfg3 cfg50k64 cfg5 midpoly" 2 2>&1 | tee gpurun_out/r2m/ab.log

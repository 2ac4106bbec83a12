cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed\|pytest rc\|mask loss of" gpurun_out/r2d/pytest.log | head -20
bash tools/ab.sh "base texq" "cfg2 cfg2@3.75 cfg2@1.8 cfg3 cfg50k64 cfg5" 2 2>&1 | tee gpurun_out/r2d/ab.log

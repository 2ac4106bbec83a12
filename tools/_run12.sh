cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 1200 python -m pytest tests -m gpu -q -k "example or pose_matrix" > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc\|^E  " gpurun_out/r2l/pytest.log | head -30

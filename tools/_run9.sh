cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 2400 python -m pytest tests -m gpu -q -s -k "tiers" > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
grep -n "^FAILED\|passed\|failed\|pytest rc\|tier [0-9]" gpurun_out/r2i/pytest.log | head -40

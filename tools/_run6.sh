cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed\|pytest rc\|^E  " gpurun_out/r2f/pytest.log | head -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/ab.sh "texq cull2" "cfg2 cfg2@3.75 cfg3 cfg50k64 cfg4 cfg5 cfg1 lowpoly midpoly" 1 2>&1 | tee gpurun_out/r2f/ab.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
for sgn in 0 1 -1; do echo "== DDX_CULL_TEST=$sgn"; DDX_CULL_TEST=$sgn bash tools/ab.sh "cull" "cfg2 cfg2@3.75 cfg2@1.8 cfg3 cfg50k64" 1; done 2>&1 | tee gpurun_out/r2e/cull.log

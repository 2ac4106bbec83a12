cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
cp diffdope_amd/libddx.so /tmp/keep.so
cp ab/trace0.so diffdope_amd/libddx.so
for c in cfg2 cfg3; do echo "=== $c update"; python tools/trace_update.py $c; echo "=== $c scatter"; python tools/trace_scatter.py $c; echo "=== $c shade colour"; python tools/trace_phases.py $c; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2g/trace.log
cp ab/trace1.so diffdope_amd/libddx.so
for c in cfg2; do echo "=== $c shade mask"; python tools/trace_phases_mask.py $c; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2g/trace.log
cp /tmp/keep.so diffdope_amd/libddx.so

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
for v in unset 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  if [ "$v" = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  bash tools/ab.sh "cur" "cfg2 cfg3 cfg1" 1
done 2>&1 | tee gpurun_out/r2h/kernarg.log

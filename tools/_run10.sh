cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
rocprofv3 --kernel-trace --stats -d gpurun_out/r2j/opbyop -o t --output-format csv -- python tools/bench_opbyop.py > gpurun_out/r2j/opbyop.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2j/opbyop/**/t_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f}us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
tail -2 gpurun_out/r2j/opbyop.log

"""One iteration of the op-by-op path as a kernel timeline, from a rocprofv3 database.

    cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
    DDX_API_NB=30 rocprofv3 --kernel-trace -d gpurun_out/opb_trace -o t -- python tools/bench_opbyop.py cfg2 --api
    python tools/opbyop_timeline.py gpurun_out/opb_trace/t_results.db > gpurun_out/<tag>_opbyop_timeline.txt

Prints the launches between the last but two and the last but one g-buffer forward pass -- one replay of the iteration captured
by DiffDope.run_optimization(fused=False, graph=True), the last thing tools/bench_opbyop.py --api runs --: start (us), duration
(us), grid, kernel; then the totals."""
import sqlite3
import sys

db = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/opb_trace/t_results.db"
c = sqlite3.connect(db)
rows = c.execute("select name, start, end-start, grid_x, grid_y from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("void gbuffer_fwd")]
a, b = idx[-3], idx[-2]
t0 = rows[a][1]
big = small = n_small = 0
for r in rows[a:b]:
    if r[2] > 8000:
        big += r[2]
    else:
        small += r[2]
        n_small += 1
    print(f"{(r[1] - t0) / 1e3:8.1f} {r[2] / 1e3:7.1f} {r[3]:8d}x{r[4]:<4d} {r[0][:120]}")
print(f"# period {(rows[b][1] - rows[a][1]) / 1e3:.1f} us, {b - a} launches; launches over 8 us: {big / 1e3:.1f} us; the {n_small} others: {small / 1e3:.1f} us")

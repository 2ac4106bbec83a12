#!/bin/bash
# tools/mfma_pass.sh <tag> -- on the GPU box: matrix-core counters of the fused engine's kernels (north_star asks for MFMA utilisation of the
# 4x4 pose x vertex product; it lives in step_kernel since round 3) -> gpurun_out/mfma_<tag>/, summary profiles-ready on stdout as JSON
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
export DDX_TWO_STREAMS=0  # (counters per full-batch launch)
out=gpurun_out/mfma_$tag
rm -rf $out; mkdir -p $out
for cfg in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $out/$cfg -o pmc --output-format csv -- \
      python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-convergence > $out/$cfg.log 2>&1
done
python - $out <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
rows = []
for cfg in ("cfg2", "cfg3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{out}/{cfg}/**/pmc_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if any(s in k for s in ("step_kernel", "shade_kernel", "edge_kernel", "finish_kernel")):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else None
        gui, busy, mf = m("GRBM_GUI_ACTIVE"), m("SQ_VALU_MFMA_BUSY_CYCLES"), m("SQ_INSTS_MFMA")
        rows.append({"workload": cfg, "kernel": k, "launches": len(c.get("SQ_INSTS_VALU", [])), "mfma_insts_per_launch": mf, "valu_insts_per_launch": m("SQ_INSTS_VALU"),
                     "mfma_busy_cycles": busy, "gui_active_cycles": gui, "busy_cu_cycles": m("SQ_BUSY_CU_CYCLES"),
                     "mfma_busy_frac": (busy / (gui * 1024.0)) if (busy is not None and gui) else None,
                     "mfma_flops_per_launch": (mf * 512.0) if mf is not None else None})
print(json.dumps({"note": "fused engine, bench.py --steps 30 --warmup 5 under rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                          "GRBM_GUI_ACTIVE SQ_INSTS_VALU; per-launch means.  v_mfma_f32_4x4x1_16b_f32 = 512 flop per instruction; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                          "(GRBM_GUI_ACTIVE x 1024 SIMDs).  The matrix core runs only the 4x4 pose x vertex product, inside step_kernel (four instructions per 64 vertex slots "
                          "of a meshlet + the 8 bounding-box corners): utilisation is a fraction of a percent by construction -- 32 flop per 28 bytes -- and the instruction is "
                          "there for the bit-exact k-ordered accumulation at one issue slot per 16 vertices, not for throughput.", "rows": rows}, indent=1))
PY

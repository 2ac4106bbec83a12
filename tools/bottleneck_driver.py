"""The workload tools/bottleneck_passes.sh profiles: cfg2's mesh and frame as ONE chain of full-batch launches at several batch
sizes in one process (64 = the headline launch, 128 / 256 = two / four of them side by side in one grid, 512 = saturation), both
faces drawn (the reference's rule).  Under `rocprofv3 --pmc` the launches of a process run one at a time, so "a second chain
beside the first" cannot be counted directly; doubling the hypotheses of one launch is the same load on every shared unit
(L2 channels, atomic units, texture addressers, scalar caches), and the kernels' grids tell the batch sizes apart in the CSV.

    python tools/bottleneck_driver.py [B ...]        (default 64 128 256 512)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl  # noqa: E402

Bs = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]
N = int(os.environ.get("BN_ITERS", "24"))
for B in Bs:
    w = wl.build(os.environ.get("BN_CONFIG", "cfg2"), torch.device("cuda:0"), B=B)
    eng, _ = wl.engine_for(w, wl.bench_lr_schedule(N, "adam"), optimizer="adam", single_stream=True)
    eng.run(N)
    eng.finish()
    st = eng.check()
    print(f"B={B} active_tiles={st['active_tiles']} slices={eng.slices}", flush=True)
    del eng, w
    torch.cuda.empty_cache()

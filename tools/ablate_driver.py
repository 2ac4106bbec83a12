"""Workload of tools/ablate_step.sh: cfg2's (BN_CONFIG) mesh and frame, both faces, one chain of full-batch launches at the given
batch sizes, learning rates ZERO -- the hypotheses stay at their initial poses whatever the loaded library draws, so the builds with
stages of the rasteriser left out (DDX_LIB=tools/_variants/libddx_a<n>.so) see the same geometry as the product.

    python tools/ablate_driver.py [B ...]        (default 64 512)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl  # noqa: E402

Bs = [int(a) for a in sys.argv[1:]] or [64, 512]
N = int(os.environ.get("BN_ITERS", "24"))
for B in Bs:
    w = wl.build(os.environ.get("BN_CONFIG", "cfg2"), torch.device("cuda:0"), B=B)
    eng, _ = wl.engine_for(w, [0.0] * N, optimizer="adam", single_stream=True, cull_backfaces=bool(int(os.environ.get("BN_CULL", "0"))))
    eng.run(N)
    eng.finish()
    print(f"B={B} slices={eng.slices} lib={os.environ.get('DDX_LIB', 'product')}", flush=True)
    del eng, w
    torch.cuda.empty_cache()

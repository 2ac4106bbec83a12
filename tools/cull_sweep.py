"""How often does deviation D5 show?  The fused engine culls the back faces of a CLOSED mesh while a hypothesis
lies inside the view volume; dr.rasterize (diffdope/diffdope.py:198-200) draws both faces.  In exact arithmetic the nearest surface at
every pixel centre is a front face, so the two rules give the same image; in float32 a back-facing SLIVER whose depth is extrapolated
through ill-conditioned barycentrics can come out nearer than the front face over it and own a pixel under the reference's rule that
it loses under culling.  This sweep counts that: random closed blob meshes, frames, distances and perturbed hypotheses through the
evaluation pass of two engines that differ ONLY in `cull_backfaces`; a hypothesis whose four losses and seven gradient components
are bit-identical under both rules has no pixel that changed owner (a changed owner changes the mask / depth / colour terms).

    python tools/cull_sweep.py [n_cases] [first_seed]      ->  one JSON line (profiles/r5_cull_sweep.json)
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda")
    B = 8
    cases_diff = hyps_diff = hyps = culled_cases = 0
    max_rel = 0.0
    worst = None
    t0 = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        rows, cols = int(rng.randint(6, 120)), int(rng.randint(8, 160))
        H, W = int(rng.randint(60, 480)), int(rng.randint(80, 640))
        dist = float(math.exp(rng.uniform(math.log(1.3), math.log(9.0))))
        textured = bool(rng.randint(2))
        wl.CONFIGS["_sweep"] = dict(rows=rows, cols=cols, B=B, H=H, W=W, textured=textured, weights=dict(rgb=0.7, depth=1.0, mask=1.0),
                                    tex=64 if textured else 0)
        w = wl.build("_sweep", dev, seed=seed0 + case, rot_deg=float(rng.uniform(2, 30)), trans_frac=float(rng.uniform(0.0, 0.08)), distance=dist)
        res = {}
        for cull in (True, False):
            eng, _ = wl.engine_for(w, [0.1], cull_backfaces=cull)
            losses, grad = eng.loss_and_grad()
            torch.cuda.synchronize()
            res[cull] = (losses.clone(), grad.clone(), eng.cull_sign, eng.status()["outside_view_volume"])
        if res[True][2] == 0:
            continue  # (not decided closed: nothing is culled, nothing to compare)
        culled_cases += 1
        la, ga = res[True][:2]
        lb, gb = res[False][:2]
        same = (la.view(torch.int32) == lb.view(torch.int32)).all(0) & (ga.view(torch.int32) == gb.view(torch.int32)).all(0)
        nd = int((~same).sum())
        hyps += B
        hyps_diff += nd
        cases_diff += int(nd > 0)
        if nd:
            rel = float(((la - lb).abs() / lb.abs().clamp_min(1e-12)).max())
            if rel > max_rel:
                max_rel, worst = rel, dict(seed=seed0 + case, mesh=[rows, cols], frame=[H, W], distance=round(dist, 3), textured=textured, hypotheses_differing=nd)
    out = dict(what="deviation D5 in float32: hypotheses whose losses / pose gradient differ in ANY bit between cull_backfaces=True and False (evaluation pass, "
                    "closed blob meshes 6x8 .. 120x160, frames 60x80 .. 480x640, distance 1.3 .. 9, 8 hypotheses per case perturbed by 2-30 deg / 0-8 %)",
               cases=culled_cases, cases_differing=cases_diff, hypotheses=hyps, hypotheses_differing=hyps_diff,
               max_relative_loss_difference=max_rel, worst_case=worst, seconds=round(time.time() - t0, 1), first_seed=seed0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The reference's examples/simple_scene.py on this build: load a config, refine, print the pose.

    python examples/make_example_data.py          # once: writes examples/data/
    python examples/simple_scene.py [configs/diffdope.yaml]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import diffdope as dd  # noqa: E402  (alias of diffdope_amd)


def main():
    cfg = dd.load_config(sys.argv[1] if len(sys.argv) > 1 else "configs/diffdope.yaml")
    ddope = dd.DiffDope(cfg=cfg)
    t0 = time.time()
    ddope.run_optimization()
    print(f"optimisation: {time.time() - t0:.3f} s for {cfg.hyperparameters.nb_iterations + 1} iterations x {cfg.hyperparameters.batchsize} hypotheses")
    print("argmin hypothesis:", int(ddope.get_argmin()))
    print("pose (OpenGL camera frame):\n", ddope.get_pose())
    try:
        from PIL import Image as PILImage

        PILImage.fromarray(ddope.render_img(batch_index=int(ddope.get_argmin()))).save("simple_scene_render.png")
        print("wrote simple_scene_render.png")
    except Exception as e:  # presentation only
        print("render_img skipped:", e)


if __name__ == "__main__":
    main()

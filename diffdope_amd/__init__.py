"""diffdope_amd -- MI355X-native render-and-compare pose refinement with the diff-dope API.

`import diffdope_amd as dd` (or the alias package `diffdope`) offers what `import diffdope as dd`
does in the reference (diffdope/__init__.py:1-7): xfm_points / xfm_vectors, render_texture_batch,
the loss functions and the DiffDope / Object3D / Mesh / Scene / Image / Camera classes.
"""
from .ops import xfm_points, xfm_vectors  # noqa: F401
from .render import (  # noqa: F401
    RasterizeContext,
    RasterizeGLContext,
    antialias,
    interpolate,
    rasterize,
    render_texture_batch,
    texture,
)
from .engine import RefineEngine, RefineEngineGroup  # noqa: F401
from .pose import matrix_batch_44_from_position_quat  # noqa: F401
from .api import (  # noqa: F401
    Camera,
    Cfg,
    DiffDope,
    Image,
    Mesh,
    Object3D,
    Scene,
    dist_batch_lr,
    l1_depth_with_mask,
    l1_edge,
    l1_mask,
    l1_rgb_with_mask,
    load_config,
    opencv_2_opengl,
)
from .viz import (  # noqa: F401  (the reference's module-level presentation helpers, diffdope.py:243-528)
    find_crop,
    getimg_stack,
    im_resize,
    make_grid_image,
    make_grid_overlay_batch,
)
from .viz import make_grid_tensor as make_grid  # noqa: F401

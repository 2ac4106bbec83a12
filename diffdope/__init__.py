"""Drop-in alias: `import diffdope as dd` resolves to the MI355X implementation (diffdope_amd)."""
from diffdope_amd import *  # noqa: F401,F403
from diffdope_amd import ops  # noqa: F401
from diffdope_amd.ops import xfm_points, xfm_vectors  # noqa: F401

__all__ = ["xfm_points", "xfm_vectors"]

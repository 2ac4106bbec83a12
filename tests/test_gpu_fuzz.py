"""A short run of the randomised parity sweep (tools/fuzz_parity.py): random mesh / frame sizes, distances down to the camera
plane, loss sets, all scatter variants -- engine losses and pose gradients, three-iteration SGD trajectories and the
materialising path against the oracle, op-level triangle ids bit for bit.  The long sweeps (thousands of cases, dense meshes
on large frames with FUZZ_BIG=1) are run by hand; this keeps a fixed slice of them in the suite."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_sweep_fixed_slice():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    old = os.environ.pop("DDX_SCATTER_EXCHANGE", None)
    try:
        # (two slices around the seeds -- 24050, 25630 -- that exposed the 32-bit edge steps of the tile pass)
        bad, stats = fuzz.sweep(120, 24000, verbose=True)
        bad2, stats2 = fuzz.sweep(120, 25560, verbose=True)
        bad += bad2
        stats = {k: (max(v, stats2.get(k, 0)) if k.startswith("max") else v + stats2.get(k, 0)) for k, v in stats.items()}
    finally:
        os.environ.pop("DDX_SCATTER_EXCHANGE", None)
        if old is not None:
            os.environ["DDX_SCATTER_EXCHANGE"] = old
    assert bad == 0, stats
    assert stats["outside"] > 20 and stats["big"] > 60 and stats["materialising"] > 20 and stats["trajectories"] > 40


def test_randomised_triangle_soups_fixed_slice():
    """300 random clip-space soups (sub-pixel to far beyond the frame, w from 1e-5 to 10 and negative, near / far violations,
    shared and degenerate triangles): op-level ids, u, v, z/w, rasterize backward, antialias and interpolate forward / backward."""
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, stats = fuzz.sweep_soups(300, 4242, verbose=True)
    assert bad == 0, stats
    assert stats["drawn"] > 1_000_000 and stats["straddlers"] > 1000 and stats["near_eye"] > 1000


def test_randomised_small_ops_and_fused_adam_fixed_slice():
    """150 cases of the small ops with random shapes and extreme values (xfm bit for bit, texture with uv far outside [0,1],
    masked L1 at any element count, the pose-matrix op) and 50 fused-Adam sequences, each step against the oracle's Adam
    teacher-forced with the oracle's gradient at the engine's own parameters."""
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, stats = fuzz.sweep_ops(150, 777, verbose=True)
    assert bad == 0, stats
    assert stats["adam"] == 50 and stats["xfm"] == 150


def test_randomised_engine_soups_fixed_slice():
    """300 arbitrary object-space triangle soups through the fused engine (open, self-intersecting, degenerate, duplicated, more
    vertices than triangles): losses and pose gradients against the oracle, same culling decision."""
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, stats = fuzz.sweep_engine_soups(150, 306440, verbose=True)   # (holds seed 306514: a flat two-sided patch)
    bad2, _ = fuzz.sweep_engine_soups(150, 313350, verbose=True)      # (holds seed 313409: the same, decided the other way round 2)
    assert bad + bad2 == 0, stats


def test_randomised_engine_state_machine_fixed_slice():
    """120 cases of the engine as a state machine: six iterations in one go against the same iterations in random pieces with
    evaluation passes in between and graph replay, after a rewind / new observation, after another observation and back, and as
    two shards with the unsharded run's slice counts -- parameters, loss log and pose log bit for bit."""
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, stats = fuzz.sweep_state(120, 31337, verbose=True)
    assert bad == 0, stats
    assert stats["graphs"] > 40

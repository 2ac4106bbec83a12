"""CPU-only checks of the boundary and the host logic: the C-ABI library loads and exports every symbol
include/ddx.h declares, the product refuses CPU tensors instead of falling back, and the host-side
pieces (topology, projection, LR schedule, pose matrix, sharding) agree with the reference's golden
vectors / the oracle.  No GPU compute is called here."""
import ctypes
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol():
    from diffdope_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge

        ge.build_lib()
    names = _lib.declared_symbols()
    assert len(names) >= 22 and "ddx_engine_run" in names and "ddx_xfm_bwd_full" in names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libddx.so does not export {n}"
        assert n in _lib._SIGNATURES, f"{n} has no ctypes signature"
    lib = _lib.load()
    assert lib.ddx_version() == 100
    # argument validation works without a device: NULL pointers / bad shapes are rejected with a message
    assert lib.ddx_xfm_fwd(None, 0, None, 1, 1, 1, None, 0, None) == -1
    assert b"NULL" in lib.ddx_last_error()
    assert lib.ddx_rasterize_scratch_bytes(0, 1, 1, 1, 1) == 0
    assert lib.ddx_rasterize_scratch_bytes(2, 60, 100, 48, 64) > 2 * 48 * 64 * 8
    d = _lib.EngineDesc()
    assert lib.ddx_engine_scratch_bytes(ctypes.byref(d)) == 0  # all-zero desc is invalid
    assert ctypes.sizeof(_lib.EngineDesc) == 27 * 4 and ctypes.sizeof(_lib.EngineBuffers) == 17 * 8


def test_ctypes_structs_follow_the_header_member_for_member():
    """ddx_engine_desc / ddx_engine_buffers in include/ddx.h against their ctypes mirrors: same members, same order, same scalar
    types (the size check above would not notice two swapped int32 fields)."""
    import re

    from diffdope_amd import _lib

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ddx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def members(name):
        body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", text, flags=re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const )?(\w+)\s*(\*?)\s*(.*)$", decl)
            ctype, ptr = m.group(2), bool(m.group(3))
            for nm in m.group(4).split(","):
                nm = nm.strip()
                p = ptr or nm.startswith("*")
                out.append((nm.lstrip("* ").strip(), "ptr" if p else ctype))
        return out

    kinds = {ctypes.c_int32: "int32_t", ctypes.c_float: "float", ctypes.c_void_p: "ptr", ctypes.c_size_t: "size_t"}
    for cname, struct in (("ddx_engine_desc", _lib.EngineDesc), ("ddx_engine_buffers", _lib.EngineBuffers)):
        want = members(cname)
        got = [(n, kinds[t]) for n, t in struct._fields_]
        assert got == want, (cname, [x for x in zip(got, want) if x[0] != x[1]][:3], len(got), len(want))


def test_ctypes_signatures_follow_the_header_prototypes():
    """Every prototype of include/ddx.h against the ctypes signature it is bound with: return type, number of arguments, and the
    kind of each (pointer / int / long long / size_t / float) -- a float handed over where the C side expects an int would be
    passed in the wrong register without any error."""
    import re

    from diffdope_amd import _lib

    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER).read(), flags=re.S)
    protos = re.findall(r"^\s*([\w][\w\s\*]*?)\s*\b(ddx_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M)
    assert len(protos) >= 45

    def kind(decl, ret=False):
        decl = " ".join(decl.split())
        if "*" in decl:
            return ctypes.c_char_p if (ret and "char" in decl) else ctypes.c_void_p
        base = decl if ret else " ".join(decl.split()[:-1])  # (an argument carries its name)
        return {"int": ctypes.c_int, "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t, "float": ctypes.c_float, "void": None}[base]

    seen = set()
    for ret, name, args in protos:
        res, argt = _lib._SIGNATURES[name]
        want_args = [kind(a) for a in args.split(",") if " ".join(a.split()) not in ("", "void")]
        assert kind(ret, ret=True) == res, (name, "return type", ret, res)
        norm = lambda t: ctypes.c_void_p if (isinstance(t, type) and issubclass(t, ctypes._Pointer)) else t  # (typed pointers count as pointers)
        got_args = [norm(t) for t in argt]
        assert got_args == want_args, (name, [i for i, (g, w) in enumerate(zip(got_args, want_args)) if g != w], len(got_args), len(want_args))
        seen.add(name)
    assert seen == set(_lib._SIGNATURES), set(_lib._SIGNATURES) ^ seen


def test_no_cpu_fallback_in_the_product_path():
    import diffdope_amd as dd

    with pytest.raises(RuntimeError, match="CUDA/ROCm"):
        dd.xfm_points(torch.randn(1, 4, 3), torch.randn(1, 4, 4))
    with pytest.raises(RuntimeError, match="CUDA/ROCm"):
        dd.rasterize(dd.RasterizeGLContext(), torch.randn(1, 3, 4), torch.zeros(1, 3, dtype=torch.int32), [8, 8])
    with pytest.raises(RuntimeError):
        dd.RefineEngine(torch.randn(4, 3), torch.zeros(1, 3, dtype=torch.int32), torch.eye(4), [8, 8],
                        {"segmentation": torch.zeros(8, 8, 3)}, torch.zeros(7, 1), torch.ones(1), [0.1], dict(mask=1.0))
    # the reference's own validation path (use_python, ops.py:137-141) is plain torch and works anywhere
    out = dd.xfm_points(torch.randn(2, 5, 3), torch.randn(2, 4, 4), use_python=True)
    assert out.shape == (2, 5, 4)
    # nothing under diffdope_amd imports the oracle
    import diffdope_amd

    root = os.path.dirname(diffdope_amd.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if f == "_smoke.py":
                    continue  # smoke() is a checker and may use the oracle
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} touches the oracle"


def test_topology_build_matches_oracle_and_handles_nonmanifold():
    from diffdope_amd import _lib, synthetic as syn
    from oracle import oracle as orc

    lib = _lib.load()
    _, tri, _ = syn.blob_mesh(9, 11, seed=3)
    tri = np.ascontiguousarray(tri)
    opp = np.empty_like(tri)
    assert lib.ddx_topology_build(tri.ctypes.data, tri.shape[0], opp.ctypes.data) == 0
    assert np.array_equal(opp, orc.build_opposite(tri))
    # an edge shared by three triangles: lowest-indexed other triangle wins
    fan = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4]], np.int32)
    opp = np.empty_like(fan)
    lib.ddx_topology_build(fan.ctypes.data, 3, opp.ctypes.data)
    assert np.array_equal(opp, orc.build_opposite(fan))
    assert opp[0, 2] == 3 and opp[1, 2] == 2 and opp[2, 2] == 2


def test_host_pose_projection_lr_match_reference_goldens(golden_dir):
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl

    g = np.load(os.path.join(golden_dir, "g2_pose.npz"))
    params = [torch.tensor(g["params"][i], requires_grad=True) for i in range(7)]
    q = torch.stack(params[:4], dim=0).T
    q = q / torch.norm(q, dim=1).reshape(-1, 1)
    mtx = dd.matrix_batch_44_from_position_quat(p=torch.stack(params[4:], dim=0).T, q=q)
    np.testing.assert_allclose(mtx.detach().numpy(), g["mtx"], rtol=1e-6, atol=1e-6)
    mtx.backward(torch.tensor(g["dmtx"]))
    np.testing.assert_allclose(np.stack([p.grad.numpy() for p in params]), g["dparams"], rtol=1e-4, atol=1e-5)
    g3 = np.load(os.path.join(golden_dir, "g3_proj.npz"))
    for i in range(int(g3["n"])):
        a = g3[f"cam{i}_args"]
        np.testing.assert_allclose(wl.projection_matrix(a[0], a[1], a[2], a[3], int(a[4]), int(a[5]), a[6], a[7]), g3[f"cam{i}_proj"], rtol=1e-12)
    g5 = np.load(os.path.join(golden_dir, "g5_lr.npz"))
    for i in range(int(g5["n"])):
        nb, base, decay = g5[f"s{i}_args"]
        np.testing.assert_allclose(wl.lr_schedule(int(nb), base, decay), g5[f"s{i}_lr"], rtol=1e-14)


def test_shard_ranges_partition_the_batch():
    from diffdope_amd.dist import shard_range

    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher (the driver's plain form) re-executes itself under
    torch.distributed.run with N ranks on 127.0.0.1; with WORLD_SIZE in the environment it does not."""
    import importlib
    import sys

    bench = importlib.import_module("bench")
    seen = {}

    class Stop(Exception):
        pass

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise Stop

    monkeypatch.setattr(bench.os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(Stop):
        bench.main()
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and a[-7].endswith("bench.py")
    # and the byte models the JSON line carries are self-consistent
    c = bench.compulsory_bytes(10449, 20480, 307200, 64, 1600, 228000.0, True, 2, 16, dict(rgb=True, depth=False, mask=True, edge=False))
    assert abs(c["iteration"] - (c["step_kernel"] + c["shade_kernel"])) < 1e-6
    assert 0 < c["shade_kernel"] < bench.algorithmic_bytes(10449, 20480, 307200, 64)["shade_kernel"]


def test_pixel_derivative_placeholders_raise_when_consumed():
    """rast_db / diff_attrs outputs are not computed (diff-dope discards them, diffdope.py:212-226): the placeholders can be
    inspected and passed on, but any arithmetic / indexing / copy on them raises instead of silently reading zeros."""
    from diffdope_amd.render import PixelDerivativesNotComputed, _not_computed

    z = _not_computed((2, 3, 4, 4), "cpu")
    assert isinstance(z, PixelDerivativesNotComputed) and tuple(z.shape) == (2, 3, 4, 4) and z.dim() == 4 and z.dtype == torch.float32
    assert "PixelDerivativesNotComputed" in repr(z)
    for use in (lambda: z + 1, lambda: z[0], lambda: z.sum(), lambda: torch.cat([z, z]), lambda: z.contiguous(), lambda: z * z, lambda: z.cpu()):
        with pytest.raises(RuntimeError, match="pixel derivatives"):
            use()

"""Measurement tool: build a VARIANT of libddx.so (extra -D flags, or the sources of another checkout) next to the product library,
as tools/_variants/libddx_<name>.so (git-ignored; it travels to the GPU box).  A process picks it with DDX_LIB=<path> (diffdope_amd/_lib.py) -- experiments only; the product
and the tests load diffdope_amd/libddx.so.
usage: python tools/build_variant.py <name> [--src <checkout root>] [-DFLAG ...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    name, src, flags = sys.argv[1], ROOT, []
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--src":
            src = os.path.abspath(args.pop(0))
        else:
            flags.append(a)
    csrc = os.path.join(src, "diffdope_amd", "csrc")
    objdir = os.path.join(ROOT, "diffdope_amd", "csrc", "build", "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs, objs = [], []
    for s in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc] + ge.HIPCC_FLAGS + ["-I", os.path.join(src, "include"), "-I", ROOT] + flags + ["-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"hipcc failed for {s}:\n{out}")
    os.makedirs(os.path.join(ROOT, "tools", "_variants"), exist_ok=True)
    lib = os.path.join(ROOT, "tools", "_variants", f"libddx_{name}.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit("link failed:\n" + r.stdout)
    print(lib)


if __name__ == "__main__":
    main()

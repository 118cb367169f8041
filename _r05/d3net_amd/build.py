"""Build libd3hip.so (the gfx950 C-ABI library) in-tree with hipcc.

`python -m d3net_amd.build` or `d3net_amd.build.build()`.  hipcc cross-compiles for gfx950
without a GPU, so this also runs in the CPU-only build container; the resulting
d3net_amd/lib/libd3hip.so is git-ignored but travels with the tree to the GPU box.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
SO = os.path.join(LIBDIR, "libd3hip.so")

ARCH = "gfx950"
# -ffp-contract=off: the indexing / segment kernels must round exactly like the reference's C
# expressions; kernels that want FMAs ask for them explicitly (fmaf / MFMA).
CXXFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-Wno-unused-result", "-I" + os.path.join(HERE, "..", "include")] + os.environ.get("D3_CXX_EXTRA", "").split()


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libd3hip.so cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(paths):
    h = hashlib.sha1()
    for p in sorted(paths):
        h.update(p.encode()); h.update(open(p, "rb").read())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "d3hip.h")]
    stamp_file = obj + ".stamp"
    stamp = _stamp(deps)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [_hipcc()] + CXXFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    open(stamp_file, "w").write(stamp)
    return obj


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        # -z defs: an undefined host symbol (a helper renamed in one translation unit only) fails the build, not the first call
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-z,defs", "-o", SO] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

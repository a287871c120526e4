"""Build libzklc_mi355.so (hipcc, gfx950 only) in-tree.

    python -m zklc_amd.build            # or: python zk-light-client-implementation_amd/build.py

Every csrc/*.hip is compiled to build/<name>.o (re-used while its sources and
headers are older than the object) and linked into lib/libzklc_mi355.so.  The
.so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libzklc_mi355.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _newest(paths):
    return max((os.path.getmtime(p) for p in paths), default=0.0)


HOST_CXX = os.environ.get("CXX", "g++")
HOST_FLAGS = ["-O3", "-funroll-loops", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]


def _compile(src, obj, log):
    # *.hip = device + host code for gfx950 (hipcc); *.cpp = host-only helpers (transcript), plain C++
    cmd = ([HIPCC] + FLAGS if src.endswith(".hip") else [HOST_CXX] + HOST_FLAGS) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if log:
        print("[zklc build] compiled", os.path.basename(src), file=sys.stderr)
    return obj


def build(verbose=True, force=False):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    dep_time = _newest(hdrs + [os.path.abspath(__file__)])
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(BUILD, os.path.basename(s).replace(".", "_") + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), dep_time):
            jobs.append((s, o))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[zklc build] linked", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

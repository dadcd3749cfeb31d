"""Builds libvanerf_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvanerf_hip.so")
SOURCES = ["api.cpp", "weights_pack.cpp", "query_kernel.hip", "render_kernels.hip", "mesh_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vanerf_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    extra = tuple(extra) + tuple(os.environ.get("VANERF_HIPCC_FLAGS", "").split())
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src + ".o")
        cmd = [hipcc, *FLAGS, *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)

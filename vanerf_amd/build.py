"""Builds libvanerf_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvanerf_hip.so")
SOURCES = ["api.cpp", "weights_pack.cpp", "pass.cpp", "query_kernel.hip", "render_kernels.hip", "mesh_kernels.hip", "train_kernels.hip", "query_backward.hip", "weights_update.hip", "weight_products.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]
# per-file flags.  query_kernel.hip: MFMA results are allocated in VGPRs (the chained layers read every accumulator element once with VALU
# instructions; with the default AGPR form each read is a v_accvgpr_read first: 667 -> 95 per 32-sample group), and the SLP vectoriser
# stays off (it packs adjacent f32 multiplies / adds into v_pk_* instructions, which cost more beside MFMAs than the scalar pairs).
FILE_FLAGS = {"query_kernel.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
              "query_backward.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
              # mesh_query_accel_kernel sits exactly at 128 registers (four waves per SIMD): loop-invariant address arithmetic goes back into the
              # work loop instead of being parked in scratch (7 spilled VGPRs without the flag; scratch fails the build, _check_no_scratch)
              "mesh_kernels.hip": ["-mllvm", "-sink-insts-to-avoid-spills"]}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vanerf_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


# query_backward_kernel once carried 1 900-3 600 spilled SGPRs (row offsets hoisted out of its group loop); the build with 3 600 restored wrong
# offsets in the second and later groups of a wave -- 2 % off in one gate's gradient, every test at one group per wave green.  A few hundred is
# what the chained kernels need for their constants.
MAX_SGPR_SPILLS = 256


def _check_no_scratch(src, remarks):
    """Every kernel of this library is written to live in registers: a build whose register allocation spills to scratch memory is a
    performance cliff that still passes every test (query_kernel once went from 0 to 59 spilled VGPRs through an innocent-looking
    change), so it fails the build.  VANERF_ALLOW_SCRATCH=1 lets experiments through."""
    name, bad, sgpr = None, [], []
    for line in remarks:
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) > 0:
            bad.append((name, int(m.group(1))))
        m = re.search(r"SGPRs Spill: (\d+)", line)
        if m and int(m.group(1)) > MAX_SGPR_SPILLS:
            sgpr.append((name, int(m.group(1))))
    if bad and os.environ.get("VANERF_ALLOW_SCRATCH") != "1":
        raise RuntimeError(f"{src}: kernels spill to scratch memory: {bad} (set VANERF_ALLOW_SCRATCH=1 to build anyway)")
    if sgpr and os.environ.get("VANERF_ALLOW_SCRATCH") != "1":
        raise RuntimeError(f"{src}: kernels spill thousands of SGPRs into VGPR lanes: {sgpr} -- loop-invariant address arithmetic hoisted out of a "
                           "work loop; make the stride opaque per iteration (query_backward.hip) (VANERF_ALLOW_SCRATCH=1 builds anyway)")


def _check_wide_stores(src, asm_path):
    """A 12/16-byte buffer store whose offset sits in an SGPR reads its data registers late; one build of query_backward_kernel had a
    v_pk_mov_b32 into them directly behind such a store (lanes 12-15 / 28-31 of one channel stored the next value, differently from run to
    run; the compiler inserted no wait state).  No kernel of this library may contain the pattern: wide buffer stores keep their whole
    offset in the VGPR (literal 0 as the scalar offset)."""
    bad = []
    try:
        text = open(asm_path).read()
    except OSError:
        return
    for m in re.finditer(r"buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*v\d+,\s*s\[\d+:\d+\],\s*(s\d+|m0|ttmp\d+)\b[^\n]*", text):
        bad.append(m.group(0).strip())
    if bad and os.environ.get("VANERF_ALLOW_SCRATCH") != "1":
        raise RuntimeError(f"{src}: {len(bad)} wide buffer store(s) with an SGPR offset (store-data hazard), e.g. `{bad[0]}`")


def build(force=False, verbose=False, extra=(), out=None):
    """out: build an experiment variant into another file (tools/build_variants.py); the product library is LIB."""
    extra = tuple(extra) + tuple(os.environ.get("VANERF_HIPCC_FLAGS", "").split())
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src + ".o") if out is None else out + "." + src + ".o"
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".hip"):
            cmd[1:1] = ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics", "-save-temps=obj"]  # remarks without source snippets; the device assembly for _check_wide_stores
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        remarks = [l for l in res.stderr.splitlines() if "kernel-resource-usage" in l]
        other = "\n".join(l for l in res.stderr.splitlines() if "kernel-resource-usage" not in l)
        if other.strip():
            print(other, file=sys.stderr, flush=True)
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)
        _check_no_scratch(src, remarks)
        if src.endswith(".hip"):
            stem = os.path.join(os.path.dirname(obj), os.path.splitext(src)[0])
            _check_wide_stores(src, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
            for f in os.listdir(os.path.dirname(obj)):  # the other intermediates of -save-temps
                if f.startswith(os.path.splitext(src)[0] + "-h") or f.startswith(src + "-hip-"):
                    os.remove(os.path.join(os.path.dirname(obj), f))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)

"""A/B builds of the per-sample kernel: compiles vanerf_amd/csrc/query_kernel.hip with extra flags into exp/libvanerf_<name>.so (the other
translation units are taken from the product build).  Run a tool against one with VANERF_HIP_LIB=exp/libvanerf_<name>.so.
usage: python tools/build_variants.py name1='-DX -mllvm -y' name2=...      (exp/ is git-ignored; it travels to the GPU box with gpurun)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vanerf_amd import build as B  # noqa: E402

B.build(verbose=False)
os.makedirs(os.path.join(ROOT, "exp"), exist_ok=True)
others = [os.path.join(B.HERE, "lib", s + ".o") for s in B.SOURCES if s != "query_kernel.hip"]
procs = []
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    obj = os.path.join(ROOT, "exp", name + ".query.o")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *B.FLAGS, *B.FILE_FLAGS["query_kernel.hip"], *flags.split(), "-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics",
           "-x", "hip", "-c", os.path.join(B.CSRC, "query_kernel.hip"), "-o", obj]
    procs.append((name, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
for name, obj, p in procs:
    err = p.communicate()[1]
    if p.returncode:
        print(err, file=sys.stderr)
        raise SystemExit(f"{name}: compile failed")
    use, on = [], False
    for l in err.splitlines():  # resource usage of the split-bf16 kernel
        if "Function Name:" in l:
            on = "query_kernelILi1" in l
        elif on and any(k in l for k in (" VGPRs:", "AGPRs:", "ScratchSize", "VGPRs Spill")):
            use.append(l.split("remark:")[-1].split("[-R")[0].strip())
    lib = os.path.join(ROOT, "exp", f"libvanerf_{name}.so")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *others])
    print(name, "->", lib, "|", "; ".join(use))

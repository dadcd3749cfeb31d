"""Approximate VGPR pressure profile of a (mostly straight-line) kernel from its gfx950 assembly: linear-scan liveness over the text of one
function, treating the body as straight-line code (query_kernel's sample loop is one 10 k-instruction block with a few short branches).
For every instruction: registers written (first operand of v_/ds_read/buffer_load/global_load ...) and read (the rest); a register is live
from a write to its last read before the next write.  Prints the pressure every `--every` lines with the number of MFMAs seen so far, and
the peak.  usage: python tools/asm_pressure.py file.s 'query_kernelILi1' [--every 200]"""
import argparse
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("asm")
ap.add_argument("func")
ap.add_argument("--every", type=int, default=250)
ap.add_argument("--loop", default=None, help="label of the loop header: registers live at the header are treated as live around the back edge")
args = ap.parse_args()

lines = open(args.asm).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(args.func) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


NO_DST = ("ds_write", "global_store", "buffer_store", "scratch_store", "v_cmp", "v_cmpx", "global_atomic_add ", "s_", "v_nop", "buffer_wbl2", "buffer_inv")
ins = []  # (line_no, writes, reads, is_mfma)
for n, l in enumerate(body):
    t = l.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        continue
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    w, r = [], []
    if op.startswith(NO_DST) and not op.startswith(("s_", "v_readlane", "v_readfirstlane")):
        for o in ops:
            r += regs(o)
    elif op.startswith("s_") or op.startswith(("v_readlane", "v_readfirstlane")):
        for o in ops[1:]:
            r += regs(o)
    else:
        if ops:
            w = regs(ops[0])
        for o in ops[1:]:
            r += regs(o)
        if op.startswith(("v_fmac", "v_mac", "v_writelane", "v_dot2c")) or (op.startswith("v_mfma") and False):
            r += w
    ins.append((n, w, r, op.startswith("v_mfma")))

# backward scan for liveness
live = set()
profile = [0] * len(ins)
for k in range(len(ins) - 1, -1, -1):
    n, w, r, _ = ins[k]
    for x in w:
        live.discard(x)
    for x in r:
        live.add(x)
    profile[k] = len(live)
peak = max(range(len(ins)), key=lambda k: profile[k])
mf = 0
for k, (n, w, r, m) in enumerate(ins):
    mf += m
    if k % args.every == 0 or k == peak:
        print(f"line {n:6d}  instr {k:6d}  mfma {mf:4d}  live vgprs {profile[k]:4d}{'   <== peak' if k == peak else ''}")

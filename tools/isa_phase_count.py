"""Development aid: instruction mix of each kernel phase, read off the gfx950 ISA.

Compiles one kernel instance with -DLSCQP_PHASE_TIMING (the s_memtime markers of LSCQP_T delimit the phases) and
counts instruction classes between consecutive markers.  usage: isa_phase_count.py [M DIM NSLOT] [extra -D flags]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lsc_dr_planner_amd", "csrc")
nums = [a for a in sys.argv[1:] if not a.startswith("-")]
flags = [a for a in sys.argv[1:] if a.startswith("-")]
M, D, NSLOT = [int(v) for v in (nums + ["5", "3", "10"][len(nums):])]
os.makedirs("/tmp/asm", exist_ok=True)
open("/tmp/asm/tu.hip", "w").write(
    '#define LSCQP_M %d\n#define LSCQP_DIM %d\n#define LSCQP_ES 1\n#define LSCQP_NSLOT %d\n#include "lscqp_inst.hip"\n' % (M, D, NSLOT))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-disable-promote-alloca-to-vector", "-DLSCQP_PHASE_TIMING",
                       "-I", SRC, "-S", "--cuda-device-only", "/tmp/asm/tu.hip", "-o", "/tmp/asm/tu_t.s"] + flags, stderr=subprocess.DEVNULL)
lines = open("/tmp/asm/tu_t.s").read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN5lscqp17lscqp_pdip_kernel") and ":" in l][0]
end = [i for i, l in enumerate(lines) if "s_endpgm" in l and i > start][-1]


def cat(op):
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "readlane"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if re.match(r"v_(fma|mul|add|max|min|rcp|div|fmac|cmp\w*)_f64", op) or op.startswith("v_pk"):
        return "f64"
    if op.startswith("v_mov"):
        return "vmov"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("buffer_"):
        return "vmem"
    return "other"


seg, cur = [], collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.split()[0].endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_memtime":
        seg.append(cur)
        cur = collections.Counter()
        continue
    cur[cat(op)] += 1
seg.append(cur)
keys = ["f64", "acc", "scratch", "ds", "readlane", "cndmask", "vmov", "valu_other", "salu", "nop", "wait", "vmem", "other"]
print("seg " + " ".join("%8s" % k for k in keys) + "    total")
for i, c in enumerate(seg):
    print("%3d " % i + " ".join("%8d" % c[k] for k in keys) + "   %6d" % sum(c.values()))
for l in lines[end:]:
    if re.search(r"ScratchSize|vgpr_spill|\.vgpr_count|\.agpr_count|NumVgprs|NumAgprs|Occupancy", l):
        print(l.strip())

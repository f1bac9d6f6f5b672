"""Work counters of the compiled PDIP kernel instances, read off the gfx950 ISA at build time.

SURVEY.md section 8d asks for the fp64-VALU figure next to the HBM one: "measured iterations x flops/iteration from the kernel's own
counters".  The iterations come from the kernel (lscqp_info.iterations, per instance); the flops of ONE iteration of a given kernel
instance are a property of its machine code, so they are counted there instead of being derived by hand: the device code object is
unbundled from the instance's object file, disassembled, the iteration body is located (between the position markers the kernel source
plants: LSCQP_MARK) and its fp64 vector instructions are counted -- v_fma/v_fmac_f64 and v_pk_fma as 2 flops per lane, every other fp64 arithmetic
instruction as 1.  Per wavefront that is x 64 lanes; per instance x the wavefronts of its workgroup.

What the number is: the fp64 work the VALUs EXECUTE for one pass through the loop body (masked lanes included -- the SIMD is busy
either way -- and the few rarely-taken blocks of the body counted as if taken: the re-centring of a jammed start and the objective
evaluation near convergence, < 7 % of the body), i.e. the quantity MI355X's 78.6 TFLOP/s fp64 vector peak is about.  It is NOT the
algorithmic flop count of an interior-point iteration (which is lower: the register LDL^T updates full rows, not triangles).
"""
import os
import re
import subprocess

LLVM = os.environ.get("LSCQP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_F64_ARITH = re.compile(r"^v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|div_fmas|div_fixup|div_scale|ldexp|trig_preop|frexp_mant|floor|ceil|trunc|rndne|fract)_f64")
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>")
_SYM = re.compile(r"^([0-9a-fA-F]+) <(\S+)>:")


def disassemble(obj_path, workdir):
    fat, co = os.path.join(workdir, "fat.bin"), os.path.join(workdir, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def count(asm_text):
    """-> dict for the FIRST lscqp_pdip_kernel of the object (the one-instance-per-workgroup form; an instance with a persistent form
    carries that second, same body): per-lane counts inside / outside the iteration loop."""
    start, ins = None, []
    for l in asm_text.split("\n"):
        s = _SYM.match(l)
        if s:
            if start is not None:
                break
            if "lscqp_pdip_kernel" in s.group(2):
                start = int(s.group(1), 16)
            continue
        if start is None:
            continue
        m = _INS.match(l)
        if not m:
            continue
        tgt = _TARGET.search(l)
        ins.append((int(m.group(3), 16), m.group(1), start + int(tgt.group(1), 16) if tgt else None))
    if not ins:
        raise RuntimeError("no lscqp_pdip_kernel in the disassembly")
    # the iteration body lies between the kernel's position markers (LSCQP_MARK in lscqp_kernel.hpp): s_nop 13 top of the body,
    # s_nop 12 behind the convergence test, s_nop 14 end of the body.  (Cold blocks the compiler moved out of line -- the
    # re-centring of a jammed start, exits -- fall outside the range, which is what a per-iteration count wants.)
    mark = {}
    for l in asm_text.split("\n"):
        m = re.match(r"^\s+s_nop (12|13|14)\s*//\s*([0-9A-Fa-f]+):", l)
        if m and ins[0][0] <= int(m.group(2), 16) <= ins[-1][0]:  # (an instance may carry two forms of the kernel: the first one counts)
            mark.setdefault(int(m.group(1)), []).append(int(m.group(2), 16))
    if sorted(mark) != [12, 13, 14] or any(len(v) != 1 for v in mark.values()):
        raise RuntimeError("iteration markers not found exactly once each: %r" % mark)
    top, conv, end = mark[13][0], mark[12][0], mark[14][0]
    if not (top < conv < end):
        raise RuntimeError("iteration markers out of order: %r" % mark)

    # Nested-dissection instances bracket the regions only some wavefronts execute: s_nop 8 = wavefront 0 only, 9 = wavefront 1 only,
    # 10 = wavefronts 0 and 1, s_nop 11 = end.  `share[addr]` = number of wavefronts that run the instruction (None: all of them).
    share, cur = {}, None
    for l in asm_text.split("\n"):
        m = _INS.match(l)
        if not m:
            continue
        op, addr = m.group(1), int(m.group(3), 16)
        if not (ins[0][0] <= addr <= ins[-1][0]):
            continue
        if op == "s_nop":
            k = re.match(r"^\s+s_nop (\d+)", l)
            k = int(k.group(1)) if k else -1
            if k in (8, 9):
                cur = 1
            elif k == 10:
                cur = 2
            elif k == 11:
                cur = None
            continue
        if cur is not None:
            share[addr] = cur

    def tally(sel):
        """per-wavefront instruction counts AVERAGED over the workgroup's wavefronts: an instruction only `k` of the W wavefronts run counts k / W
        (weights resolved by the caller through `waves`)."""
        t = {"fma_f64": 0.0, "other_f64": 0.0, "valu": 0.0, "lds": 0.0, "partial_fma_f64": 0.0, "partial_other_f64": 0.0, "partial_valu": 0.0,
             "partial_lds": 0.0}
        for addr, op, _ in ins:
            if not sel(addr):
                continue
            k = share.get(addr)
            pre = "" if k is None else "partial_"
            wgt = 1.0 if k is None else float(k)  # partial_*: summed over the wavefronts that run it (1 or 2), not per wavefront
            if op.startswith("v_"):
                t[pre + "valu"] += wgt
                if _F64_ARITH.match(op):
                    if op.startswith(("v_fma_f64", "v_fmac_f64")):
                        t[pre + "fma_f64"] += wgt
                    else:
                        t[pre + "other_f64"] += wgt
            elif op.startswith("ds_"):
                t[pre + "lds"] += wgt
        return t

    body = tally(lambda a: top <= a <= end)
    head = tally(lambda a: top <= a <= conv)  # residual pass + convergence test: what the final, partial pass through the body executes
    rest = tally(lambda a: not (top <= a <= end))  # prologue + epilogue (+ out-of-line blocks)
    out = {"body_bytes": end - top, "code_bytes": ins[-1][0] - ins[0][0]}
    for name, t in (("iter", body), ("last", head), ("fixed", rest)):
        for k, v in t.items():
            out["%s_%s" % (name, k)] = v
    return out


def of_object(obj_path, workdir):
    os.makedirs(workdir, exist_ok=True)
    return count(disassemble(obj_path, workdir))


if __name__ == "__main__":
    import json
    import sys
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        for p in sys.argv[1:]:
            print(os.path.basename(p), json.dumps(of_object(p, td)))

"""Build liblscqp.so (the C-ABI library of include/lscqp.h) for gfx950 with hipcc, in-tree.

One translation unit per kernel instance, compiled in parallel; hipcc cross-compiles without a GPU.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
_AB = os.environ.get("LSCQP_AB", "")  # development: LSCQP_AB=<name> builds liblscqp_<name>.so from its own object directory
OBJ = os.path.join(HERE, "csrc", "_obj" + ("_" + _AB if _AB else ""))
LIB = os.path.join(HERE, "liblscqp%s.so" % ("_" + _AB if _AB else ""))
SYNC_LIB = os.path.join(HERE, "liblscqp%s_sync.so" % ("_" + _AB if _AB else ""))  # the race test's twin (csrc/lscqp_das.hip: LSCQP_DAS_FULL_SYNC)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-promote-alloca-to-vector: the kernel keeps its per-lane row state in small arrays indexed by fully unrolled
# loops.  AMDGPUPromoteAlloca turns them into 512/1024-bit vector registers BEFORE the loops are unrolled and SROA could
# split them into scalars; every single-element access then moves a whole 16/32-register tuple between AGPRs and VGPRs
# (measured on MI355X: 0.192 -> 0.158 ms per 64-QP batch, 0.875 -> 0.671 ms per 4096-QP batch, scratch 336 -> 0 B/lane).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-disable-promote-alloca-to-vector",
         "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + os.environ.get("LSCQP_EXTRA_FLAGS", "").split()


def instances():
    txt = open(os.path.join(CSRC, "lscqp_launch.hpp")).read()
    body = txt[txt.index("#define LSCQP_INSTANCES"):]
    return [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", body)]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stderr


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in ("lscqp_kernel.hpp", "lscqp_launch.hpp", "lscqp_staging.hpp")] + [
        os.path.join(HERE, "..", "include", "lscqp.h"), os.path.abspath(__file__)]
    tasks = []
    objs = []
    for (M, D, E, S, W, X) in instances():
        o = os.path.join(OBJ, "inst_%d_%d_%d_%d_%d_%d.o" % (M, D, E, S, W, X))
        objs.append(o)
        src = os.path.join(CSRC, "lscqp_inst.hip")
        if force or _newer(o, hdrs + [src]):
            extra = os.environ.get("LSCQP_EXTRA_MIXED_FLAGS", "").split() if X else os.environ.get("LSCQP_EXTRA_F64_FLAGS", "").split()
            tasks.append([HIPCC] + FLAGS + extra + ["-DLSCQP_M=%d" % M, "-DLSCQP_DIM=%d" % D, "-DLSCQP_ES=%d" % E, "-DLSCQP_NSLOT=%d" % S, "-DLSCQP_W=%d" % W, "-DLSCQP_MIXED=%d" % X, "-c", src, "-o", o])
    api_o = os.path.join(OBJ, "api.o")
    objs.append(api_o)
    api_src = os.path.join(CSRC, "lscqp_api.hip")
    if force or _newer(api_o, hdrs + [api_src]):
        tasks.append([HIPCC] + FLAGS + os.environ.get("LSCQP_EXTRA_F64_FLAGS", "").split() + ["-c", api_src, "-o", api_o])
    post_o = os.path.join(OBJ, "lscpost.o")
    objs.append(post_o)
    post_src = os.path.join(CSRC, "lscpost.hip")
    if force or _newer(post_o, hdrs + [post_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", post_src, "-o", post_o])
    goal_o = os.path.join(OBJ, "lscgoal.o")
    objs.append(goal_o)
    goal_src = os.path.join(CSRC, "lscgoal.hip")
    if force or _newer(goal_o, hdrs + [goal_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", goal_src, "-o", goal_o])
    sfc_o = os.path.join(OBJ, "lscsfc.o")
    objs.append(sfc_o)
    sfc_src = os.path.join(CSRC, "lscsfc.hip")
    if force or _newer(sfc_o, hdrs + [sfc_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", sfc_src, "-o", sfc_o])
    sfct_o = os.path.join(OBJ, "lscsfc_tp.o")
    objs.append(sfct_o)
    sfct_src = os.path.join(CSRC, "lscsfc_tp.hip")
    if force or _newer(sfct_o, hdrs + [sfct_src, sfc_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", sfct_src, "-o", sfct_o])
    comm_o = os.path.join(OBJ, "lscqp_comm.o")
    objs.append(comm_o)
    comm_src = os.path.join(CSRC, "lscqp_comm.hip")
    if force or _newer(comm_o, hdrs + [comm_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", comm_src, "-o", comm_o])
    plan_o = os.path.join(OBJ, "lscplan.o")
    objs.append(plan_o)
    plan_src = os.path.join(CSRC, "lscplan.hip")
    if force or _newer(plan_o, hdrs + [plan_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", plan_src, "-o", plan_o])
    gen_o = os.path.join(OBJ, "lscgen.o")
    objs.append(gen_o)
    gen_src = os.path.join(CSRC, "lscgen.hip")
    if force or _newer(gen_o, hdrs + [gen_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", gen_src, "-o", gen_o])
    gen_o = os.path.join(OBJ, "lscqp_generic.o")
    objs.append(gen_o)
    gen_src = os.path.join(CSRC, "lscqp_generic.hip")
    if force or _newer(gen_o, hdrs + [gen_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", gen_src, "-o", gen_o])
    das_o = os.path.join(OBJ, "lscqp_das.o")
    objs.append(das_o)
    das_src = os.path.join(CSRC, "lscqp_das.hip")
    if force or _newer(das_o, hdrs + [das_src]):
        # -ffp-contract=on: a multiply-add is fused where the SOURCE writes a * b + c in one expression and nowhere else.  The default (fast) lets
        # the backend fuse across statements as the surrounding code happens to allow -- the kernel's instantiations (row formats, wavefronts
        # per QP, launch forms) then differ in the last bit, and the phase's results are required to be identical across all of them
        tasks.append([HIPCC] + FLAGS + ["-ffp-contract=on", "-c", das_src, "-o", das_o])
    # the race test's twin of the dual active-set kernel (csrc/lscqp_das.hip: LSCQP_DAS_FULL_SYNC) -> liblscqp_sync.so, linked below from the
    # product's own objects with this one in place of lscqp_das.o (tests/test_race_twin.py)
    das_sync_o = os.path.join(OBJ, "lscqp_das_sync.o")
    if force or _newer(das_sync_o, hdrs + [das_src]):
        tasks.append([HIPCC] + FLAGS + ["-ffp-contract=on", "-DLSCQP_DAS_FULL_SYNC", "-c", das_src, "-o", das_sync_o])
    diag_o = os.path.join(OBJ, "lscqp_diag.o")
    objs.append(diag_o)
    diag_src = os.path.join(CSRC, "lscqp_diag.hip")
    if force or _newer(diag_o, hdrs + [diag_src]):
        tasks.append([HIPCC] + FLAGS + ["-c", diag_src, "-o", diag_o])
    if tasks:
        with ThreadPoolExecutor(max_workers=jobs or os.cpu_count() or 4) as ex:
            for msg in ex.map(_run, tasks):
                if verbose and msg:
                    sys.stderr.write(msg)
    # work counters of every instance, read off its machine code (isa_work.py) -> one small generated host TU
    work_o = os.path.join(OBJ, "lscqp_work_table.o")
    objs.append(work_o)
    inst_objs = [o for o in objs if os.path.basename(o).startswith("inst_")]
    if force or _newer(work_o, inst_objs + [os.path.join(HERE, "isa_work.py"), os.path.abspath(__file__)]):
        _work_table(inst_objs, work_o, jobs)
        tasks.append("work table")
    if tasks or not os.path.exists(LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"])
    if tasks or not os.path.exists(SYNC_LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SYNC_LIB] + [das_sync_o if o == das_o else o for o in objs] + ["-ldl", "-lpthread"])
    return LIB


def _work_table(inst_objs, work_o, jobs=None):
    import json
    import tempfile

    sys.path.insert(0, HERE)
    import isa_work

    def one(o):
        # reporting only (lscqp_instance_work): an instance whose machine code cannot be read (another ROCm layout, markers moved
        # or duplicated by the compiler) gets NO row -- lscqp_work_table_ then returns 1 and the API answers UNSUPPORTED for it --
        # and never stops the library from linking
        js = o[:-2] + ".work.json"
        try:
            if not _newer(js, [o, os.path.join(HERE, "isa_work.py")]):
                return json.load(open(js))
            with tempfile.TemporaryDirectory() as td:
                w = isa_work.of_object(o, td)
            json.dump(w, open(js, "w"))
            return w
        except Exception as ex:  # noqa: BLE001
            sys.stderr.write("build.py: warning: no work counters for %s (%s: %s)\n" % (os.path.basename(o), type(ex).__name__, str(ex)[:200]))
            return None

    with ThreadPoolExecutor(max_workers=jobs or os.cpu_count() or 4) as ex:
        works = list(ex.map(one, inst_objs))
    src = os.path.join(OBJ, "lscqp_work_table.cpp")
    with open(src, "w") as f:
        f.write("// generated by lsc_dr_planner_amd/build.py from the instances' machine code (isa_work.py); per wavefront\n")
        f.write("struct Row { int M, D, E, S, W, X; double t[24]; };\nstatic const Row kRows[] = {\n")
        f.write("    {-1, -1, -1, -1, -1, -1, {0}},\n")  # (keeps the array non-empty when no instance could be read)
        for o, w in zip(inst_objs, works):
            if w is None:
                continue
            key = os.path.basename(o)[5:-2].split("_")
            # per section: instructions every wavefront runs (fma, other fp64, valu, lds), then the ones only SOME wavefronts run, summed
            # over those wavefronts (nested-dissection instances)
            vals = [w["%s_%s" % (a, b)] for a in ("iter", "last", "fixed")
                    for b in ("fma_f64", "other_f64", "valu", "lds", "partial_fma_f64", "partial_other_f64", "partial_valu", "partial_lds")]
            f.write("    {%s, {%s}},\n" % (", ".join(key), ", ".join(str(v) for v in vals)))
        f.write("};\nextern \"C\" int lscqp_work_table_(int M, int D, int E, int S, int W, int X, double* out24) {\n"
                "    for (const Row& r : kRows)\n        if (r.M == M && r.D == D && r.E == E && r.S == S && r.W == W && r.X == X) {\n"
                "            for (int i = 0; i < 24; i++) out24[i] = r.t[i];\n            return 0;\n        }\n    return 1;\n}\n")
    _run(["g++", "-O1", "-fPIC", "-c", src, "-o", work_o])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

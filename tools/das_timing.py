"""Development aid: per-phase cycle breakdown of the dual active-set kernel (csrc/lscqp_das.hip built with -DLSCQP_DAS_TIMING into
liblscqp_dastime.so, linked against the objects of the product build) on the bench's batches.

usage: python tools/das_timing.py --build-only   (here)      python tools/das_timing.py [c1 c0 c3s ...]   (GPU box; LSCQP_LIB is set by the tool)"""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lsc_dr_planner_amd", "csrc")
OUT = os.path.join(ROOT, "lsc_dr_planner_amd", "liblscqp_dastime.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-disable-promote-alloca-to-vector", "-ffp-contract=on"]

if "--build-only" in sys.argv:
    o = "/tmp/lscqp_das_timing.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DLSCQP_DAS_TIMING=0xffff", "-DLSCQP_DAS_TIMING_MIN_STEPS=%s" % os.environ.get("DAS_TIMING_MIN_STEPS", "0"), "-c", os.path.join(CSRC, "lscqp_das.hip"), "-o", o])
    objs = [f for f in glob.glob(os.path.join(CSRC, "_obj", "*.o")) if os.path.basename(f) != "lscqp_das.o"] + [o]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-lpthread"])
    print(OUT)
    sys.exit(0)

os.environ["LSCQP_LIB"] = os.environ.get("DAS_TIMING_LIB", OUT)
import torch  # noqa: E402

import bench  # noqa: E402
from lsc_dr_planner_amd import api, synth  # noqa: E402

L = api.lib()
NAMES = ["header/boxes/offset", "intervals + c_u", "first pass: reduction", "later passes: reduction", "verification", "candidate (decode, table, w_p)", "partial steps", "epilogue", "passes: LSC rows", "passes: two-sided rows",
         "step: decision (wavefront 0)", "step: c and W", "step: a leaving row", "decision: v = A'w_p", "decision: r = S^-1 v", "decision: sums, lengths, argmin"]
dev = torch.device("cuda", 0)
for key in [a for a in sys.argv[1:] if not a.startswith("-")] or ["c1", "c0", "c2", "c3s", "c4_f64"]:
    cfg = bench.CONFIGS[key]
    N, M, dim = cfg["agents"], cfg["segments"], cfg["dim"]
    sw, sol, build, (hdr, rows, off, sfc) = bench.make_batch(api, synth, lambda s: api.Solver(api.make_desc(M=M, dim=dim, world_min=s.world_min, world_max=s.world_max)),
                                                             N, M, dim, cfg["obs"], seed=cfg["seed"], style=cfg["style"], warm_steps=cfg.get("warm_steps", 3))
    t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
    d_xi = torch.from_numpy(np.ascontiguousarray(api.x_init_from_swarm(build, dim))).to(dev)
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    d_info = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    cyc = (C.c_ulonglong * 16)()
    for _ in range(3):
        sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
    torch.cuda.synchronize()
    L.lscqp_das_cycles(cyc, 1)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
    e1.record()
    torch.cuda.synchronize()
    L.lscqp_das_cycles(cyc, 0)
    info = d_info.cpu().numpy().view(api.INFO_DTYPE)
    nbook = int(os.environ.get("DAS_TIMING_MIN_STEPS", "0"))
    Nb = max(1, int((info["iterations"] >= nbook).sum()))  # (DAS_TIMING_DIV: a build with LSCQP_DAS_TIMING_MIN_STEPS books those instances only)
    c = np.array(list(cyc)[:16], dtype=float) / reps / Nb
    print("%s: %d QPs, steps mean %.2f max %d, %.1f us per call | cycles per QP (thread 0 of each workgroup, mean over the batch): total %.0f" % (
        key, N, info["iterations"].mean(), info["iterations"].max(), e0.elapsed_time(e1) / reps * 1e3, c.sum()))
    for n_, v in zip(NAMES, c):
        print("      %-34s %9.0f" % (n_, v))

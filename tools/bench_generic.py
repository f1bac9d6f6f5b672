"""Development aid: the run-time-shaped kernel (csrc/lscqp_generic.hip) against the compiled instances, HIP-event timed.
usage: python tools/bench_generic.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import api, synth  # noqa: E402


def up(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def run(N, M, dim, n_obs, mode, force):
    dev = torch.device("cuda", 0)
    os.environ.pop("LSCQP_FORCE_GENERIC", None)
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=4000 + N + M)
    sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=mode, world_min=sw.world_min, world_max=sw.world_max))
    for _ in range(2):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        r = sol.solve_host(hdr, rows, off, sfc, x_init=api.x_init_from_swarm(b, dim))
        r["x"][r["status"] != 0] = api.x_init_from_swarm(b, dim)[r["status"] != 0]
        sw.advance(r["x"])
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    dh, dr, do_, ds, dxi = up(hdr, dev), up(rows, dev), up(off, dev), up(sfc, dev), up(api.x_init_from_swarm(b, dim), dev)
    dx = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    dob = torch.zeros(N, dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    if force:
        os.environ["LSCQP_FORCE_GENERIC"] = "1"
    call = lambda: sol.solve_device(N, sw.n_obs, dh, dr, do_, ds, dx, dob, dst, dinfo, d_x_init=dxi)  # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    it = dinfo.cpu().numpy().view(api.INFO_DTYPE)["iterations"]
    os.environ.pop("LSCQP_FORCE_GENERIC", None)
    return e0.elapsed_time(e1) / 10, float(it.mean()), int(it.max()), int((dst.cpu().numpy() != 0).sum())


def phases():
    """--phases: per-phase cycles of the run-time-shaped kernel (a library built with LSCQP_EXTRA_FLAGS=-DLSCQP_GEN_TIMING, LSCQP_LIB=...)."""
    import ctypes as C

    L = api.lib()
    names = ["prologue / loop top", "pass 1", "gradient + tests", "assembly", "LDL^T", "predictor solve + expand", "pass 2", "corrector solve + expand",
             "pass 3 + step length", "update"]
    for N, M, dim, n_obs, mode in ((64, 5, 3, 20, api.PLANNER_LSC), (64, 10, 3, 20, api.PLANNER_DLSC)):
        buf = (C.c_ulonglong * 16)()
        L.lscqp_generic_cycles(buf, 1)
        ms, itm, itx, bad = run(N, M, dim, n_obs, mode, True)
        L.lscqp_generic_cycles(buf, 1)
        launches = 13.0  # run(): 3 warm + 10 timed launches of the same batch
        nit = itm * N * launches
        print("%d x M%d dim %d x %d: %.3f ms per launch, %.2f iterations" % (N, M, dim, n_obs, ms, itm))
        # GEN_T(k) closes the phase that ENDS at marker k: slot 0 = what precedes pass 1 (prologue on the first pass, else nothing), slot k = phase k-1's successor
        for k in range(10):
            print("   %-28s %9.0f cycles per iteration and QP" % (names[k], buf[(k + 1) % 10 if k else 0] / nit if False else buf[k] / nit))


if __name__ == "__main__":
    if "--phases" in sys.argv:
        phases()
        sys.exit(0)
    for N, M, dim, n_obs, mode, name in ((64, 5, 3, 20, api.PLANNER_LSC, "lsc"), (1024, 5, 3, 20, api.PLANNER_LSC, "lsc"), (64, 10, 3, 20, api.PLANNER_LSC, "lsc"),
                                         (64, 10, 3, 20, api.PLANNER_DLSC, "dlsc"), (512, 10, 3, 20, api.PLANNER_DLSC, "dlsc"), (64, 10, 2, 9, api.PLANNER_LSC, "lsc"),
                                         (64, 6, 3, 20, api.PLANNER_DLSC, "dlsc"), (64, 9, 3, 20, api.PLANNER_LSC, "lsc")):
        res = {}
        for force in (False, True):
            try:
                res[force] = run(N, M, dim, n_obs, mode, force)
            except Exception as ex:
                res[force] = str(ex)[:60]
        print("%5d x M%-2d dim %d x %2d %-4s   as selected: %s   forced generic: %s" % (N, M, dim, n_obs, name, res[False], res[True]))

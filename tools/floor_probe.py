"""Development aid: per-iteration trace of the stopping quantities for the instances of a bench config that end with
LSCQP_INFO_FLOOR_ACCEPTED (or any other flag / status of interest).  Builds ONE instrumented instance (-DLSCQP_TRACE) as its own
library, runs bench.py's batch of the config through it and prints the trajectories.

    python tools/floor_probe.py c3 [-DLSCQP_...]        (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "lsc_dr_planner_amd", "csrc")
key = ([a for a in sys.argv[1:] if not a.startswith("-")] + ["c3"])[0]
xflags = [a for a in sys.argv[1:] if a.startswith("-D")]
import bench  # noqa: E402

cfg = bench.CONFIGS[key]
N, M, D, NOBS = int(os.environ.get("PROBE_N", 1 if os.environ.get("PROBE_NPZ") else cfg["agents"])), cfg["segments"], cfg["dim"], cfg["obs"]
NSLOT, W = {(10, 3): (10, 4), (5, 3): (10, 1), (6, 3): (7, 2), (10, 2): (5, 2)}[(M, D)]
if os.environ.get("PROBE_NSLOT"):
    NSLOT, W = int(os.environ["PROBE_NSLOT"]), int(os.environ["PROBE_W"])
OUT = "/tmp/liblscqp_trace.so"
drv = r'''
extern "C" int lscqp_trace_read(double* out, int nq) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lscqp::lscqp_dbg_trace), sizeof(double) * 64 * 12 * nq);
}
'''
ES = int(os.environ.get("PROBE_ES", "1"))  # 0: the end-stop-free class (DLSC / BVC / RSFC planner modes)
tu = ('#define LSCQP_M %d\n#define LSCQP_DIM %d\n#define LSCQP_ES %d\n#define LSCQP_NSLOT %d\n#define LSCQP_W %d\n#define LSCQP_MIXED 0\n#include "lscqp_inst.hip"\n'
      % (M, D, ES, NSLOT, W)) + drv
open("/tmp/trace_tu.hip", "w").write(tu)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-mllvm",
                       "-disable-promote-alloca-to-vector", "-DLSCQP_TRACE", "-DLSCQP_TRACE_Q=%d" % N, "-I", SRC, "/tmp/trace_tu.hip", "-o", OUT] + xflags,
                      stderr=subprocess.DEVNULL)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402

from lsc_dr_planner_amd import api, synth  # noqa: E402

L = C.CDLL(OUT)


def factory(sw):
    return api.Solver(api.make_desc(M=M, dim=D, world_min=sw.world_min, world_max=sw.world_max))


if os.environ.get("PROBE_NPZ"):
    # PROBE_NPZ=file : ONE dumped instance (hdr[1], rows[], sfc[1, M], world_min, world_max in the ABI's dtypes), cold start
    class _W:  # what the rest of the script reads off a swarm
        pass
    dmp = np.load(os.environ["PROBE_NPZ"])
    sw = _W()
    sw.world_min, sw.world_max, sw.n_obs = dmp["world_min"], dmp["world_max"], int(dmp["hdr"]["n_obs"][0])
    N, b = 1, None
    sol = api.Solver(api.make_desc(M=M, dim=D, planner_mode=api.PLANNER_LSC if ES else api.PLANNER_DLSC, world_min=sw.world_min, world_max=sw.world_max))
    hdr, rows, sfc = dmp["hdr"], dmp["rows"], dmp["sfc"]
    off = np.array([0, len(rows)], dtype=np.uint64)
    os.environ["WARM"] = "0"
elif os.environ.get("PROBE_SWARM"):
    # PROBE_SWARM=seed,steps : synth.Swarm(N, seed=seed) of the config's shape advanced `steps` replans (failed QPs keep their start), as the
    # parity tests build their batches; PROBE_N overrides the agent count
    seed_, steps_ = [int(v) for v in os.environ["PROBE_SWARM"].split(",")]
    sw = synth.Swarm(N, M=M, dim=D, n_obs=NOBS, seed=seed_, style=cfg["style"])
    sol = factory(sw)
    for _ in range(steps_):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        sw.advance(sol.solve_host(hdr, rows, off, sfc)["x"])
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
else:
    sw, sol, b, (hdr, rows, off, sfc) = bench.make_batch(api, synth, factory, N, M, D, NOBS, seed=cfg["seed"], style=cfg["style"], warm_steps=3)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
dx = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
dob = torch.zeros(N, dtype=torch.float64, device=dev)
dst = torch.zeros(N, dtype=torch.int32, device=dev)
dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)


class DevClass(C.Structure):
    _fields_ = [("dt", C.c_double), ("w_c", C.c_double), ("w_t", C.c_double), ("comm_range", C.c_double), ("world_min", C.c_double * 3),
                ("world_max", C.c_double * 3), ("q2s", C.c_double), ("dQ", C.c_double * 36), ("tol", C.c_double), ("max_iter", C.c_int),
                ("use_sfc", C.c_int), ("n_obs_max", C.c_int), ("rows_f32", C.c_int), ("rsfc", C.c_int), ("repair", C.c_int),
                ("warm_mu0", C.c_double), ("warm_s0", C.c_double), ("warm_net", C.c_double), ("order", C.c_void_p), ("queue", C.c_void_p)]


cls = DevClass()
cls.dt, cls.w_c, cls.w_t, cls.comm_range = 0.2, 0.01, 1.0, 3.0
for k in range(3):
    cls.world_min[k], cls.world_max[k] = sw.world_min[k], sw.world_max[k]
cls.q2s = 2 * 0.01 * 0.2 ** -5
cls.tol, cls.max_iter, cls.use_sfc, cls.n_obs_max = 1e-10, 60, 1, sw.n_obs
cls.warm_mu0, cls.warm_s0, cls.warm_net = 1e-3, 0.03, 0.0
d_xi = torch.from_numpy(api.x_init_from_swarm(b, D)).to(dev) if os.environ.get("WARM", "1") != "0" else None
fn = getattr(L, "lscqp_launch_%d_%d_%d_%d_%d_0" % (M, D, ES, NSLOT, W))
fn.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10
rc = fn(C.byref(cls), N, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), d_xi.data_ptr() if d_xi is not None else None,
        dx.data_ptr(), dob.data_ptr(), dst.data_ptr(), dinfo.data_ptr(), None)
assert rc == 0, rc
torch.cuda.synchronize()
tr = np.zeros((N, 64, 12))
L.lscqp_trace_read.argtypes = [C.c_void_p, C.c_int]
assert L.lscqp_trace_read(tr.ctypes.data, N) == 0
info = dinfo.cpu().numpy().view(api.INFO_DTYPE)
st = dst.cpu().numpy()
fl = np.where((info["flags"] & api.INFO_FLOOR_ACCEPTED) != 0)[0]
print("config %s: %d instances, status histogram %s, floor accepted %d: %s" % (key, N, np.bincount(st).tolist(), len(fl), fl.tolist()))
print("iterations: mean %.3f max %d; iterations with a shifted refactorisation (slot 11): %d in %d instances" % (
    info["iterations"].mean(), info["iterations"].max(), int((tr[:, :, 11] > 0).sum()), int(((tr[:, :, 11] > 0).sum(axis=1) > 0).sum())))
bad = np.where(st != 0)[0]
show = bad if len(bad) else (fl if len(fl) else np.argsort(-info["iterations"])[:3])
names = {0: "", 1: "OPTIMAL", 2: "PIVOT", 3: "STALL", 4: "INFEAS"}
for q in show[:int(os.environ.get("PROBE_SHOW", "12"))]:
    print("--- instance %d: iterations %d, res_dual %.2e gap %.2e" % (q, info["iterations"][q], info["res_dual"][q], info["gap"][q]))
    for it in range(min(int(info["iterations"][q]) + 1, 64)):
        r = tr[q, it]
        print("   it %2d  rp %.1e  rd %.2e  gap* %.2e  mu %.1e  alpha %.4f sigma %.1e  gls %.1e wmax %.1e lmax %.1e |dz| %.1e %s%s" % (
            it, r[0], r[1], r[2], r[3], r[4], r[5], r[7], r[8], r[9], r[10], names.get(int(r[6]), "?"), " SHIFTED" if r[11] else ""))

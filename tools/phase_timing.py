"""Development aid: per-phase cycle breakdown of the PDIP kernel (build with -DLSCQP_PHASE_TIMING)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.environ.get("LSCQP_SRC", os.path.join(ROOT, "lsc_dr_planner_amd", "csrc"))
OUT = "/tmp/liblscqp_prof.so"
args = [a for a in sys.argv[1:] if not a.startswith("-")]
xflags = [a for a in sys.argv[1:] if a.startswith("-D")]
M, D, N, NOBS, NSLOT, W = [int(v) for v in (args + ["5", "3", "64", "20", "10", "1"][len(args):])]
MIXED = 1 if "--mixed" in sys.argv else 0  # the float32-factorisation instance (LSCQP_PRECISION_MIXED) instead of the fp64 one
drv = r'''
#include "lscqp_kernel.hpp"
extern "C" int lscqp_dbg_read(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(lscqp::lscqp_dbg_cycles), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(lscqp::lscqp_dbg_cycles), z, sizeof z); }
    return 0;
}
'''
if not os.path.exists(OUT) or "--rebuild" in sys.argv or True:
    open("/tmp/prof_drv.hip", "w").write(drv)
    # single TU so that the __device__ symbol is shared: include the instance + api sources
    tu = '#define LSCQP_M %d\n#define LSCQP_DIM %d\n#define LSCQP_ES 1\n#define LSCQP_NSLOT %d\n#define LSCQP_W %d\n#define LSCQP_MIXED %d\n#include "lscqp_inst.hip"\n' % (M, D, NSLOT, W, MIXED) + drv
    open(os.path.join("/tmp", "prof_tu.hip"), "w").write(tu)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-mllvm", "-disable-promote-alloca-to-vector",
                           "-DLSCQP_PHASE_TIMING", "-I", SRC, "/tmp/prof_tu.hip", "-o", OUT] + xflags, stderr=subprocess.DEVNULL)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
from lsc_dr_planner_amd import api, synth
L = C.CDLL(OUT)
sw = synth.Swarm(N, M=M, dim=D, n_obs=NOBS, seed=1)
sol = api.Solver(api.make_desc(M=M, dim=D, world_min=sw.world_min, world_max=sw.world_max))
for _ in range(3):
    b = sw.build(); hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    r = sol.solve_host(hdr, rows, off, sfc); sw.advance(r["x"])
b = sw.build(); hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
dx = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev); dob = torch.zeros(N, dtype=torch.float64, device=dev)
dst = torch.zeros(N, dtype=torch.int32, device=dev); dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
# call the instrumented launcher directly with the same DevClass the product would build: reuse product handle fields
class DevClass(C.Structure):
    _fields_ = [("dt", C.c_double), ("w_c", C.c_double), ("w_t", C.c_double), ("comm_range", C.c_double), ("world_min", C.c_double * 3),
                ("world_max", C.c_double * 3), ("q2s", C.c_double), ("dQ", C.c_double * 36), ("tol", C.c_double), ("max_iter", C.c_int),
                ("use_sfc", C.c_int), ("n_obs_max", C.c_int), ("rows_f32", C.c_int), ("rsfc", C.c_int), ("repair", C.c_int),
                ("warm_mu0", C.c_double), ("warm_s0", C.c_double), ("warm_net", C.c_double), ("order", C.c_void_p), ("queue", C.c_void_p)]
kq = [720,-1800,1200,0,0,-120,-1800,4800,-3600,0,600,0,1200,-3600,3600,-1200,0,0,0,0,-1200,3600,-3600,1200,0,600,0,-3600,4800,-1800,-120,0,0,1200,-1800,720]
cls = DevClass(); cls.dt = 0.2; cls.w_c = 0.01; cls.w_t = 1.0; cls.comm_range = 3.0
for k in range(3): cls.world_min[k] = sw.world_min[k]; cls.world_max[k] = sw.world_max[k]
cls.q2s = 2 * 0.01 * 0.2 ** -5
cls.tol = 1e-10; cls.max_iter = 60; cls.use_sfc = 1; cls.n_obs_max = sw.n_obs
cls.warm_mu0, cls.warm_s0, cls.warm_net = (1e-7, 0.003, 0.6) if os.environ.get("TIGHT") else (1e-3, 0.03, 0.0)
d_xi = torch.from_numpy(api.x_init_from_swarm(b, D)).to(dev) if os.environ.get("WARM", "1") != "0" else None
XINIT = d_xi.data_ptr() if d_xi is not None else None
fn = getattr(L, "lscqp_launch_%d_%d_1_%d_%d_%d" % (M, D, NSLOT, W, MIXED))
fn.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10
def launch():
    rc = fn(C.byref(cls), N, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), XINIT, dx.data_ptr(), dob.data_ptr(), dst.data_ptr(), dinfo.data_ptr(), None)
    assert rc == 0, rc
launch(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)(); L.lscqp_dbg_read(buf, 1)
launch(); torch.cuda.synchronize(); L.lscqp_dbg_read(buf, 1)
it = dinfo.cpu().numpy().view(api.INFO_DTYPE)["iterations"]
names = ["loop-top", "pass1", "grad/conv", "assembly", "factor", "solve1+expand", "pass2", "solve2+expand", "pass3", "update", "epilogue"]
tot = sum(buf[i] for i in range(11)); nit = it.sum()
print("instance <%d,%d,true,%d,%d,%s>  %d QPs x %d neighbours" % (M, D, NSLOT, W, "float" if MIXED else "double", N, sw.n_obs))
print("status: optimal", int((dst.cpu().numpy() == 0).sum()), "of", N, "(the float instance leaves the rest to the fp64 second pass)", "iters total", nit, "mean", it.mean())
for i, nm in enumerate(names):
    print("%-16s %10.0f cycles/iter/QP  %5.1f%%" % (nm, buf[i] / max(nit, 1), 100.0 * buf[i] / tot))
print("total cycles/iter/QP %.0f" % (tot / nit))
if buf[14]:
    print("nested dissection: block phase %.0f cycles/iter/QP (in addition to `factor` above = hand-over + separator)" % (buf[14] / max(nit, 1)))
pro = [buf[i] / N for i in (11, 12, 13, 15)]
print("prologue %.0f cycles/QP (header+cp init %.0f, row staging %.0f, two-sided setup %.0f, start point %.0f), epilogue %.0f cycles/QP, "
      "loop %.0f cycles/QP (%.2f iterations)" % (sum(pro), pro[0], pro[1], pro[2], pro[3], buf[10] / N, (tot - buf[10]) / N, nit / N))

"""Measurement of the kernels either side of the QP: constraint generation in the three planner modes,
safety metrics, voxel-map construction and corridor construction.
    python tools/bench_next_rows.py [N] [--cpu]
Kernel times are HIP-event averages on the launch stream; --cpu also times the OpenMP oracle on the same inputs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lsc_dr_planner_amd import api, synth  # noqa: E402

N = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [4096])[0]
CPU = "--cpu" in sys.argv
M, dim, n_obs = 5, (2 if "--dim2" in sys.argv else 3), 20
dev = torch.device("cuda", 0)
if CPU:
    from oracle import oracle as O


def timed(fn, reps=50, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=1)
init = sw.initial_traj()
nbr = sw.neighbours().astype(np.int32)
goal_all = np.ascontiguousarray(sw.pos + 0.5, dtype=np.float64)
sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
d_traj, d_nbr = up(init), up(nbr)
d_r = torch.full((N,), sw.radius, dtype=torch.float64, device=dev)
d_dw = torch.full((N,), sw.downwash, dtype=torch.float64, device=dev)
d_goal = up(goal_all)
d_rows = torch.zeros(N * n_obs * M * 6 * 4, dtype=torch.float64, device=dev)
nbytes = sol.generate_lsc_bytes(N, n_obs, N)
for name, mode in (("generateLSC", api.GEN_LSC), ("generateCLSC", api.GEN_CLSC), ("generateBVC", api.GEN_BVC)):
    ms = timed(lambda: sol.generate_constraints_device(mode, N, n_obs, 0, d_traj, d_nbr, d_r, d_dw, d_goal, d_rows), reps=200)
    rec = {"kernel": name, "agents": N, "units": N * n_obs * M, "kernel_ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6,
           "hbm_frac": nbytes / (ms * 1e-3) / 8e12}
    if CPU:
        t0 = time.perf_counter()
        O.generate_constraints(mode, init, nbr, sw.radius, sw.downwash, goal_all, dim=dim)
        rec["cpu_oracle_ms_openmp"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(rec))

# safety metrics: all pairs of the N plans, one sample (multisim_time_step = multisim_save_time_step = 0.1 in the launch file)
x_all = np.ascontiguousarray(init.transpose(0, 3, 1, 2)[:, :dim].reshape(N, -1))  # [n][dim][M][6]
hdr = np.zeros(N, api.HEADER_DTYPE)
hdr["vmax"], hdr["amax"] = 1.0, 2.0
d_x, d_hdr = up(x_all), torch.from_numpy(hdr.view(np.uint8).reshape(-1).copy()).to(dev)
d_out = torch.zeros(N * api.SAFETY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
ms = timed(lambda: sol.safety_metrics_device(N, 0, N, 1, 0.1, d_x, d_r, d_dw, d_hdr, d_out), reps=100)
rec = {"kernel": "safety_metrics", "agents": N, "pairs": N * (N - 1), "kernel_ms": ms, "Gpairs_per_s": N * (N - 1) / ms / 1e6,
       "algorithmic_bytes": N * (x_all.shape[1] * 8 + 16 + 64)}
if CPU and N <= 4096:
    cls = O.make_class(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max)
    ag = np.zeros(N, O.AGENT_DTYPE)
    ag["vmax"], ag["amax"] = 1.0, 2.0
    nn = min(N, 256)
    t0 = time.perf_counter()
    O.safety_metrics(cls, ag[:nn], x_all, sw.radius, sw.downwash, 1, 0.1)
    rec["cpu_oracle_ms_1core_scaled"] = (time.perf_counter() - t0) * 1e3 * N / nn
print(json.dumps(rec))

# the small per-agent kernels of a replan step: neighbour selection, shift of the previous plans, goal LP, validity / next state
d_nb2 = torch.zeros(N * n_obs, dtype=torch.int32, device=dev)
d_cnt = torch.zeros(N, dtype=torch.int32, device=dev)
d_pos = up(np.float32(sw.pos).astype(np.float64))
ms = timed(lambda: sol.select_neighbours_device(N, 0, N, n_obs, 3.0, d_pos, d_nb2, d_cnt), reps=100)
print(json.dumps({"kernel": "select_neighbours (broadcastMsgs range filter)", "agents": N, "kernel_ms": ms, "pairs": N * (N - 1)}))
ms = timed(lambda: sol.shift_traj_device(N, d_x, d_traj, z_2d=1.0, shift=1), reps=100)
print(json.dumps({"kernel": "shift_traj (initialTrajPlanningPrevSol)", "agents": N, "kernel_ms": ms}))
d_off = up((np.arange(N + 1) * n_obs * M * 6).astype(np.int64))
d_sfc0 = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
d_sfc0.view(N, M, 6)[:, :, :3] = -1e3
d_sfc0.view(N, M, 6)[:, :, 3:] = 1e3
d_gst = torch.zeros(N, dtype=torch.int32, device=dev)
ms = timed(lambda: sol.optimize_goal_device(N, d_hdr, d_rows, d_off, d_sfc0, d_gst), reps=100)
print(json.dumps({"kernel": "optimize_goal (GoalOptimizer LP, closed form)", "agents": N, "kernel_ms": ms}))
d_valid = torch.zeros(N, dtype=torch.int32, device=dev)
d_state = torch.zeros(N * 9, dtype=torch.float64, device=dev)
ms = timed(lambda: sol.validate_step_device(N, 0.2, d_x, d_hdr, d_sfc0, d_valid, d_state), reps=100)
print(json.dumps({"kernel": "validate_step (isSolValid + doStep)", "agents": N, "kernel_ms": ms}))

# voxel map + corridors: a random 3-D forest of pillars and blocks scaled to the swarm's world, 0.1 m cells
rng = np.random.default_rng(2)
wmin, wmax = np.array(sw.world_min, dtype=np.float64), np.array(sw.world_max, dtype=np.float64)
vol = float(np.prod(wmax - wmin))
nb = int(vol / 8.0)  # one obstacle per 8 m^3
boxes = np.concatenate([rng.uniform(wmin, wmax, (nb, 3)), rng.choice([0.3, 0.5, 0.8], (nb, 3))], axis=1)
t0 = time.perf_counter()
wm = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
torch.cuda.synchronize()
t_map = (time.perf_counter() - t0) * 1e3
nvox = int(np.prod(wm.dims))
rec = {"kernel": "map_create(rasterise + 3 nearest passes, incl. alloc/upload/sync)", "boxes": nb, "voxels": nvox, "wall_ms": t_map,
       "bytes_resident": nvox * 5}
if CPU:
    t0 = time.perf_counter()
    om = O.Map(boxes, wmin, wmax, 0.1, 1.0)
    rec["cpu_oracle_ms_openmp"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(rec))
starts = np.float32(sw.pos).astype(np.float64)
P = np.zeros((N, 3, 3))
P[:, 0] = P[:, 1] = P[:, 2] = starts
d_P = up(P.reshape(-1))
d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
d_st = torch.zeros(N, dtype=torch.int32, device=dev)
ms = timed(lambda: sol.construct_sfc_device(wm, api.SFC_INIT, N, d_P, d_r, d_sfc, d_st), reps=20)
ok = d_st.cpu().numpy() == 1
rec = {"kernel": "construct_sfc INIT (initializeSFC)", "agents": N, "kernel_ms": ms, "agents_per_s": N / ms * 1e3, "feasible_starts": int(ok.sum())}
if CPU:
    sfc = np.zeros((N, M), O.BOX_DTYPE)
    t0 = time.perf_counter()
    om.construct_sfc(O.SFC_INIT, P, sw.radius, sfc)
    rec["cpu_oracle_ms_openmp"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(rec))
# replan update: last point 0.3 m, goal / waypoint 0.5 m ahead
step = rng.normal(size=(N, 3))
step /= np.linalg.norm(step, axis=1, keepdims=True)
P2 = np.stack([starts + 0.3 * step, starts + 0.5 * step, starts + 0.5 * step], axis=1)
P2 = np.float32(P2).astype(np.float64)
d_P2 = up(P2.reshape(-1))
base = d_sfc.clone()


def upd():
    d_sfc.copy_(base)
    sol.construct_sfc_device(wm, api.SFC_FROM_HULL, N, d_P2, d_r, d_sfc, d_st)


ms_copy = timed(lambda: d_sfc.copy_(base), reps=20)
ms = timed(upd, reps=20) - ms_copy
rec = {"kernel": "construct_sfc FROM_HULL (constructSFCFromConvexHull)", "agents": N, "kernel_ms": ms, "agents_per_s": N / ms * 1e3,
       "new_boxes": int((d_st.cpu().numpy() == 1).sum())}
if CPU:
    sfc2 = sfc.copy()
    t0 = time.perf_counter()
    om.construct_sfc(O.SFC_FROM_HULL, P2, sw.radius, sfc2)
    rec["cpu_oracle_ms_openmp"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(rec))
# the same two launches with the map's free-space table in place (lscqp_map_prepare): identical boxes, tests in open space pass without sampling
wm.prepare(float(np.max(sw.radius)))
ref_init = None
d_sfc.zero_()
ms = timed(lambda: sol.construct_sfc_device(wm, api.SFC_INIT, N, d_P, d_r, d_sfc, d_st), reps=20)
print(json.dumps({"kernel": "construct_sfc INIT with the free-space table", "agents": N, "kernel_ms": ms, "agents_per_s": N / ms * 1e3,
                  "feasible_starts": int((d_st.cpu().numpy() == 1).sum())}))
base_t = d_sfc.clone()
assert torch.equal(base_t, base), "the free-space table changed a corridor"
ms = timed(upd, reps=20) - ms_copy
print(json.dumps({"kernel": "construct_sfc FROM_HULL with the free-space table", "agents": N, "kernel_ms": ms, "agents_per_s": N / ms * 1e3,
                  "new_boxes": int((d_st.cpu().numpy() == 1).sum())}))
# ... and in the work order a plan carries from replan to replan (round 4): the corridors that cost the most in the previous launch first
d_cost = torch.zeros(N, dtype=torch.int32, device=dev)
d_ord = torch.zeros(N, dtype=torch.int32, device=dev)
d_sfc.copy_(base)
sol.construct_sfc_device(wm, api.SFC_FROM_HULL, N, d_P2, d_r, d_sfc, d_st, d_cost=d_cost)
ref_boxes = d_sfc.clone()
sol.order_by_cost_device(N, d_cost, d_ord)


def upd_ordered():
    d_sfc.copy_(base)
    sol.construct_sfc_device(wm, api.SFC_FROM_HULL, N, d_P2, d_r, d_sfc, d_st, d_order=d_ord, d_cost=d_cost)


ms_o = timed(upd_ordered, reps=20) - ms_copy
assert torch.equal(d_sfc, ref_boxes), "the work order changed a corridor"
print(json.dumps({"kernel": "construct_sfc FROM_HULL with the free-space table, most expensive previous corridor first (lscqp_order_by_cost_device)",
                  "agents": N, "kernel_ms": ms_o, "agents_per_s": N / ms_o * 1e3, "as_given_ms": ms}))

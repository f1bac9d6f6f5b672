"""Closed-loop replanning with every row of the path on the device, in the reference's default configuration
(mode/planner = lsc, mode/goal = grid_based_planner: generateCLSC + constructSFCFromConvexHull + GoalOptimizer + TrajOptimizer):

    world boxes -> voxel map -> [ per replan: shift previous plans -> neighbours in range -> corridors -> CLSC rows -> goal LP -> trajectory QP
                                  -> isSolValid / next state -> safety metrics ]

    python tools/closed_loop.py [--steps 40] [--world tests/golden/forest10_world.json]

The host does only what the out-of-scope parts of the reference do: it picks each agent's next waypoint (the grid-based
planner / MAPF stand-in: 0.5 m towards the desired goal from the end of the current plan) and keeps the per-agent
headers.  Agents whose QP fails or whose solution is invalid keep their shifted previous plan, like the reference
(src/traj_planner.cpp:767-797).  Prints one JSON summary; tests/test_closed_loop.py runs it.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class GridRouter:
    """Stand-in for the reference's grid-based planner (out of scope: src/grid_based_planner.cpp, MAPF): shortest paths on
    the 0.5 m grid of launch/simulation.launch:88 around the inflated obstacles, one agent at a time (no conflict
    resolution between agents -- the LSCs keep them apart)."""

    def __init__(self, g, occ, key0, spacing=0.5):
        self.sp, self.wmin, self.wmax = spacing, np.array(g["world_min"][:2]), np.array(g["world_max"][:2])
        self.n = np.round((self.wmax - self.wmin) / spacing).astype(int) + 1
        res, reach = g["resolution"], g["radius"] + 0.1
        zk = int(np.floor(g["z_2d"] / res)) - key0[2]
        plane = occ[zk]
        self.free = np.zeros(self.n, bool)
        for i in range(self.n[0]):
            for j in range(self.n[1]):
                p = self.wmin + spacing * np.array([i, j])
                a = np.floor((p - reach) / res + 1e-6).astype(int) - key0[:2]
                b = np.ceil((p + reach) / res - 1e-6).astype(int) - key0[:2]
                a, b = np.maximum(a, 0), np.minimum(b, [plane.shape[1], plane.shape[0]])
                self.free[i, j] = not plane[a[1]:b[1], a[0]:b[0]].any()
        self.dist = {}

    def node(self, p):
        return tuple(np.clip(np.round((np.asarray(p[:2]) - self.wmin) / self.sp).astype(int), 0, self.n - 1))

    def field(self, goal):  # BFS distance-to-goal over free nodes (8-connected, no corner cutting)
        key = self.node(goal)
        if key in self.dist:
            return self.dist[key]
        D = np.full(self.n, np.inf)
        D[key] = 0
        todo = [key]
        while todo:
            nxt = []
            for (i, j) in todo:
                for di in (-1, 0, 1):
                    for dj in (-1, 0, 1):
                        a, b = i + di, j + dj
                        if (di or dj) and 0 <= a < self.n[0] and 0 <= b < self.n[1] and self.free[a, b]:
                            if di and dj and not (self.free[i + di, j] and self.free[i, j + dj]):
                                continue
                            w = D[i, j] + (1.4142 if di and dj else 1.0)
                            if w < D[a, b] - 1e-9:
                                D[a, b] = w
                                nxt.append((a, b))
            todo = nxt
        self.dist[key] = D
        return D

    def next_waypoint(self, pos, goal):
        D = self.field(goal)
        i, j = self.node(pos)
        best, arg = D[i, j], (i, j)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                a, b = i + di, j + dj
                if 0 <= a < self.n[0] and 0 <= b < self.n[1] and self.free[a, b] and D[a, b] < best - 1e-9:
                    if di and dj and not (self.free[i + di, j] and self.free[i, j + dj]):
                        continue
                    best, arg = D[a, b], (a, b)
        return self.wmin + self.sp * np.array(arg), self.wmin + self.sp * np.array([i, j])


def random_forest_world(n_agents=64, side=24.0, n_boxes=150, seed=0, clearance=0.8):
    """A synthetic 2-D forest in the reference's world format (pillars 0.5 x 0.5 x 2.5 m, like world/forest/*.csv) with
    n_agents start / goal pairs on opposite sides of a circle (the antipodal swap of the reference's missions), all on the
    0.5 m grid and clear of the pillars."""
    rng = np.random.default_rng(seed)
    half = side / 2
    ang = np.linspace(0, 2 * np.pi, n_agents, endpoint=False)
    starts = np.round(np.c_[np.cos(ang), np.sin(ang)] * (half - 2.0) / 0.5) * 0.5
    goals = -starts
    boxes = []
    while len(boxes) < n_boxes:
        c = rng.uniform(-half + 1, half - 1, 2)
        if np.abs(starts - c).max(axis=1).min() < clearance or np.abs(goals - c).max(axis=1).min() < clearance:
            continue
        boxes.append([c[0], c[1], 1.25, 0.5, 0.5, 2.5])
    z = 0.6
    return {"boxes": boxes, "world_min": [-half, -half, 0.0], "world_max": [half, half, 2.5], "resolution": 0.1, "max_dist": 1.0,
            "z_2d": z, "radius": 0.15, "starts": [[p[0], p[1], z] for p in starts], "goals": [[p[0], p[1], z] for p in goals]}


def run(world_json, steps=40, M=10, dt=0.2, verbose=False, dump=None, keep_step=None, n_obs=None, script=None):
    """script (optional): {"waypoint": (K, N, 3), "state": (K, N, 9)} -- replay of a recorded mission: replan k takes every agent's
    state and waypoint from the script instead of the loop's own step / router (the plans, goal points, corridors and neighbour sets are
    still the loop's own), and the result carries every replan's solution (`x`, (K, N, nv)) and goal point (`goal`, (K, N, 3))."""
    import torch

    from lsc_dr_planner_amd import api

    g = world_json if isinstance(world_json, dict) else json.load(open(world_json))
    dev = torch.device("cuda", 0)
    dim, z2d, radius = 2, float(g["z_2d"]), float(g["radius"])
    starts, desired = np.array(g["starts"], dtype=np.float64), np.array(g["goals"], dtype=np.float64)
    N = len(starts)
    # Row slots per agent.  The reference hands EVERY in-range agent to the planner (src/multi_sync_simulator.cpp:318-333), so the
    # slots follow the largest in-range count seen (lscqp_select_neighbours_device reports it) up to the largest compiled kernel
    # instance of the shape (lscqp_max_obstacles); only beyond that are the nearest kept, and that is counted in the result
    # (`truncated_agent_steps`) instead of happening silently.
    sol = api.Solver(api.make_desc(M=M, dim=dim, dt=dt, world_min=g["world_min"], world_max=g["world_max"]))
    cap = sol.max_obstacles()
    n_obs = min(N - 1, 8) if n_obs is None else n_obs
    n_obs = max(1, min(n_obs, cap))
    wmap = api.WorldMap(g["boxes"], g["world_min"], g["world_max"], g["resolution"], g["max_dist"])
    wmap.prepare(radius)  # the corridor kernel's free-space table (same boxes, tests in open space pass without sampling)
    nv = sol.nv
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    upb = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731

    # neighbours: chosen on the device each replan, the other agents within the communication range of the launch file (3 m)
    def row_buffers(k):
        return (torch.full((N * k,), -1, dtype=torch.int32, device=dev), up((np.arange(N + 1) * k * M * 6).astype(np.uint64).view(np.int64)),
                torch.zeros(N * k * M * 6 * 4, dtype=torch.float64, device=dev))

    d_nbr, d_off, d_rows = row_buffers(n_obs)
    d_ncount = torch.zeros(N, dtype=torch.int32, device=dev)
    comm_range = 3.0
    d_rad = torch.full((N,), radius, dtype=torch.float64, device=dev)
    d_dw = torch.full((N,), 2.0, dtype=torch.float64, device=dev)
    d_traj = torch.zeros(N * M * 6 * 3, dtype=torch.float64, device=dev)
    d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
    d_sst = torch.zeros(N, dtype=torch.int32, device=dev)
    d_gst = torch.zeros(N, dtype=torch.int32, device=dev)
    d_qst = torch.zeros(N, dtype=torch.int32, device=dev)
    d_x = torch.zeros(N * nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_valid = torch.zeros(N, dtype=torch.int32, device=dev)
    d_state = torch.zeros(N * 9, dtype=torch.float64, device=dev)
    d_saf = torch.zeros(N * api.SAFETY_DTYPE.itemsize, dtype=torch.uint8, device=dev)

    # t = 0: hover plans at the start points, corridors from initializeSFC
    state = np.zeros((N, 9))
    state[:, :3] = starts
    x_plan = np.zeros((N, dim, M, 6))
    for k in range(dim):
        x_plan[:, k] = starts[:, k, None, None]
    d_xprev = up(x_plan.reshape(N, -1))
    P0 = np.repeat(starts[:, None, :], 3, axis=1)
    sol.construct_sfc_device(wmap, api.SFC_INIT, N, up(P0.reshape(-1)), d_rad, d_sfc, d_sst)
    torch.cuda.synchronize()
    assert (d_sst.cpu().numpy() == 1).all(), "a start point lies inside an inflated obstacle"

    router = GridRouter(g, wmap.download()[0], wmap.key0) if script is None else None
    waypoint = starts.copy()  # first waypoint: the start node itself; it advances in the loop
    goal_pt = starts.copy()   # agent.current_goal_point
    log = {"steps": steps, "agents": N, "qp_failed": 0, "invalid": 0, "sfc_kept": 0, "goal_infeasible": 0, "min_safety_ratio": np.inf,
           "max_vel_excess": 0.0, "max_acc_excess": 0.0, "iters": []}
    d_info = torch.zeros(N * api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    dist0 = np.linalg.norm(desired[:, :2] - starts[:, :2], axis=1)
    for step in range(steps):
        # initial trajectories of this replan: the previous plans shifted by one segment (none before the first solve)
        sol.shift_traj_device(N, d_xprev, d_traj, z_2d=z2d, shift=0 if step == 0 else 1)
        torch.cuda.synchronize()
        traj = d_traj.cpu().numpy().reshape(N, M, 6, 3)
        last = traj[:, M - 1, 5]
        # the waypoint advances to the next node of the grid path once the agent is close to the current one (the reference's
        # planner hands out one grid step at a time, which keeps the QP's communication-range rows |x - waypoint| <= R/2
        # satisfiable); the current goal point is the outcome of the previous goal LP (src/traj_planner.cpp:545-549)
        if script is not None:
            state, waypoint = np.array(script["state"][step], dtype=np.float64), np.array(script["waypoint"][step], dtype=np.float64)
        else:
            for a in range(N):
                if np.abs(state[a, :2] - waypoint[a, :2]).max() < 0.3:
                    waypoint[a, :2] = router.next_waypoint(waypoint[a], desired[a])[0]
        waypoint = np.float32(waypoint).astype(np.float64)
        if step > 0:
            P = np.stack([last, goal_pt, waypoint], axis=1)
            sol.construct_sfc_device(wmap, api.SFC_FROM_HULL, N, up(P.reshape(-1)), d_rad, d_sfc, d_sst)
        d_goal_all = up(goal_pt)
        sol.select_neighbours_device(N, 0, N, n_obs, comm_range, up(state[:, :3]), d_nbr, d_ncount)  # broadcastMsgs' range filter
        need = int(d_ncount.max().item())
        if need > n_obs and n_obs < cap:  # more agents in range than slots: grow the row buffers and select again
            n_obs = min(need, cap)
            d_nbr, d_off, d_rows = row_buffers(n_obs)
            sol.select_neighbours_device(N, 0, N, n_obs, comm_range, up(state[:, :3]), d_nbr, d_ncount)
        log["row_slots"] = n_obs
        log["truncated_agent_steps"] = log.get("truncated_agent_steps", 0) + int((d_ncount > n_obs).sum().item())
        sol.generate_constraints_device(api.GEN_CLSC, N, n_obs, 0, d_traj, d_nbr, d_rad, d_dw, d_goal_all, d_rows)
        hdr = np.zeros(N, api.HEADER_DTYPE)
        hdr["p0"], hdr["v0"], hdr["a0"] = state[:, 0:3], state[:, 3:6], state[:, 6:9]
        hdr["goal"], hdr["next_waypoint"] = goal_pt, waypoint
        hdr["vmax"], hdr["amax"], hdr["radius"], hdr["nominal_velocity"], hdr["n_obs"] = 1.0, 2.0, radius, 1.0, n_obs
        d_hdr = upb(hdr)
        sol.optimize_goal_device(N, d_hdr, d_rows, d_off, d_sfc, d_gst)
        # x_init: the shifted plan in the solver's layout
        x_init = np.ascontiguousarray(traj.transpose(0, 3, 1, 2)[:, :dim].reshape(N, -1))
        d_xinit = up(x_init)
        sol.solve_device(N, n_obs, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_qst, d_info=d_info, d_x_init=d_xinit)
        sol.validate_step_device(N, dt, d_x, d_hdr, d_sfc, d_valid, d_state, z_2d=z2d)
        torch.cuda.synchronize()
        qst, valid = d_qst.cpu().numpy(), d_valid.cpu().numpy()
        if ((qst == 2) | (qst == 3)).any():
            # a warm-started iteration that ran out of iterations or broke down: one cold re-launch of the batch, results taken
            # for those agents only (what the host entry point lscqp_solve_batch does by itself)
            d_x2, d_obj2, d_st2 = torch.zeros_like(d_x), torch.zeros_like(d_obj), torch.zeros_like(d_qst)
            sol.solve_device(N, n_obs, d_hdr, d_rows, d_off, d_sfc, d_x2, d_obj2, d_st2)
            torch.cuda.synchronize()
            st2 = d_st2.cpu().numpy()
            fix = ((qst == 2) | (qst == 3)) & (st2 == 0)
            if fix.any():
                sel = torch.from_numpy(np.nonzero(fix)[0]).to(dev)
                d_x.view(N, nv)[sel] = d_x2.view(N, nv)[sel]
                d_obj[sel] = d_obj2[sel]
                d_qst[sel] = d_st2[sel]
                log["cold_retries"] = log.get("cold_retries", 0) + int(fix.sum())
                sol.validate_step_device(N, dt, d_x, d_hdr, d_sfc, d_valid, d_state, z_2d=z2d)
                torch.cuda.synchronize()
                qst, valid = d_qst.cpu().numpy(), d_valid.cpu().numpy()
        good = (qst == 0) & (valid == 1)
        x_new = d_x.cpu().numpy().reshape(N, nv)
        if keep_step is not None and step == keep_step:  # inputs and outputs of one replan, for the parity test
            log["kept"] = dict(hdr=d_hdr.cpu().numpy().view(api.HEADER_DTYPE).copy(), rows=d_rows.cpu().numpy().view(api.ROW_DTYPE).copy(),
                               sfc=d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(N, M).copy(), x=d_x.cpu().numpy().reshape(N, nv).copy(),
                               obj=d_obj.cpu().numpy().copy(), status=qst.copy(), n_obs=n_obs, world_min=g["world_min"], world_max=g["world_max"])
        if dump and (qst != 0).any() and not os.path.exists(dump):
            np.savez(dump, hdr=d_hdr.cpu().numpy(), rows=d_rows.cpu().numpy(), sfc=d_sfc.cpu().numpy(), qst=qst, x_init=x_init,
                     info=d_info.cpu().numpy(), step=step)
        if not good.all():  # the reference substitutes the initial trajectory (src/traj_planner.cpp:767-797)
            x_new[~good] = x_init[~good]
            d_x.copy_(up(x_new.reshape(-1)))
            sol.validate_step_device(N, dt, d_x, d_hdr, d_sfc, d_valid, d_state, z_2d=z2d)
        sol.safety_metrics_device(N, 0, N, 2, 0.1, d_x, d_rad, d_dw, d_hdr, d_saf, z_2d=z2d)
        torch.cuda.synchronize()
        saf = d_saf.cpu().numpy().view(api.SAFETY_DTYPE)
        state = d_state.cpu().numpy().reshape(N, 9)
        d_xprev.copy_(d_x.view(N, nv))
        goal_pt = d_hdr.cpu().numpy().view(api.HEADER_DTYPE)["goal"].copy()  # the goal LP's result is the next current goal point
        if script is not None:
            log.setdefault("x", []).append(x_new.copy())
            log.setdefault("goal", []).append(goal_pt.copy())
            log.setdefault("n_in_range", []).append(d_ncount.cpu().numpy().copy())
        log["max_in_range"] = int(max(log.get("max_in_range", 0), d_ncount.cpu().numpy().max()))
        log["qp_failed"] += int((qst != 0).sum())
        log["invalid"] += int(((qst == 0) & (valid != 1)).sum())
        log["sfc_kept"] += int((d_sst.cpu().numpy() == 0).sum()) if step > 0 else 0
        log["goal_infeasible"] += int((d_gst.cpu().numpy() != 0).sum())
        log["min_safety_ratio"] = float(min(log["min_safety_ratio"], saf["safety_ratio"].min()))
        log["max_vel_excess"] = float(max(log["max_vel_excess"], saf["vel_excess_ratio"].max()))
        log["max_acc_excess"] = float(max(log["max_acc_excess"], saf["acc_excess_ratio"].max()))
        log["iters"].append(int(d_info.cpu().numpy().view(api.INFO_DTYPE)["iterations"].max()))
        if verbose:
            print(step, "failed", int((qst != 0).sum()), "invalid", int(((qst == 0) & (valid != 1)).sum()), "safety",
                  float(saf["safety_ratio"].min()), "dist", np.linalg.norm(desired[:, :2] - state[:, :2], axis=1).round(2), file=sys.stderr)
    dist1 = np.linalg.norm(desired[:, :2] - state[:, :2], axis=1)
    log["mean_progress_m"] = float((dist0 - dist1).mean())
    log["min_progress_m"] = float((dist0 - dist1).min())
    log["sim_time_s"] = steps * dt
    log["max_iters"] = int(max(log["iters"]))
    del log["iters"]
    # occupancy check of the flown positions: the agents (L-infinity radius) never touch an occupied cell
    wmap.close()
    return log


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--world", default=os.path.join(ROOT, "tests", "golden", "forest10_world.json"))
    ap.add_argument("-v", action="store_true")
    ap.add_argument("--forest", type=int, default=0, help="N > 0: a synthetic forest with N agents instead of --world")
    ap.add_argument("--obs", type=int, default=None, help="neighbour capacity per agent")
    ap.add_argument("--dump", default=None, help="npz path: inputs of the first replan with a failed QP")
    a = ap.parse_args()
    world = random_forest_world(a.forest) if a.forest > 0 else a.world
    print(json.dumps(run(world, steps=a.steps, verbose=a.v, dump=a.dump, n_obs=a.obs)))

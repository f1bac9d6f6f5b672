"""Whole-replan timing of lscqp_plan (include/lscqp.h): eager chain vs captured hipGraph, closed loop on the device.

    python tools/plan_timing.py [--agents 64] [--steps 100]

--agents 0 flies the reference's forest10 world (10 agents); N > 0 a synthetic forest with N agents (tools/closed_loop.py).  The
waypoints come from the same host-side grid router as in tools/closed_loop.py, re-evaluated every replan (that is the host work the
chain leaves over).  Prints one JSON line; run it under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=0)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--obs", type=int, default=0)
    a = ap.parse_args()
    import torch

    import closed_loop
    from lsc_dr_planner_amd import api

    W = closed_loop.random_forest_world(a.agents) if a.agents > 0 else json.load(open(os.path.join(ROOT, "tests", "golden", "forest10_world.json")))
    N = len(W["starts"])
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    n_obs = a.obs or min(N - 1, sol.max_obstacles())
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, n_obs, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=W["z_2d"])
    router = closed_loop.GridRouter(W, wmap.download()[0], wmap.key0)
    starts, desired = np.array(W["starts"], dtype=np.float64), np.array(W["goals"], dtype=np.float64)
    out = dict(agents=N, n_obs=n_obs, steps=a.steps)
    for mode in ("eager", "graph"):
        plan.reset(starts)
        way = starts.copy()
        t_dev = t_host = t_sub = 0.0
        failed = cut = 0
        for k in range(a.steps):
            t0 = time.perf_counter()
            state = plan.get(api.PLAN_STATE).reshape(N, 9)
            for i in range(N):
                if np.abs(state[i, :2] - way[i, :2]).max() < 0.3:
                    way[i, :2] = router.next_waypoint(way[i], desired[i])[0]
            plan.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
            t1 = time.perf_counter()
            plan.step(graph=(mode == "graph"))
            t1s = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            t_host += t1 - t0
            if k >= 2:
                t_dev += t2 - t1
                t_sub += t1s - t1
            failed += int((plan.get(api.PLAN_STATUS) != 0).sum())
            cut += int((plan.get(api.PLAN_IN_RANGE) > n_obs).sum())
        out[mode] = dict(replan_us=t_dev / (a.steps - 2) * 1e6, host_submit_us=t_sub / (a.steps - 2) * 1e6, host_waypoints_us=t_host / a.steps * 1e6, failed_qps=failed, cut_neighbour_lists=cut,
                         progress_m=float((np.linalg.norm(desired[:, :2] - starts[:, :2], axis=1) - np.linalg.norm(desired[:, :2] - plan.get(api.PLAN_STATE).reshape(N, 9)[:, :2], axis=1)).mean()))
    out["graph_nodes"] = plan.graph_nodes()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Development aid: per-phase cycle breakdown of the corridor kernel inside the replan chain.  Needs a library built with
-DLSCSFC_DEBUG (LSCQP_AB=dbg LSCQP_EXTRA_FLAGS=-DLSCSFC_DEBUG python -m lsc_dr_planner_amd.build, then
LSCQP_LIB=lsc_dr_planner_amd/liblscqp_dbg.so python tools/sfc_phase_timing.py [n_agents]).  Per agent and replan: batches of
look-ahead tests, boxes tested, 64-column chunks, shader cycles in the batch tests / the whole kernel / table fill / column rounds /
look-ahead generation / state update, and the number of column rounds."""
import sys, os, json, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from lsc_dr_planner_amd import api
import closed_loop
L = api.lib()
W = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'forest10_world.json'))) if len(sys.argv) < 2 else closed_loop.random_forest_world(int(sys.argv[1]))
N = len(W["starts"])
sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
plan = api.Plan(sol, wmap, N, min(N - 1, sol.max_obstacles()), ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=W["z_2d"])
router = closed_loop.GridRouter(W, wmap.download()[0], wmap.key0)
starts, desired = np.array(W["starts"], dtype=np.float64), np.array(W["goals"], dtype=np.float64)
plan.reset(starts); way = starts.copy()
buf = (C.c_ulonglong * 32)()
for k in range(30):
    state = plan.get(api.PLAN_STATE).reshape(N, 9)
    for i in range(N):
        if np.abs(state[i, :2] - way[i, :2]).max() < 0.3:
            way[i, :2] = router.next_waypoint(way[i], desired[i])[0]
    plan.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
    L.lscsfc_dbg_read(buf, 1)
    plan.step(); torch.cuda.synchronize()
    L.lscsfc_dbg_read(buf, 1)
    b = list(buf)
    if k % 5 == 0 or k < 3:
        print("replan", k, "per agent: batches %.1f boxes %.1f col-chunks %.1f alone %.1f test-cycles %.0f kernel-cycles %.0f (exp calls %.1f)" % tuple(v / N for v in b[:7]), "| fill %.0f rounds-cyc %.0f gen %.0f replay %.0f nrounds %.1f" % tuple(b[i] / N for i in (7, 8, 9, 10, 11)),
              "| filter: chunks %.0f todo %.0f n %.1f nofilter-chunks %.0f | batch ends: boundary %.2f obstacle %.2f limit %.2f passes %.1f end %.2f | gen cumulative: tables %.0f segs %.0f box %.0f free %.0f A-done %.0f | filter-cyc %.0f / both %.0f" % tuple(b[i] / N for i in (12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 31)))

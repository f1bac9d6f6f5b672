"""Development aid: the corridor kernel's phase counters (-DLSCSFC_DEBUG library) on bench.py's 3-D chain workload."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsc_dr_planner_amd import api  # noqa: E402

L = api.lib()
N, radii = 64, (10.0, 10.0, 4.0)
rng = np.random.default_rng(7)
i = np.arange(N) + 0.5
phi, th = np.arccos(1 - 2 * i / N), np.pi * (1 + 5 ** 0.5) * i
starts = np.round((np.c_[np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)] * list(radii) + [0, 0, radii[2] + 1.0]) * 4) / 4
goals = np.c_[-starts[:, 0], -starts[:, 1], 2 * (radii[2] + 1.0) - starts[:, 2]]
boxes = []
while len(boxes) < 24:
    c = np.r_[rng.uniform(-0.7 * radii[0], 0.7 * radii[0], 2), rng.uniform(1.5, 2 * radii[2] + 0.5)]
    if np.abs(starts - c).max(axis=1).min() > 1.2:
        boxes.append([c[0], c[1], c[2], 0.8, 0.8, 0.8])
wmin, wmax = [-radii[0] - 2.0, -radii[1] - 2.0, 0.0], [radii[0] + 2.0, radii[1] + 2.0, 2 * radii[2] + 2.0]
sol = api.Solver(api.make_desc(M=5, dim=3, dt=0.2, world_min=wmin, world_max=wmax))
wmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
plan = api.Plan(sol, wmap, N, 20, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True, closed_loop=True)
plan.reset(starts)
buf = (C.c_ulonglong * 32)()
for k in range(30):
    pos = plan.get(api.PLAN_STATE).reshape(N, 9)[:, :3]
    d = goals - pos
    dist = np.linalg.norm(d, axis=1, keepdims=True)
    way = pos + d / np.maximum(dist, 1e-9) * np.minimum(dist, 0.75)
    plan.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
    L.lscsfc_dbg_read(buf, 1)
    plan.step(graph=False)
    torch.cuda.synchronize()
    L.lscsfc_dbg_read(buf, 1)
    b = list(buf)
    if k % 5 == 0 or k < 3:
        print("replan", k, "per agent: batches %.1f boxes %.1f col-chunks %.1f alone %.1f test-cycles %.0f kernel-cycles %.0f (exp calls %.1f)" % tuple(v / N for v in b[:7]),
              "| fill %.0f rounds-cyc %.0f gen %.0f replay %.0f nrounds %.1f | filter: chunks %.0f todo %.0f n %.1f nofilter-chunks %.0f" % tuple(b[i] / N for i in (7, 8, 9, 10, 11, 12, 13, 14, 15)),
              "| batch ends: boundary %.2f obstacle %.2f limit %.2f passes %.1f end %.2f" % tuple(b[i] / N for i in (16, 17, 18, 19, 20)),
              "| gen cumulative: F %.0f segs %.0f box %.0f free %.0f A-done %.0f" % tuple(b[i] / N for i in (21, 22, 23, 24, 25)),
              "| filter: thread 0 before the table %.0f, in it %.0f (x %.1f), setup %.0f, pass 1 done %.0f, both %.0f" % tuple(b[i] / N for i in (27, 28, 29, 30, 26, 31)))

"""Golden vectors for the goal LP: the rows GoalOptimizer::populatebyrow builds (reference
src/goal_optimizer.cpp:72-147), assembled by the oracle restatement from reference-style LSC / Box records, solved by an
independent LP solver (scipy.optimize.linprog, HiGHS).  The reference solves the same LP with CPLEX, which is absent here.

    python tools/make_golden_goal.py  ->  tests/golden/goal_lp.json   (inputs, rows, HiGHS optimum / infeasibility)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import linprog

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(20240929)
cases = []
for t in range(160):
    dim = 3 if t % 3 else 2
    M = [5, 10, 6][t % 3]
    cls = O.make_class(M=M, dim=dim, use_sfc=bool(t % 4))
    w = np.float32(rng.uniform(-3, 3, 3)).astype(float)
    g = np.float32(w + rng.normal(size=3) * rng.uniform(0.1, 1.0)).astype(float)
    if dim == 2:
        g[2] = w[2] = 1.0
    n_obs = int(rng.integers(0, 12))
    t0 = rng.uniform(0, 1)
    lsc = np.zeros((n_obs, M, 6), O.LSC_DTYPE)
    for oi in range(n_obs):
        nrm = rng.normal(size=3)
        if dim == 2:
            nrm[2] = 0
        nrm /= np.linalg.norm(nrm)
        if rng.random() < 0.1:
            nrm *= 1e-7  # dropped row
        p = w + rng.normal(size=3) * 1.5
        # rows hold at x0 = (g - w) t0 + w with a random slack (zero for some: active at t0), so the LP is feasible with
        # t* <= t0; every seventh case gets one row that cuts the whole segment off (infeasible)
        x0 = (g - w) * t0 + w
        d = (x0 - p) @ np.float32(nrm).astype(float) - (0.0 if rng.random() < 0.3 else rng.uniform(0, 0.5))
        if t % 7 == 0 and oi == 0:
            d = max((w - p) @ nrm, (g - p) @ nrm) + 0.3
        lsc["p"][oi, M - 1, 5] = np.float32(p)
        lsc["nrm"][oi, M - 1, 5] = np.float32(nrm)
        lsc["d"][oi, M - 1, 5] = d
    box = np.zeros(1, O.BOX_DTYPE)
    x0 = (g - w) * t0 + w
    lo = np.minimum(w, x0) - rng.uniform(0.0, 1.0, 3) + (rng.random(3) < 0.3) * np.abs(x0 - w) * 0.5
    hi = np.maximum(w, x0) + rng.uniform(0.0, 1.0, 3) - (rng.random(3) < 0.3) * np.abs(x0 - w) * 0.5
    box["bmin"], box["bmax"] = np.minimum(lo, x0), np.maximum(hi, x0)
    a, c = O.goal_rows(cls, g, w, lsc if n_obs else None, box[0] if cls.use_sfc else None)
    if np.linalg.norm(g - w) < 1e-5:
        continue
    res = linprog([1.0], A_ub=-a.reshape(-1, 1) if len(a) else None, b_ub=c if len(a) else None, bounds=[(0, 1 + 1e-5)],
                  method="highs", options={"primal_feasibility_tolerance": 1e-10, "dual_feasibility_tolerance": 1e-10})
    cases.append({"M": M, "dim": dim, "use_sfc": int(cls.use_sfc), "goal": g.tolist(), "next_waypoint": w.tolist(),
                  "lsc_p": lsc["p"][:, M - 1, 5].tolist(), "lsc_nrm": lsc["nrm"][:, M - 1, 5].tolist(),
                  "lsc_d": lsc["d"][:, M - 1, 5].tolist(), "box_min": box["bmin"][0].tolist(), "box_max": box["bmax"][0].tolist(),
                  "rows_a": a.tolist(), "rows_c": c.tolist(),
                  "status": int(res.status), "t": float(res.x[0]) if res.status == 0 else None})
print(len(cases), "cases;", sum(c["status"] != 0 for c in cases), "infeasible;", sum(c["status"] == 0 and c["t"] > 1e-9 for c in cases),
      "with t > 0")
json.dump({"source": "rows: oracle restatement of reference src/goal_optimizer.cpp:118-155; optimum: scipy linprog (HiGHS)",
           "cases": cases}, open(os.path.join(ROOT, "tests", "golden", "goal_lp.json"), "w"))

"""Generates tests/golden/kat_log_active.json in the BUILD container (reads /root/reference/log): known answers of the reference's
own run in which an LSC ROW IS ACTIVE.

tools/make_golden_log_replay.py certifies, for 337 replans, the QP inputs (initial state from the log, inferred waypoint) with which
the row-for-row restatement reproduces the logged motion WITHOUT any LSC / SFC row.  Those replans give back the plans the
reference's agents held at those times (the restatement's optimum, truncated to float32 like TrajOptResult::desired_traj).  For a
replan k of agent a that was NOT matched, if the previous replan of a and of every agent within communication range WAS matched, all
inputs of generateCLSC are known: the shifted previous plans (initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol,
src/traj_planner.cpp:273-310, 399-411) and the previous goal points.  The oracle's generateCLSC restatement makes the rows, its
GoalOptimizer restatement the goal for every candidate waypoint, the QP is solved WITH the rows, and a candidate is accepted on the same
criterion as before (all twelve logged numbers of t + 0.1 s and t + 0.2 s).  Accepted cases whose optimum carries a non-zero LSC
multiplier are the known answers wanted: reference-logged motion that only the LSC rows explain.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "tests", "golden", "kat_log_active.json")


def ulp6(v):
    return 10.0 ** (np.floor(np.log10(abs(v))) - 5) if v != 0 else 1e-6


def main():
    from oracle import oracle as O

    O.build()
    rows = list(csv.reader(open(os.path.join(REF, "log", "simulation_1663743693.650981_LSC_10agents.csv"))))
    ncol = 12
    nag = len(rows[0]) // ncol
    T = {}
    for r in rows[1:]:
        for a in range(nag):
            f = [float(v) for v in r[a * ncol:(a + 1) * ncol]]
            T.setdefault(a, []).append(dict(t=f[1], p=f[2:5], v=f[5:8], a=f[8:11]))
    g = json.load(open(os.path.join(HERE, "tests", "golden", "kat_log_replay.json")))
    p = g["params"]
    M, dim, R = p["M"], 2, p["comm_range"]
    cls = O.make_class(M=M, dim=dim, dt=p["dt"], w_c=p["w_c"], w_t=p["w_t"], comm_range=R, planner_lsc=True, use_sfc=False,
                       world_min=p["world_min"], world_max=p["world_max"])
    matched = {(c["agent"], c["replan"]): c for c in g["cases"]}
    mission = json.load(open(os.path.join(REF, "missions", "forest10", "forest10_10.json")))
    starts = [a["start"] for a in mission["agents"]]
    # the plans of the matched replans (float32 like desired_traj), (M, 6, 3)
    plan = {}
    for key, c in matched.items():
        ag = O.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"], vmax=p["vmax"], amax=p["amax"],
                          radius=p["radius"], nominal_velocity=p["nominal_velocity"])
        x = O.solve(cls, ag, None, None)["x"].reshape(dim, M, 6)
        tr = np.zeros((M, 6, 3))
        tr[..., 0], tr[..., 1], tr[..., 2] = x[0], x[1], 0.6
        plan[key] = np.float32(tr).astype(np.float64)
    n_replans = (len(T[0]) - 2) // 2
    cases, tried = [], 0
    # Every accepted replan (with or without an active row) extends the known history, so the sweep over replans is in time order
    # and a replan accepted at k is available as the previous plan at k + 1.
    for k in range(2, n_replans):
        pos = np.float32([T[a][2 * k]["p"] for a in range(nag)])
        for a in range(nag):
            if (a, k) in matched or (a, k - 1) not in matched:
                continue
            st, s1, s2 = T[a][2 * k], T[a][2 * k + 1], T[a][2 * k + 2]
            if max(abs(v) for v in st["v"][:2]) < 1e-4:
                continue
            d = np.abs(pos - pos[a]).astype(np.float64).max(axis=1)
            nbr = [j for j in range(nag) if j != a and d[j] <= R]
            if not nbr or any((j, k - 1) not in matched for j in nbr):
                continue
            tried += 1
            traj = np.zeros((nag, M, 6, 3))
            goal_all = np.zeros((nag, 3))
            for j in [a] + nbr:
                prev = plan[(j, k - 1)]
                traj[j, :-1] = prev[1:]
                traj[j, -1] = prev[-1, 5]
                goal_all[j] = matched[(j, k - 1)]["goal"]
            nb = np.array([nbr], dtype=np.int32)
            L = O.generate_constraints(O.MODE_CLSC, traj, nb, p["radius"], 2.0, goal_all, dim=dim, first_agent=a)[0]
            p0 = np.array(st["p"])
            s = np.array(starts[a][:2])
            base = np.round((p0[:2] - s) / 0.5)
            best = None
            for dx in range(-3, 4):
                for dy in range(-3, 4):
                    w = np.array([*(s + 0.5 * (base + np.array([dx, dy]))), 0.6])
                    gst, goal, tpar = O.goal_opt(cls, goal_all[a], w, lsc=L)
                    if gst != 0:
                        continue
                    goal = np.float32(goal).astype(np.float64)
                    ag = O.make_agent(p0=[p0[0], p0[1], 0.6], v0=st["v"], a0=st["a"], goal=goal, next_waypoint=w, vmax=p["vmax"], amax=p["amax"],
                                      radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(nbr))
                    Rs = O.solve(cls, ag, L, None)
                    if Rs["status"] != 0:
                        continue
                    err = 0.0
                    for sl in (s1, s2):
                        ps, vl, ac = O.state_at(cls, Rs["x"], sl["t"] - st["t"])
                        for got, logged in ((ps, sl["p"]), (vl, sl["v"]), (ac, sl["a"])):
                            for gk, lk in zip(got[:2], logged[:2]):
                                err = max(err, abs(gk - lk) / max(ulp6(lk), 1e-6))
                    if best is None or err < best[0]:
                        sz = O.count(cls, ag, L)
                        lam = Rs["lam"][sz.n_sfc:sz.n_sfc + sz.n_lsc]
                        best = (err, w, goal, tpar, float(lam.max()) if len(lam) else 0.0, Rs["obj"])
            if best is None:
                continue
            err, w, goal, tpar, lam_max, obj = best
            print("agent %d replan %d: %d neighbours, best match %.1f units, goal-LP t %.4f, max LSC multiplier %.3e" % (a, k, len(nbr), err, tpar, lam_max), flush=True)
            if err <= 150.0:  # accepted: its plan and goal become history for the next replans
                ag = O.make_agent(p0=[p0[0], p0[1], 0.6], v0=st["v"], a0=st["a"], goal=goal, next_waypoint=w, vmax=p["vmax"], amax=p["amax"],
                                  radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(nbr))
                xx = O.solve(cls, ag, L, None)["x"].reshape(dim, M, 6)
                tr = np.zeros((M, 6, 3))
                tr[..., 0], tr[..., 1], tr[..., 2] = xx[0], xx[1], 0.6
                plan[(a, k)] = np.float32(tr).astype(np.float64)
                matched[(a, k)] = dict(goal=goal.tolist())
            if err <= 150.0 and lam_max > 1e-6:
                cases.append(dict(agent=a, replan=k, t=st["t"], p0=[p0[0], p0[1], 0.6], v0=st["v"], a0=st["a"], goal=goal.tolist(),
                                  goal_before_lp=goal_all[a].tolist(), goal_lp_t=tpar,
                                  next_waypoint=w.tolist(), neighbours=nbr, lsc_p=L["p"].tolist(), lsc_nrm=L["nrm"].tolist(), lsc_d=L["d"].tolist(),
                                  states=[s1, s2], match_units_of_6th_digit=round(err, 2), max_lsc_multiplier=lam_max, oracle_obj=obj))
    print("tried %d replans with a fully known neighbourhood; %d known answers with an ACTIVE LSC row" % (tried, len(cases)))
    if cases:
        json.dump(dict(source="reference log/simulation_1663743693.650981_LSC_10agents.csv; see tools/make_golden_log_active.py", params=p, cases=cases),
                  open(OUT, "w"))


if __name__ == "__main__":
    main()

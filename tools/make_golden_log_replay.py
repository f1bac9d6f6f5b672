"""Generates tests/golden/kat_log_replay.json in the BUILD container (reads /root/reference/log): known answers of LATER replans of
the reference's own run, with non-zero initial velocity and acceleration.

The reference's result log (log/simulation_1663743693.650981_LSC_10agents.csv: forest10_10, dim 2, M 10, launch/simulation.launch
parameters) records every agent's state every 0.1 s; replans happen every 0.2 s.  For replan k of agent a the QP's initial state
(p0, v0, a0) is the logged state at t = 0.2 k (6 printed digits), and the states at t + 0.1 and t + 0.2 are what the solved
trajectory evaluates to.  What the log does not record is the waypoint the grid planner handed the agent; it is INFERRED: every
lattice point of the 0.5 m waypoint grid (launch/simulation.launch:88) near the agent is tried as goal = next_waypoint
(GoalOptimizer returns the waypoint itself when nothing blocks it, src/goal_optimizer.cpp:109-165), the row-for-row restatement
of the QP is solved without LSC / SFC rows, and a candidate is accepted only if it reproduces all twelve logged numbers
(p, v, a in x and y at both times).  Matches are unmistakable: accepted candidates deviate by a few units of the log's sixth
digit (the inputs themselves are only known to six digits), rejected ones by 1e3 .. 1e5 units -- so a match also certifies
that no LSC / SFC row was active in that replan.  Replans where an inequality of the other agents or the corridor was active,
or where GoalOptimizer moved the goal off the lattice, find no candidate and are left out.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "tests", "golden", "kat_log_replay.json")


def ulp6(v):
    return 10.0 ** (np.floor(np.log10(abs(v))) - 5) if v != 0 else 1e-6


def main():
    from oracle import oracle as O

    O.build()
    rows = list(csv.reader(open(os.path.join(REF, "log", "simulation_1663743693.650981_LSC_10agents.csv"))))
    ncol = 12
    nag = len(rows[0]) // ncol
    mission = json.load(open(os.path.join(REF, "missions", "forest10", "forest10_10.json")))
    starts = [a["start"] for a in mission["agents"]]
    T = {}
    for r in rows[1:]:
        for a in range(nag):
            f = [float(v) for v in r[a * ncol:(a + 1) * ncol]]
            T.setdefault(a, []).append(dict(t=f[1], p=f[2:5], v=f[5:8], a=f[8:11]))
    wmin = mission["world"][0]["dimension"][:3]
    wmax = mission["world"][0]["dimension"][3:]
    params = dict(M=10, n=5, phi=3, dim=2, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0, world_z_2d=0.6, world_min=wmin, world_max=wmax,
                  vmax=[1.0, 1.0, 1.0], amax=[2.0, 2.0, 2.0], radius=0.15, nominal_velocity=1.0, planner_mode="LSC", use_sfc=False)
    cls = O.make_class(M=10, dim=2, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0, planner_lsc=True, use_sfc=False, world_min=wmin, world_max=wmax)
    cases, rejected = [], 0
    n_replans = (len(T[0]) - 2) // 2
    for a in range(nag):
        s = np.array(starts[a][:2])
        for k in range(1, n_replans):  # k = 0 is tests/golden/kat_log.json
            st, s1, s2 = T[a][2 * k], T[a][2 * k + 1], T[a][2 * k + 2]
            if max(abs(v) for v in st["v"][:2]) < 1e-4:
                continue  # hovering at its goal: nothing to pin
            p0 = np.array(st["p"])
            base = np.round((p0[:2] - s) / 0.5)
            best = None
            for dx in range(-3, 4):
                for dy in range(-3, 4):
                    w = s + 0.5 * (base + np.array([dx, dy]))
                    ag = O.make_agent(p0=[p0[0], p0[1], 0.6], v0=st["v"], a0=st["a"], goal=[w[0], w[1], 0.6], next_waypoint=[w[0], w[1], 0.6],
                                      vmax=[1, 1, 1], amax=[2, 2, 2], radius=0.15, nominal_velocity=1.0)
                    R = O.solve(cls, ag, None, None)
                    if R["status"] != 0:
                        continue
                    err, abs_err = 0.0, dict(p=0.0, v=0.0, a=0.0)
                    for sl in (s1, s2):
                        pos, vel, acc = O.state_at(cls, R["x"], sl["t"] - st["t"])
                        for key, got, logged in (("p", pos, sl["p"]), ("v", vel, sl["v"]), ("a", acc, sl["a"])):
                            for gk, lk in zip(got[:2], logged[:2]):
                                err = max(err, abs(gk - lk) / max(ulp6(lk), 1e-6))
                                abs_err[key] = max(abs_err[key], abs(gk - lk))
                    if best is None or err < best[0]:
                        best = (err, w, R["obj"], abs_err)
            if best is not None and best[0] <= 150.0:
                err, w, obj, abs_err = best
                cases.append(dict(agent=a, replan=k, t=st["t"], p0=[p0[0], p0[1], 0.6], v0=st["v"], a0=st["a"],
                                  goal=[float(w[0]), float(w[1]), 0.6], next_waypoint=[float(w[0]), float(w[1]), 0.6],
                                  states=[s1, s2], match_units_of_6th_digit=round(err, 2), match_abs=abs_err, oracle_obj=obj))
            else:
                rejected += 1
    json.dump(dict(source="reference log/simulation_1663743693.650981_LSC_10agents.csv, replans 1.. of every agent; waypoints inferred "
                          "(see tools/make_golden_log_replay.py); launch/simulation.launch:44-100; missions/forest10/forest10_10.json",
                   params=params, cases=cases, rejected=rejected), open(OUT, "w"), indent=0)
    e = np.array([c["match_units_of_6th_digit"] for c in cases])
    print("kat_log_replay.json: %d cases (%d rejected), match error units: median %.1f max %.1f; abs p %.1e v %.1e a %.1e" % (
        len(cases), rejected, np.median(e), e.max(), max(c["match_abs"]["p"] for c in cases), max(c["match_abs"]["v"] for c in cases),
        max(c["match_abs"]["a"] for c in cases)))


if __name__ == "__main__":
    main()

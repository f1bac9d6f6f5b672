"""Generates tests/golden/*.json in the BUILD container (needs /root/reference for the log-derived vectors and
scipy for the independent solutions).  The fixtures are data: inputs and expected outputs.

  kat_log.json     reference-authored known answers: the first replan of forest10_10 as recorded in the reference's
                   result log (log/simulation_1663743693.650981_LSC_10agents.csv rows t=0.1 and t=0.2), with the
                   launch parameters of launch/simulation.launch:44-100 and the start positions of
                   missions/forest10/forest10_10.json.  Agents 0,2..8: no active inequality; agents 1,9: the SFC face
                   at 2.55 m (world/forest/forest10.csv box rasterised on the 0.1 m grid) is active.
  scipy_*.json     synthetic instances (lsc_dr_planner_amd/synth.py) with the solution scipy's trust-constr finds on
                   the dense model assembled row-for-row by the oracle; polished by an active-set solve in 80-bit.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "tests", "golden")


def kat_log():
    path = os.path.join(REF, "log", "simulation_1663743693.650981_LSC_10agents.csv")
    rows = list(csv.reader(open(path)))
    hdr = rows[0]
    ncol = 12
    nag = len(hdr) // ncol
    mission = json.load(open(os.path.join(REF, "missions", "forest10", "forest10_10.json")))
    starts = [a["start"] for a in mission["agents"]]
    rec = {}
    for r in rows[1:4]:
        for a in range(nag):
            f = [float(v) for v in r[a * ncol:(a + 1) * ncol]]
            rec.setdefault(a, []).append(dict(t=f[1], p=f[2:5], v=f[5:8], a=f[8:11]))
    out = dict(
        source="reference log/simulation_1663743693.650981_LSC_10agents.csv rows t=0,0.1,0.2; "
               "launch/simulation.launch:44-100; missions/forest10/forest10_10.json",
        params=dict(M=10, n=5, phi=3, dim=2, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0, world_z_2d=0.6,
                    world_min=mission["world"][0]["dimension"][:3], world_max=mission["world"][0]["dimension"][3:],
                    vmax=[1.0, 1.0, 1.0], amax=[2.0, 2.0, 2.0], radius=0.15, nominal_velocity=1.0, grid_step=0.5,
                    planner_mode="LSC"),
        agents=[],
    )
    for a in range(nag):
        s = starts[a]
        out["agents"].append(dict(id=a, start=[s[0], s[1], 0.6], states=rec[a]))
    # Per-agent QP inputs of the first replan, as derived in SURVEY.md §8c: goal = next_waypoint = one 0.5 m grid
    # step along the dominant axis towards the goal; agents 1 and 9 have the SFC face |x| >= ... at 2.55 m and the
    # GoalOptimizer clips their goal to it.
    out["cases"] = [
        dict(name="kat1_agent0", agent=0, p0=[4.0, 0.0, 0.6], goal=[3.5, 0.0, 0.6], next_waypoint=[3.5, 0.0, 0.6], sfc=None),
        dict(name="kat2_agent1", agent=1, p0=[3.0, 2.5, 0.6], goal=[2.55, 2.5, 0.6], next_waypoint=[2.5, 2.5, 0.6],
             sfc=dict(bmin=[2.55, -5.0, 0.0], bmax=[5.0, 5.0, 2.5])),
    ]
    json.dump(out, open(os.path.join(OUT, "kat_log.json"), "w"), indent=1)
    print("kat_log.json:", nag, "agents")


def polish(A, x0):
    """Active-set polish in 80-bit long double (x86): solve the KKT system of the equality-constrained QP on the
    rows active at x0."""
    import scipy.linalg as sl

    LD = np.longdouble
    P, q, r, Aeq, beq, G, h, lb, ub = [A[k] for k in ("P", "q", "r", "Aeq", "beq", "G", "h", "lb", "ub")]
    nv = len(q)
    rows, rhs = [Aeq], [beq]
    act = np.where(G @ x0 - h > -1e-7)[0]
    rows.append(G[act]); rhs.append(h[act])
    for l in range(nv):
        if np.isfinite(lb[l]) and x0[l] - lb[l] < 1e-7:
            e = np.zeros(nv); e[l] = 1; rows.append(e[None]); rhs.append([lb[l]])
        if np.isfinite(ub[l]) and ub[l] - x0[l] < 1e-7:
            e = np.zeros(nv); e[l] = 1; rows.append(e[None]); rhs.append([ub[l]])
    Aa = np.concatenate(rows); ba = np.concatenate(rhs)
    _, Rm, piv = sl.qr(Aa.T, pivoting=True, mode="economic")
    rank = int((np.abs(np.diag(Rm)) > 1e-9 * abs(Rm[0, 0])).sum())
    keep = np.sort(piv[:rank])
    Aa, ba = Aa[keep].astype(LD), ba[keep].astype(LD)
    na = len(ba)
    K = np.zeros((nv + na, nv + na), LD)
    K[:nv, :nv] = 2 * P.astype(LD); K[:nv, nv:] = Aa.T; K[nv:, :nv] = Aa
    Mx = np.concatenate([K, np.concatenate([-q.astype(LD), ba])[:, None]], axis=1)
    n = nv + na
    for c in range(n):
        p = c + int(np.argmax(np.abs(Mx[c:, c])))
        Mx[[c, p]] = Mx[[p, c]]
        Mx[c] /= Mx[c, c]
        for rr in range(n):
            if rr != c and Mx[rr, c] != 0:
                Mx[rr] -= Mx[rr, c] * Mx[c]
    xs = Mx[:nv, -1]
    obj = xs @ (P.astype(LD) @ xs) + q.astype(LD) @ xs + LD(r)
    mult = np.asarray(Mx[nv:, -1], float)
    return np.asarray(xs, float), float(obj), int(len(act)), mult


def scipy_cases():
    from scipy.optimize import Bounds, LinearConstraint, minimize

    from lsc_dr_planner_amd import synth
    from oracle import oracle as O

    specs = [("scipy_m5d3", 6, 5, 3, 6, "forest", 11), ("scipy_m10d2", 6, 10, 2, 5, "forest", 12),
             ("scipy_m6d3_maze", 6, 6, 3, 5, "maze", 13)]
    for name, N, M, dim, n_obs, style, seed in specs:
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
        for _ in range(2):  # two replans so that v0, a0 != 0
            b = sw.build()
            X = np.zeros((N, dim * M * 6))
            for q in range(N):
                ag = O.make_agent(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q],
                                  next_waypoint=b["next_waypoint"][q], n_obs=sw.n_obs)
                X[q] = O.solve(cls, ag, np.ascontiguousarray(b["lsc"][q]), np.ascontiguousarray(b["sfc"][q]))["x"]
            sw.advance(X)
        b = sw.build()
        cases = []
        for q in range(N):
            ag = O.make_agent(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q],
                              next_waypoint=b["next_waypoint"][q], n_obs=sw.n_obs)
            lsc = np.ascontiguousarray(b["lsc"][q]); sfc = np.ascontiguousarray(b["sfc"][q])
            A = O.assemble(cls, ag, lsc, sfc)
            x0 = np.repeat(b["p0"][q][:dim], M * 6)
            cons = [LinearConstraint(A["Aeq"], A["beq"], A["beq"]), LinearConstraint(A["G"], -np.inf, A["h"])]
            res = minimize(lambda x: x @ A["P"] @ x + A["q"] @ x + A["r"], x0, jac=lambda x: 2 * A["P"] @ x + A["q"],
                           hess=lambda x: 2 * A["P"], method="trust-constr", constraints=cons,
                           bounds=Bounds(A["lb"], A["ub"]), options=dict(gtol=1e-12, xtol=1e-14, maxiter=3000))
            xs, objs, nact, _ = polish(A, res.x)
            # the polished point must still be feasible, otherwise the active-set guess was wrong
            feas = max((A["G"] @ xs - A["h"]).max(), (A["lb"] - xs).max(), (xs - A["ub"]).max())
            cases.append(dict(
                p0=b["p0"][q].tolist(), v0=b["v0"][q].tolist(), a0=b["a0"][q].tolist(), goal=b["goal"][q].tolist(),
                next_waypoint=b["next_waypoint"][q].tolist(),
                lsc_p=lsc["p"].tolist(), lsc_nrm=lsc["nrm"].tolist(), lsc_d=lsc["d"].tolist(),
                sfc_min=sfc["bmin"].tolist(), sfc_max=sfc["bmax"].tolist(),
                scipy_x=res.x.tolist(), scipy_obj=float(res.fun), x=xs.tolist(), obj=objs, n_active=nact,
                polish_feas=float(feas), scipy_vs_polish=float(np.abs(res.x - xs).max())))
            print(name, q, "scipy obj %.10f polished %.12f nact %d |dx| %.1e feas %.1e" % (res.fun, objs, nact, np.abs(res.x - xs).max(), feas))
        json.dump(dict(params=dict(M=M, dim=dim, n_obs=sw.n_obs, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0,
                                   world_min=sw.world_min.tolist(), world_max=sw.world_max.tolist(),
                                   vmax=[1.0] * 3, amax=[2.0] * 3, radius=0.15, nominal_velocity=1.0, planner_mode="LSC",
                                   use_sfc=True),
                       generator="tools/make_golden.py (scipy %s trust-constr + 80-bit active-set polish)" % __import__("scipy").__version__,
                       cases=cases), open(os.path.join(OUT, name + ".json"), "w"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    kat_log()
    scipy_cases()

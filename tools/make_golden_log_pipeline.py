"""Generates tests/golden/kat_log_pipeline.json in the BUILD container (reads /root/reference/log): the reference's OWN RUN replayed
replan by replan through the restated pipeline -- previous plans -> generateCLSC rows -> corridors over the forest10 world ->
GoalOptimizer -> trajectory QP -- and every replan whose logged motion the replay reproduces kept as a known answer.

What the log (log/simulation_1663743693.650981_LSC_10agents.csv, forest10_10, launch/simulation.launch parameters) gives: every agent's
state every 0.1 s.  What it does not give: the waypoint the grid planner / MAPF layer handed each agent (out of scope, SURVEY.md
section 2 row 8).  The replay therefore carries, per agent, exactly the state the reference's planner carries -- previous plan
(TrajPlanner::prev_traj), current goal point, corridor boxes -- and INFERS the waypoint of each replan: every lattice point of the
0.5 m waypoint grid near the agent is tried, the whole chain is evaluated for it

    initial / predicted trajectories  initialTrajPlanningPrevSol, obstaclePredictionWithPrevSol       src/traj_planner.cpp:273-310, 399-411
    who is in range                   MultiSyncSimulator::broadcastMsgs                               src/multi_sync_simulator.cpp:318-333
    LSC rows                          generateCLSC (the default launch)                               src/traj_planner.cpp:659-706
    corridor                          initializeSFC / constructSFCFromConvexHull                      src/collision_constraints.cpp:366-436
    goal                              GoalOptimizer::solve                                            src/goal_optimizer.cpp:7-165
    trajectory                        TrajOptimizer::solve (row-for-row model)                        src/traj_optimizer.cpp:216-514

and the candidate that reproduces the twelve logged numbers of t + 0.1 s and t + 0.2 s (p, v, a in x and y) is accepted: a few units
of the log's sixth digit for the right waypoint (the inputs are only known to six digits), 1e3 .. 1e5 units for a wrong one.  An
accepted replan updates the agent's state (plan, goal, corridor) for the next one; a replan without an acceptable candidate ends that
agent's chain (its state is unknown from then on, and so is the neighbourhood of every agent that has it in range).

Kept in the fixture:
  replay  one line per (replan, agent), all of them: inferred waypoint, goal after the goal LP, its step t, neighbours in range, the
          match in units of the log's sixth digit, largest LSC / corridor multiplier.  With tests/golden/sim_log_states.json (the
          logged states) this is what the DEVICE replay test feeds the HIP pipeline with (tools/closed_loop.py `script=`).
  cases   a bounded number of self-contained replans (inputs of the goal LP and of the QP, expected states): the ones with the
          strongest ACTIVE LSC rows, the strongest active corridor faces, a moved goal, and a few plain ones.
Takes about nine minutes of CPU.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "tests", "golden", "kat_log_pipeline.json")
MATCH = 400.0  # units of the log's sixth digit: accepted up to 150 outright, up to 400 if every different answer misses by 10x more


def ulp6(v):
    return 10.0 ** (np.floor(np.log10(abs(v))) - 5) if v != 0 else 1e-6


def f32list(a):
    """float32 values as the shortest decimals that round-trip through float32 (read back with np.float32)"""
    a = np.asarray(a)
    return np.array([float(np.format_float_positional(np.float32(v), unique=True)) for v in a.reshape(-1)]).reshape(a.shape).tolist()


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
    W = json.load(open(os.path.join(HERE, "tests", "golden", "forest10_world.json")))
    M, dim, R, dt, radius, z2d = 10, 2, 3.0, 0.2, W["radius"], W["z_2d"]
    p = dict(M=M, n=5, phi=3, dim=dim, dt=dt, w_c=0.01, w_t=1.0, comm_range=R, world_z_2d=z2d, world_min=W["world_min"], world_max=W["world_max"],
             vmax=[1.0, 1.0, 1.0], amax=[2.0, 2.0, 2.0], radius=radius, nominal_velocity=1.0, planner_mode="LSC", use_sfc=True, downwash=2.0)
    cls = O.make_class(M=M, dim=dim, dt=dt, w_c=0.01, w_t=1.0, comm_range=R, planner_lsc=True, use_sfc=True, world_min=W["world_min"], world_max=W["world_max"])
    omap = O.Map(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    starts = np.array(W["starts"], dtype=np.float64)

    alive = [True] * nag
    plan = [None] * nag                          # previous plan (M, 6, 3), float32 values
    goal = [starts[a].copy() for a in range(nag)]  # agent.current_goal_point (AgentManager ctor: the start position)
    sfc = [np.zeros((1, M), O.BOX_DTYPE) for _ in range(nag)]
    n_replans = (len(T[0]) - 2) // 2
    kept, table, stats = [], [], dict(tried=0, matched=0, lsc_active=0, sfc_active=0, chains_ended={})

    def hover(pos):
        tr = np.zeros((M, 6, 3))
        tr[:] = np.float32(pos)
        return tr.astype(np.float64)

    for k in range(n_replans):
        pos = np.float32([T[a][2 * k]["p"] for a in range(nag)])
        # trajectories every planner works with in this replan: shifted previous plans (k >= 1), hover at k = 0 (planner_seq < 2)
        traj = np.zeros((nag, M, 6, 3))
        for j in range(nag):
            if k == 0 or plan[j] is None:
                traj[j] = hover(pos[j])
            else:
                traj[j, :-1] = plan[j][1:]
                traj[j, -1] = plan[j][-1, 5]
        goal_prev = np.array(goal)
        new = {}
        for a in range(nag):
            if not alive[a]:
                continue
            st, s1, s2 = T[a][2 * k], T[a][2 * k + 1], T[a][2 * k + 2]
            d = np.abs(pos - pos[a]).astype(np.float64).max(axis=1)
            nbr = [j for j in range(nag) if j != a and d[j] <= R]
            if any(not alive[j] for j in nbr):
                alive[a] = False
                stats["chains_ended"][a] = "replan %d: a neighbour's state is unknown" % k
                continue
            stats["tried"] += 1
            L = None
            if nbr:
                L = O.generate_constraints(O.MODE_CLSC, traj, np.array([nbr], dtype=np.int32), radius, p["downwash"], goal_prev, dim=dim, first_agent=a)[0]
            p0 = np.array(st["p"])
            base = np.round((p0[:2] - starts[a][:2]) / 0.5)
            best, errs = None, []
            for dx in range(-3, 4):
                for dy in range(-3, 4):
                    w = np.array([*(starts[a][:2] + 0.5 * (base + np.array([dx, dy]))), z2d])
                    box = sfc[a].copy()
                    if k == 0:
                        sst = omap.construct_sfc(O.SFC_INIT, np.array([[p0, p0, p0]]), radius, box)
                        if sst[0] != 1:
                            continue
                    else:
                        omap.construct_sfc(O.SFC_FROM_HULL, np.array([[traj[a][-1, 5], goal_prev[a], w]]), radius, box)
                    gst, g_new, tpar = O.goal_opt(cls, goal_prev[a], w, lsc=L, sfc_last=box[0, M - 1])
                    if gst != 0:
                        continue
                    g_new = np.float32(g_new).astype(np.float64)
                    ag = O.make_agent(p0=[p0[0], p0[1], z2d], v0=st["v"], a0=st["a"], goal=g_new, next_waypoint=w, vmax=p["vmax"], amax=p["amax"],
                                      radius=radius, nominal_velocity=1.0, n_obs=len(nbr))
                    Rs = O.solve(cls, ag, L, box[0])
                    if Rs["status"] != 0:
                        continue
                    err = 0.0
                    for sl in (s1, s2):
                        ps, vl, ac = O.state_at(cls, Rs["x"], sl["t"] - st["t"])
                        for got, logged in ((ps, sl["p"]), (vl, sl["v"]), (ac, sl["a"])):
                            for gk, lk in zip(got[:2], logged[:2]):
                                err = max(err, abs(gk - lk) / max(ulp6(lk), 1e-6))
                    errs.append(err)
                    if best is None or err < best[0]:
                        best = (err, w, g_new, tpar, box, Rs, ag)
            # runner-up: the best among the candidates that are not the same answer (several waypoints can lead to one QP, e.g. when
            # GoalOptimizer clips the goal to the same corridor face)
            second = min([e for e in errs if best is not None and e > 1.05 * best[0] + 1.0] or [1e30])
            if best is None or best[0] > MATCH or (best[0] > 150.0 and second < 10.0 * best[0]):
                alive[a] = False
                stats["chains_ended"][a] = "replan %d: best candidate %s units, runner-up %.0f" % (k, "none" if best is None else "%.0f" % best[0], second)
                continue
            err, w, g_new, tpar, box, Rs, ag = best
            stats["matched"] += 1
            sz = O.count(cls, ag, L)
            lam = Rs["lam"]
            lam_sfc = float(lam[:sz.n_sfc].max()) if sz.n_sfc else 0.0
            lam_lsc = float(lam[sz.n_sfc:sz.n_sfc + sz.n_lsc].max()) if sz.n_lsc else 0.0
            stats["lsc_active"] += lam_lsc > 1e-6
            stats["sfc_active"] += lam_sfc > 1e-6
            x = Rs["x"].reshape(dim, M, 6)
            tr = np.zeros((M, 6, 3))
            tr[..., 0], tr[..., 1], tr[..., 2] = x[0], x[1], z2d
            new[a] = (np.float32(tr).astype(np.float64), g_new, box)
            table.append(dict(replan=k, agent=a, waypoint=[float(w[0]), float(w[1])], goal=[float(g_new[0]), float(g_new[1])], goal_lp_t=round(tpar, 9),
                              neighbours=nbr, match=round(err, 1), lam_lsc=float("%.3e" % lam_lsc), lam_sfc=float("%.3e" % lam_sfc)))
            if True:
                kept.append(dict(agent=a, replan=k, t=st["t"], p0=[p0[0], p0[1], z2d], v0=st["v"], a0=st["a"], goal_before_lp=goal_prev[a].tolist(),
                                 goal=g_new.tolist(), goal_lp_t=tpar, next_waypoint=w.tolist(), neighbours=nbr,
                                 lsc_p=None if L is None else f32list(L["p"]), lsc_nrm=None if L is None else f32list(L["nrm"]),
                                 lsc_d=None if L is None else L["d"].tolist(), sfc_min=f32list(box[0]["bmin"]), sfc_max=f32list(box[0]["bmax"]),
                                 states=[s1, s2], match_units_of_6th_digit=round(err, 2), max_lsc_multiplier=lam_lsc, max_sfc_multiplier=lam_sfc,
                                 oracle_obj=Rs["obj"]))
            print("replan %2d agent %d: %d neighbours, match %6.1f units (runner-up %.0f), goal-LP t %.4f, multipliers LSC %.2e SFC %.2e" % (
                k, a, len(nbr), err, second, tpar, lam_lsc, lam_sfc), flush=True)
        for a, (tr, g_new, box) in new.items():
            plan[a], goal[a], sfc[a] = tr, g_new, box
        if not any(alive):
            break
    stats["lsc_active"], stats["sfc_active"] = int(stats["lsc_active"]), int(stats["sfc_active"])
    print(json.dumps(stats))
    # bounded choice of self-contained cases: cost of a case ~ its number of neighbours
    def top(key, n, pred=lambda c: True):
        return sorted([c for c in kept if pred(c)], key=key, reverse=True)[:n]

    few = lambda c: len(c["neighbours"]) <= 4  # noqa: E731
    chosen = {}
    for c in (top(lambda c: c["max_lsc_multiplier"], 8, few) + top(lambda c: c["max_lsc_multiplier"], 10) + top(lambda c: c["max_sfc_multiplier"], 10, few)
              + top(lambda c: c["goal_lp_t"], 4, few) + top(lambda c: c["match_units_of_6th_digit"], 3, few)
              + [c for c in kept if c["replan"] in (0, 1, 40) and c["agent"] in (0, 5)]):
        chosen[(c["replan"], c["agent"])] = c
    kept = [chosen[k] for k in sorted(chosen)]
    json.dump(dict(source="reference log/simulation_1663743693.650981_LSC_10agents.csv replayed through the restated pipeline; see tools/make_golden_log_pipeline.py",
                   params=p, stats=stats, replay=table, cases=kept), open(OUT, "w"), separators=(",", ":"))
    print("kat_log_pipeline.json: %d cases kept (%d matched of %d tried)" % (len(kept), stats["matched"], stats["tried"]))


if __name__ == "__main__":
    main()

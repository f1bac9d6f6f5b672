"""Development aid (NOT product): Goldfarb-Idnani DUAL ACTIVE SET on the kernel's reduced QP  min 1/2 z'Hz + g'z  s.t. Gz >= h,
numpy prototype on the bench's batches (tools/proto_tail.py).  H is the class's constant matrix (no barrier): H^-1 is tabulated, the
unconstrained optimum is one product, and every added row costs a rank-one step -- the question is how MANY rows a plan's optimum holds.

usage: python tools/proto_gi.py c1 [replans]"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import proto_pdip as PP  # noqa: E402
from tools import proto_tail as PT  # noqa: E402


def gi(H, g, G, h, tol=1e-9, max_steps=200, pick="most", verbose=False):
    """returns z, u (multipliers on all rows), steps (adds + drops), status (0 ok, 1 infeasible, 2 step limit)"""
    nz, m = len(g), len(h)
    Hi = np.linalg.inv(H)
    z = -Hi @ g
    A = []          # active rows
    u = np.zeros(0)
    steps = adds = drops = 0
    gn = np.linalg.norm(G, axis=1)
    while True:
        s = G @ z - h
        sv = s / np.maximum(gn, 1e-300)
        sv[A] = 0.0  # active rows sit at zero by construction
        p = int(np.argmin(sv))
        if sv[p] >= -tol:
            ufull = np.zeros(m); ufull[A] = u
            return z, ufull, steps, adds, drops, 0
        npv = G[p]
        up = 0.0
        while True:  # partial steps until p is added
            steps += 1
            if steps > max_steps:
                ufull = np.zeros(m); ufull[A] = u
                return z, ufull, steps, adds, drops, 2
            yp = Hi @ npv
            if A:
                N = G[A]
                Y = Hi @ N.T
                S = N @ Y
                r = np.linalg.solve(S, N @ yp)
                zd = yp - Y @ r
            else:
                r = np.zeros(0)
                zd = yp
            curv = npv @ zd
            sp = npv @ z - h[p]
            t2 = -sp / curv if curv > 1e-13 * (npv @ yp) else np.inf
            t1, l = np.inf, -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1:
                    t1, l = u[j] / r[j], j
            t = min(t1, t2)
            if verbose:
                print("   p %d sp %.2e t1 %.2e t2 %.2e |A| %d" % (p, sp, t1, t2, len(A)))
            if not np.isfinite(t):
                ufull = np.zeros(m); ufull[A] = u
                return z, ufull, steps, adds, drops, 1
            if np.isfinite(t2):
                z = z + t * zd
            u = u - t * r
            up += t
            if t == t2:
                A.append(p); u = np.append(u, up); adds += 1
                break
            A.pop(l); u = np.delete(u, l); drops += 1


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
    replans = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seed, N, M, n_obs, dim, style = PT.CFG[cfg]
    B = PT.batches(cfg, replans)
    for bi, (b, wmin, wmax) in enumerate(B):
        st, ad, dr, nact, dxs, dos, its, bad = [], [], [], [], [], [], [], 0
        for q in range(N):
            pr = PT.solve_one(b, q, M, dim, wmin, wmax, return_problem=True)
            x, obj, status, it = PT.solve_one(b, q, M, dim, wmin, wmax)
            z, u, steps, adds, drops, stat = gi(pr["K"], pr["g"], pr["G"], pr["h"])
            nzA = pr["nzA"]
            xg = np.concatenate([pr["cfix"][k] + pr["T"] @ z[k * nzA:(k + 1) * nzA] for k in range(dim)]) + np.repeat(pr["org"][:dim], 6 * M)
            if stat != 0 or status != 0:
                bad += 1
                print("   q %d gi status %d pdip status %d steps %d" % (q, stat, status, steps))
                continue
            st.append(steps); ad.append(adds); dr.append(drops); nact.append(int((u > 0).sum())); its.append(it)
            dxs.append(np.abs(xg - x).max())
            rd = np.abs(pr["K"] @ z + pr["g"] - pr["G"].T @ u).max() / max(1.0, np.abs(pr["g"]).max())
            dos.append(rd)
        st = np.array(st)
        print("%s batch %d: GI steps mean %.2f max %d hist %s | adds mean %.2f drops mean %.2f | active at optimum max %d | PDIP iters mean %.2f max %d | max|dx| vs PDIP %.1e  max stationarity %.1e  bad %d" % (
            cfg, bi, st.mean(), st.max(), np.bincount(st).tolist(), np.mean(ad), np.mean(dr), max(nact), np.mean(its), max(its), max(dxs), max(dos), bad))

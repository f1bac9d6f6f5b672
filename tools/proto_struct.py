"""Development aid: checks the STRUCTURED assembly the HIP kernel uses against a dense reference.

The kernel never forms G.  It keeps, per iteration,
  S[cp][k][l]   sum of w * n n^T over the LSC rows of control point cp            (3x3 per cp)
  om[k][t]      w_lo + w_hi of the two-sided per-axis rows (interval / vel / acc / comm-pair)
and builds the reduced Hessian  Hred = T'(H + G'WG)T  lane by lane from per-(axis,segment) local 6x6 blocks.
Same for right-hand sides: per-cp vectors V[cp][k] and per-row scalars rho[k][t] -> x-space gradient -> T' gather.
This file re-derives both with explicit "lane" loops (mirroring the kernel's index arithmetic) and compares with
dense matrix algebra on random weights.
"""
import numpy as np

from tools.proto_pdip import TB, build_T, q_base


class Layout:
    def __init__(self, M, dim, end_stop):
        self.M, self.dim, self.es = M, dim, end_stop
        self.P = 6 * M
        self.CP = self.P - 3
        self.nzA = 3 * (M - 1) + (1 if end_stop else 3)
        self.nz = dim * self.nzA
        # two-sided per-axis rows, in kernel order: intervals [cpi], vel, acc, comm pairs
        self.NI = self.CP
        self.NV = 5 * M - 2
        self.NA = 4 * M - 1
        self.NC = M * (M - 1) // 2
        self.NRA = self.NI + self.NV + self.NA + self.NC

    # ---- row index helpers (exactly what the kernel computes with integer arithmetic) ----
    def iv(self, m, i):  # interval row of cp (m,i); requires (m,i) != (0,<3)
        return 6 * m + i - 3

    def vel(self, m, i):  # c[m][i+1]-c[m][i]; m==0 -> i>=2
        return self.NI + (5 * m + i - 2)

    def acc(self, m, i):  # c[m][i+2]-2c[m][i+1]+c[m][i]; m==0 -> i>=1
        return self.NI + self.NV + (4 * m + i - 1)

    def comm(self, u, up):  # c5 of segment u minus c5 of segment up (== c0 of segment up+1), up < u
        return self.NI + self.NV + self.NA + (u * (u - 1) // 2 + up)

    def zidx(self, m, j):  # per-axis z index of free control point (m, 3+j)
        if self.es and m == self.M - 1:
            return 3 * (self.M - 1)
        return 3 * m + j

    def row_vector(self, t):
        """dense per-axis P-vector of two-sided row t (for the dense reference only)."""
        e = np.zeros(self.P)
        if t < self.NI:
            e[t + 3] = 1
        elif t < self.NI + self.NV:
            v = t - self.NI + 2
            m, i = divmod(v, 5)
            e[6 * m + i + 1], e[6 * m + i] = 1, -1
        elif t < self.NI + self.NV + self.NA:
            a = t - self.NI - self.NV + 1
            m, i = divmod(a, 4)
            e[6 * m + i + 2], e[6 * m + i + 1], e[6 * m + i] = 1, -2, 1
        else:
            c = t - self.NI - self.NV - self.NA
            u = 1
            while u * (u + 1) // 2 <= c:
                u += 1
            up = c - u * (u - 1) // 2
            e[6 * u + 5] += 1
            e[6 * (up + 1) + 0] += -1
        return e


def local_block(L, k, m, om, S, Q2, wt2, ts):
    """6x6 x-space block of (axis k, segment m): 2 w_c Q + terminal + interval/vel/acc weights + LSC same-axis."""
    B = Q2.copy()
    if m >= L.M - ts:
        B[5, 5] += wt2
    for i in range(6):
        if m == 0 and i < 3:
            continue
        B[i, i] += om[k][L.iv(m, i)] + S[6 * m + i - 3][k][k]
    for i in range(5):
        if m == 0 and i < 2:
            continue
        w = om[k][L.vel(m, i)]
        B[i, i] += w; B[i + 1, i + 1] += w; B[i, i + 1] -= w; B[i + 1, i] -= w
    for i in range(4):
        if m == 0 and i < 1:
            continue
        w = om[k][L.acc(m, i)]
        d = np.zeros(6); d[i], d[i + 1], d[i + 2] = 1, -2, 1
        B += w * np.outer(d, d)
    return B


def assemble_structured(L, om, S, Q2, wt2, ts):
    """Row r = (k, a) of Hred, computed 'per lane'."""
    M, dim, nzA = L.M, L.dim, L.nzA
    H = np.zeros((L.nz, L.nz))
    for k in range(dim):
        for a in range(nzA):
            r = k * nzA + a
            last = L.es and a == 3 * (M - 1)
            m = M - 1 if last else a // 3
            js = [0, 1, 2] if last else [a % 3]  # cps (m, 3+j) this variable drives directly
            Bm = local_block(L, k, m, om, S, Q2, wt2, ts)
            # --- same axis, own segment: columns (m, j')
            for jp in range(3):
                cidx = k * nzA + L.zidx(m, jp)
                H[r, cidx] += sum(Bm[3 + j, 3 + jp] for j in js)
            # --- same axis, previous segment columns (m-1, j'): c[m][i] = sum_j' TB[i][j'] z(m-1,j')
            if m >= 1:
                for jp in range(3):
                    cidx = k * nzA + L.zidx(m - 1, jp)
                    H[r, cidx] += sum(TB[i, jp] * Bm[i, 3 + j] for i in range(3) for j in js)
            # --- next segment: this variable also drives c[m+1][i] = sum TB[i][j] z(m,j)
            if m + 1 < M:
                Bn = local_block(L, k, m + 1, om, S, Q2, wt2, ts)
                j = js[0]
                for jp in range(3):  # columns (m, j') via Baa
                    cidx = k * nzA + L.zidx(m, jp)
                    H[r, cidx] += sum(TB[i, j] * TB[ip, jp] * Bn[i, ip] for i in range(3) for ip in range(3))
                for jp in range(3):  # columns (m+1, j') via Bab
                    cidx = k * nzA + L.zidx(m + 1, jp)
                    H[r, cidx] += sum(TB[i, j] * Bn[i, 3 + jp] for i in range(3))
            # --- cross axis (LSC only): block-diagonal in the segment index
            for l in range(dim):
                if l == k:
                    continue
                for jp in range(3):
                    cidx = l * nzA + L.zidx(m, jp)
                    for j in js:
                        if j == jp:
                            H[r, cidx] += S[6 * m + 3 + j - 3][k][l]
                    if m + 1 < M:
                        j = js[0]
                        H[r, cidx] += sum(TB[i, j] * TB[i, jp] * S[6 * (m + 1) + i - 3][k][l] for i in range(3))
            # --- comm pairs: only the c5 variables
            if last or a % 3 == 2:
                u = m
                for up in range(M):
                    if up == u:
                        continue
                    w = om[k][L.comm(max(u, up), min(u, up))]
                    H[r, r] += w
                    H[r, k * nzA + L.zidx(up, 2)] -= w
    return H


def gather_structured(L, V, rho):
    """x-space gradient Gx[k][cp] from per-cp vectors V[cpi][k] and row scalars rho[k][t], then T' gather."""
    M, dim, nzA = L.M, L.dim, L.nzA
    Gx = np.zeros((dim, L.P))
    for k in range(dim):
        for m in range(M):
            for i in range(6):
                if m == 0 and i < 3:
                    # fixed control points still receive vel/acc row contributions but are never gathered
                    continue
                g = V[6 * m + i - 3][k] + rho[k][L.iv(m, i)]
                # vel rows: row (m,i-1) has +1 on cp i, row (m,i) has -1 on cp i
                if i >= 1 and not (m == 0 and i - 1 < 2):
                    g += rho[k][L.vel(m, i - 1)]
                if i <= 4 and not (m == 0 and i < 2):
                    g -= rho[k][L.vel(m, i)]
                # acc rows: (m,i-2): +1, (m,i-1): -2, (m,i): +1
                if i >= 2 and not (m == 0 and i - 2 < 1):
                    g += rho[k][L.acc(m, i - 2)]
                if 1 <= i <= 4 and not (m == 0 and i - 1 < 1):
                    g -= 2 * rho[k][L.acc(m, i - 1)]
                if i <= 3 and not (m == 0 and i < 1):
                    g += rho[k][L.acc(m, i)]
                if i == 5:
                    for up in range(m):
                        g += rho[k][L.comm(m, up)]
                if i == 0 and m >= 1:  # cp (m,0) is the "minus" end of pairs (u, up=m-1), u >= m
                    for u in range(m, M):
                        g -= rho[k][L.comm(u, m - 1)]
                Gx[k, 6 * m + i] = g
    out = np.zeros(L.nz)
    for k in range(dim):
        for a in range(nzA):
            last = L.es and a == 3 * (M - 1)
            m = M - 1 if last else a // 3
            js = [0, 1, 2] if last else [a % 3]
            v = sum(Gx[k, 6 * m + 3 + j] for j in js)
            if m + 1 < M:
                j = js[0]
                v += sum(TB[i, j] * Gx[k, 6 * (m + 1) + i] for i in range(3))
            out[k * nzA + a] = v
    return out


def check(M=5, dim=3, es=True, n_obs=4, seed=0):
    rng = np.random.default_rng(seed)
    L = Layout(M, dim, es)
    T, nzA = build_T(M, es)
    assert nzA == L.nzA
    dt, w_c, w_t, ts = 0.2, 0.01, 1.0, 2
    Q2 = 2 * w_c * q_base(dt)
    om = rng.random((dim, L.NRA)) * 10
    rho = rng.standard_normal((dim, L.NRA))
    nrm = rng.standard_normal((n_obs, L.CP, 3))
    if dim == 2:
        nrm[..., 2] = 0
    w = rng.random((n_obs, L.CP)) * 5
    qv = rng.standard_normal((n_obs, L.CP))
    S = np.einsum("oc,ock,ocl->ckl", w, nrm, nrm)
    V = np.einsum("oc,ock->ck", qv, nrm)
    # dense reference
    Hx = np.zeros((dim * L.P, dim * L.P))
    gx = np.zeros(dim * L.P)
    for k in range(dim):
        for m in range(M):
            sl = slice(k * L.P + 6 * m, k * L.P + 6 * m + 6)
            Hx[sl, sl] += Q2
            if m >= M - ts:
                Hx[k * L.P + 6 * m + 5, k * L.P + 6 * m + 5] += 2 * w_t
        for t in range(L.NRA):
            e = np.zeros(dim * L.P); e[k * L.P:(k + 1) * L.P] = L.row_vector(t)
            Hx += om[k, t] * np.outer(e, e)
            gx += rho[k, t] * e
    for o in range(n_obs):
        for c in range(L.CP):
            e = np.zeros(dim * L.P)
            for k in range(dim):
                e[k * L.P + c + 3] = nrm[o, c, k]
            Hx += w[o, c] * np.outer(e, e)
            gx += qv[o, c] * e
    Tf = np.zeros((dim * L.P, L.nz))
    for k in range(dim):
        Tf[k * L.P:(k + 1) * L.P, k * nzA:(k + 1) * nzA] = T
    Hd = Tf.T @ Hx @ Tf
    gd = Tf.T @ gx
    Hs = assemble_structured(L, om, S, Q2, 2 * w_t, ts)
    gs = gather_structured(L, V, rho)
    eh = np.abs(Hs - Hd).max() / np.abs(Hd).max()
    eg = np.abs(gs - gd).max() / max(1, np.abs(gd).max())
    return eh, eg


if __name__ == "__main__":
    for (M, dim, es) in [(5, 3, True), (5, 3, False), (6, 3, True), (10, 2, True), (2, 3, True), (3, 2, False)]:
        print(M, dim, es, check(M, dim, es))

"""Golden vectors for closestPointsBetweenLineSegments (reference include/geometry.hpp:174-263): the TRUE closest pair of
two 3-D segments, computed independently of the oracle -- exact enumeration of the KKT cases in fp64 (interior optimum,
the four edges of the (s,t) square), confirmed by scipy's bounded-variable least squares (lsq_linear, BVLS: an exact
active-set method for min |A [s t]' - b|, 0 <= s,t <= 1).  The reference's
routine is a float32 procedure that must agree with it wherever it is exact; tests/test_lscmode.py checks the oracle's
restatement against these vectors.  octomap/Eigen are absent, so the reference routine itself cannot be compiled here.

    python tools/make_golden_segseg.py   ->  tests/golden/segseg.json
"""
import json
import os

import numpy as np
from scipy.optimize import lsq_linear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def true_segseg(p1, q1, p2, q2):
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    best = [np.inf, None, None]

    def cand(s, t):
        c1, c2 = p1 + s * d1, p2 + t * d2
        d = np.linalg.norm(c1 - c2)
        if d < best[0]:
            best[:] = [d, c1, c2]

    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    b, c = d1 @ d2, d1 @ r
    den = a * e - b * b
    if den > 1e-14 * max(a * e, 1e-300):
        s, t = (b * f - c * e) / den, (a * f - b * c) / den
        if 0 <= s <= 1 and 0 <= t <= 1:
            cand(s, t)
    for s in (0.0, 1.0):
        cand(s, np.clip(((p1 + s * d1 - p2) @ d2) / e, 0, 1) if e > 0 else 0.0)
    for t in (0.0, 1.0):
        cand(np.clip(((p2 + t * d2 - p1) @ d1) / a, 0, 1) if a > 0 else 0.0, t)
    return best


def main():
    rng = np.random.default_rng(20260928)
    cases = []
    for k in range(240):
        P = rng.uniform(-3, 3, (4, 3))
        kind = "generic"
        if k % 5 == 1:  # planar mission (dim 2): all z equal
            P[:, 2] = 0.6
            kind = "planar"
        if k % 12 == 2:  # parallel segments, overlapping / disjoint / reversed
            P[3] = P[2] + (P[1] - P[0]) * rng.uniform(-2, 2)
            kind = "parallel"
        if k % 12 == 3:  # agent already at its goal: degenerate segment 2
            P[3] = P[2]
            kind = "degenerate2"
        if k % 12 == 4:  # neighbour at its goal: degenerate segment 1
            P[1] = P[0]
            kind = "degenerate1"
        if k % 12 == 5:  # crossing segments in the plane: distance 0
            P[:, 2] = 0.6
            mid = 0.5 * (P[0] + P[1])
            dirn = rng.normal(size=3)
            dirn[2] = 0
            P[2], P[3] = mid - dirn, mid + dirn
            kind = "crossing"
        P = np.float32(P).astype(np.float64)  # float32-representable like the reference's point3d
        d, c1, c2 = true_segseg(*P)
        A = np.stack([P[1] - P[0], -(P[3] - P[2])], axis=1)
        if kind not in ("degenerate1", "degenerate2"):
            r = lsq_linear(A, P[2] - P[0], bounds=(0, 1), method="bvls", tol=1e-14)
            assert abs(np.linalg.norm(A @ r.x - (P[2] - P[0])) - d) < 1e-9, (k, kind, r, d)
        cases.append({"kind": kind, "l1s": P[0].tolist(), "l1e": P[1].tolist(), "l2s": P[2].tolist(), "l2e": P[3].tolist(),
                      "dist": float(d), "cp1": c1.tolist(), "cp2": c2.tolist()})
    out = {"source": "tools/make_golden_segseg.py: exact fp64 closest pair of two segments (KKT case enumeration), confirmed by scipy "
                     "lsq_linear (BVLS); inputs float32-representable",
           "cases": cases}
    with open(os.path.join(ROOT, "tests", "golden", "segseg.json"), "w") as f:
        json.dump(out, f)
    print(len(cases), "cases;", {k: sum(c["kind"] == k for c in cases) for k in sorted({c["kind"] for c in cases})})


if __name__ == "__main__":
    main()

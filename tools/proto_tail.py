"""Development aid (NOT product): iteration TAIL experiments on the bench's own batches (numpy prototype, tools/proto_pdip.py).

The batch of a bench config = synth.Swarm(seed of bench.CONFIGS) after 3 warm-up replans; cached under /tmp/pt.  Every variant solves the
SAME batch (the timed one) and, with REPLANS > 1, the following replans of a swarm advanced with the baseline's plans.

usage: python tools/proto_tail.py c1 [replans]      env: VARIANTS=name,name   VERBOSE=q (trace of instance q)"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import synth  # noqa: E402
from tools import proto_pdip as PP  # noqa: E402

CFG = {"c0": (3020, 10, 10, 9, 2, "forest"), "c1": (1000, 64, 5, 20, 3, "forest"), "c2": (3518, 512, 6, 20, 3, "maze"),
       "c3s": (3138, 128, 10, 40, 3, "forest"), "c4s": (7101, 256, 5, 20, 3, "forest")}


def rows_of(b, q):
    L = b["lsc"][q]
    r = np.zeros(L.shape + (4,))
    r[..., :3] = L["nrm"]
    r[..., 3] = L["d"] + (L["nrm"] * L["p"]).sum(-1)
    return r


def solve_one(b, q, M, dim, sw_min, sw_max, warm=True, **kw):
    hdr = dict(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q], vmax=[1.0] * 3, amax=[2.0] * 3,
               radius=0.15, init=b["init"][q] if warm else None)
    d = np.linalg.norm(np.float32(b["goal"][q]) - np.float32(b["p0"][q]))
    ts = min(M, max(int((M * 0.2 - d / 1.0 + 1e-9) / 0.2), 1))
    sfc = np.stack([b["sfc"][q]["bmin"], b["sfc"][q]["bmax"]], axis=1)
    return PP.solve(M, dim, 0.2, 0.01, 1.0, 3.0, True, True, sw_min, sw_max, hdr, rows_of(b, q), sfc, ts, nbr_ids=b["nbr"][q], **kw)


def batches(cfg, replans):
    seed, N, M, n_obs, dim, style = CFG[cfg]
    path = "/tmp/pt/%s_%d.pkl" % (cfg, replans)
    if os.path.exists(path):
        return pickle.load(open(path, "rb"))
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    out = []
    for step in range(3 + replans):
        b = sw.build()
        if step >= 3:
            out.append((b, np.array(sw.world_min), np.array(sw.world_max)))
        X = np.zeros((N, dim * M * 6))
        for q in range(N):
            x, obj, st, it = solve_one(b, q, M, dim, sw.world_min, sw.world_max, warm=step > 0)
            if st != 0:
                x = np.concatenate([np.asarray(b["init"], float)[q, :, :, k].reshape(-1) for k in range(dim)])
            X[q] = x
        sw.advance(X)
    pickle.dump(out, open(path, "wb"))
    return out


VARIANTS = {
    "base": {},
    "g1": dict(gondzio=1),
    "g2": dict(gondzio=2),
    "tol9": dict(tol=1e-9),
    "tol8": dict(tol=1e-8),
}

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
    replans = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seed, N, M, n_obs, dim, style = CFG[cfg]
    B = batches(cfg, replans)
    names = os.environ.get("VARIANTS", "base").split(",")
    vq = os.environ.get("VERBOSE")
    ref = None
    for name in names:
        kw = VARIANTS[name] if name in VARIANTS else eval("dict(%s)" % name.replace(";", ","))
        its, xs, objs, bad = [], [], [], 0
        PP.EXTRA_SOLVES = 0
        for (b, wmin, wmax) in B:
            row = []
            for q in range(N):
                if vq is not None and int(vq) != q:
                    continue
                x, obj, st, it = solve_one(b, q, M, dim, wmin, wmax, verbose=vq is not None, **kw)
                bad += st != 0
                row.append(it); xs.append(x); objs.append(obj)
            its.append(row)
        h = np.array(its)
        xs = np.array([x if x is not None else np.zeros(dim * M * 6) for x in xs]); objs = np.array(objs)
        if ref is None:
            ref = (xs, objs)
        print("%-40s mean %.3f  extra solves/QP %.3f  per-replan max %s  hist %s  bad %d  max|dx| %.1e  max rel dobj %.1e" % (
            name[:40], h.mean(), PP.EXTRA_SOLVES / h.size, h.max(axis=1).tolist(), np.bincount(h.reshape(-1)).tolist(), bad,
            np.abs(xs - ref[0]).max(), (np.abs(objs - ref[1]) / np.maximum(1, np.abs(ref[1]))).max()))
        if name == names[0] and vq is None:
            print("   slowest of the first batch:", np.argsort(-h[0])[:8].tolist(), np.sort(-h[0])[:8].tolist())

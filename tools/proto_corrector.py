"""Development aid (NOT product): weight of Mehrotra's second-order term in the corrector -- 1 (round 1-4 kernel), alpha_aff, alpha_aff^2 --
on the bench's own workload generator, numpy prototype (tools/proto_pdip.py).  Every variant replans the same swarm states (the swarm
advances with the FIRST variant's plans; a failed QP keeps its start), cold first replan, warm afterwards unless COLD=1.

usage: python tools/proto_corrector.py N M dim n_obs replans seed [end_stop=1]      env: STYLE=forest|maze, COLD=1, VARIANTS=1,aff,aff2"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import synth  # noqa: E402
from tools import proto_pdip as PP  # noqa: E402

N, M, dim, n_obs, replans, seed, es = [int(v) for v in (sys.argv[1:] + ["64", "5", "3", "20", "4", "1000", "1"][len(sys.argv) - 1:])]
style = os.environ.get("STYLE", "forest")
cold = os.environ.get("COLD", "0") == "1"
variants = [None if v == "1" else v for v in os.environ.get("VARIANTS", "1,aff,aff2").split(",")]
variants = [v.replace(";", ",") if v else v for v in variants]


def rows_of(b, q):
    L = b["lsc"][q]
    r = np.zeros(L.shape + (4,))
    r[..., :3] = L["nrm"]
    r[..., 3] = L["d"] + (L["nrm"] * L["p"]).sum(-1)
    return r


sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
its = {v: [] for v in variants}
bad = {v: 0 for v in variants}
worst = {v: 0.0 for v in variants}
for step in range(replans):
    b = sw.build()
    for v in variants:
        xs = []
        for q in range(N):
            hdr = dict(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q], vmax=[1.0] * 3, amax=[2.0] * 3,
                       radius=0.15, init=b["init"][q] if (step > 0 and not cold) else None)
            d = np.linalg.norm(np.float32(b["goal"][q]) - np.float32(b["p0"][q]))
            ts = min(M, max(int((M * 0.2 - d / 1.0 + 1e-9) / 0.2), 1))
            sfc = np.stack([b["sfc"][q]["bmin"], b["sfc"][q]["bmax"]], axis=1)
            x, obj, status, it = PP.solve(M, dim, 0.2, 0.01, 1.0, 3.0, bool(es), True, sw.world_min, sw.world_max, hdr, rows_of(b, q), sfc, ts, nbr_ids=b["nbr"][q],
                                          corr_weight=v)
            if status == 0:
                its[v].append(it)
            elif status == 2:
                bad[v] += 1
            xs.append((x, obj, status))
        if v is variants[0]:
            first = xs
        else:
            for (x, o, st), (x0, o0, st0) in zip(xs, first):
                if st == 0 and st0 == 0:
                    worst[v] = max(worst[v], float(np.abs(x - x0).max()))
    init = np.asarray(b["init"], float)  # (N, M, 6, 3); a failed QP keeps its start
    Xadv = np.stack([x if st == 0 else np.concatenate([init[q, :, :, k].reshape(-1) for k in range(dim)]) for q, (x, o, st) in enumerate(first)])
    sw.advance(Xadv)
for v in variants:
    h = np.array(its[v])
    print("weight %-5s solved %4d  iteration-limit %2d  mean %.3f  max %2d  p95 %2d  hist %s  max|x - x(weight 1)| %.1e" % (
        v or "1", len(h), bad[v], h.mean(), h.max(), int(np.percentile(h, 95)), np.bincount(h).tolist(), worst[v]))

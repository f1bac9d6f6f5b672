"""Development aid (NOT product): does a DUAL warm start -- the previous replan's slacks and multipliers, shifted by one segment -- cut
the iteration count of the interior-point iteration, in particular its TAIL (a batch lasts as long as its slowest QP)?  numpy
prototype (tools/proto_pdip.py) on the bench's own workload generator; every variant replans the same swarm states.

usage: python tools/proto_dual_warm.py [N M dim n_obs replans seed]"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import synth  # noqa: E402
from tools import proto_pdip as PP  # noqa: E402

N, M, dim, n_obs, replans, seed = [int(v) for v in (sys.argv[1:] + ["64", "5", "3", "20", "6", "1000"][len(sys.argv) - 1:])]
style = os.environ.get("STYLE", "forest")


def rows_of(b, q):
    L = b["lsc"][q]  # (K, M, 6)
    r = np.zeros(L.shape + (4,))
    r[..., :3] = L["nrm"]
    r[..., 3] = L["d"] + (L["nrm"] * L["p"]).sum(-1)
    return r


def run(rule, mu0_s0=None, label="", **kw):
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    prev = [dict() for _ in range(N)]
    hist = []
    for step in range(replans):
        b = sw.build()
        X, its = np.zeros((N, dim * M * 6)), []
        for q in range(N):
            hdr = dict(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q], vmax=[1.0] * 3, amax=[2.0] * 3,
                       radius=0.15, init=b["init"][q] if step > 0 else None)
            d = np.linalg.norm(np.float32(b["goal"][q]) - np.float32(b["p0"][q]))
            ts = min(M, max(int((M * 0.2 - d / 1.0 + 1e-9) / 0.2), 1))
            sfc = np.stack([b["sfc"][q]["bmin"], b["sfc"][q]["bmax"]], axis=1)
            st = {}
            ds = (prev[q], rule) if (rule is not None and step > 0 and prev[q]) else None
            x, obj, status, it = PP.solve(M, dim, 0.2, 0.01, 1.0, 3.0, True, True, sw.world_min, sw.world_max, hdr, rows_of(b, q), sfc, ts,
                                          dual_start=ds, nbr_ids=b["nbr"][q], state_out=st, mu0_s0=mu0_s0 if step > 0 else None, **kw)
            assert status == 0, (label, step, q, status)
            prev[q] = st
            X[q] = x
            its.append(it)
        if step > 0:
            hist.append(its)
        sw.advance(X)
    h = np.array(hist)
    print("%-44s mean %.2f  per-replan max %s  hist %s" % (label, h.mean(), h.max(axis=1).tolist(), np.bincount(h.reshape(-1)).tolist()))
    return h


if __name__ == "__main__":
    which = os.environ.get("VARIANTS", "base,split").split(",")
    if "base" in which:
        run(None, label="primal warm start (the kernel's): (1e-3, 0.03)")
    if "split" in which:
        run(None, label="  + separate primal / dual step lengths", split_steps=True)
    if "tight" in which:
        run(None, mu0_s0=(1e-7, 0.003), label="tight primal (1e-7, 3 mm), no safety net")
    if "dual" in which:
        for mu_t in (1e-5, 1e-6):
            def rule(res, sp, lp, s0, l0, mu_t=mu_t):
                if lp > sp:
                    lam = max(lp, 1e-12)
                    return max(mu_t / lam, 1e-9), lam
                s = max(res, 0.003)
                return s, mu_t / s
            run(rule, label="dual: active set + multipliers, mu_t = %g" % mu_t)
    if "dualA" in which:
        def ruleA(res, sp, lp, s0, l0):
            # only the ACTIVE SET of the previous optimum is used, at the safe centring: previously active rows get a tighter slack floor
            if lp > sp:
                s = max(res, 0.003)
                return s, 1e-3 / s
            return s0, l0
        run(ruleA, label="dual-A: active rows slack floor 3 mm at mu0 = 1e-3")
        def ruleB(res, sp, lp, s0, l0):
            # previously INACTIVE rows that start close to their bound get a larger slack floor (their multiplier has to vanish)
            if lp <= sp and res < 0.03:
                s = 0.1
                return s, 1e-3 / s
            return s0, l0
        run(ruleB, label="dual-B: inactive rows near the bound start at s = 0.1")

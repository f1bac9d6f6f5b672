"""Development aid: does a float32 factorisation (with fp64 residuals) keep the interior-point iteration on track?

Runs the numpy prototype on a synthetic swarm (carried forward by the CPU oracle, test infrastructure) with the reduced
system factorised in fp64, and in emulated float32 with 0 / 1 / 2 refinement sweeps; prints iterations and the deviation of
x / objective from the fp64 run.   python -m tools.proto_fp32 [N] [M] [n_obs] [style]
"""
import sys

import numpy as np

from lsc_dr_planner_amd import synth
from oracle import oracle as O
from tools import proto_pdip as PP


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_obs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    style = sys.argv[4] if len(sys.argv) > 4 else "forest"
    PP.PIVOT_FLOOR = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    PP.JACOBI = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
    dim = 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=5, style=style)
    cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)

    def agents(b):
        ag = np.zeros(N, O.AGENT_DTYPE)
        for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
            ag[f] = b[f]
        ag["vmax"], ag["amax"], ag["radius"], ag["nominal_velocity"], ag["n_obs"] = 1.0, 2.0, 0.15, 1.0, sw.n_obs
        return ag

    for _ in range(3):
        b = sw.build()
        R = O.solve_batch(cls, agents(b), np.ascontiguousarray(b["lsc"]).reshape(-1), np.arange(N) * sw.n_obs * M * 6,
                          np.ascontiguousarray(b["sfc"]).reshape(-1), threads=8)
        assert (R["status"] == 0).all()
        sw.advance(R["x"])
    b = sw.build()
    lsc = b["lsc"]
    rows = np.zeros(lsc.shape + (4,))
    rows[..., :3] = lsc["nrm"]
    rows[..., 3] = lsc["d"] + (lsc["nrm"] * lsc["p"]).sum(-1)
    res = {}
    for mode in (None, 0, 1):
        its, xs, objs, sts = [], [], [], []
        for q in range(N):
            hdr = dict(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q],
                       vmax=[1.0] * 3, amax=[2.0] * 3, radius=0.15, init=b["init"][q])
            g = np.asarray(b["goal"][q], float) - np.asarray(b["p0"][q], float)
            ts = max(int((M * 0.2 - np.linalg.norm(g) / 1.0 + 1e-9) / 0.2), 1)
            sfc = np.stack([b["sfc"][q]["bmin"], b["sfc"][q]["bmax"]], axis=1)
            x, obj, st, it = PP.solve(M, dim, 0.2, 0.01, 1.0, 3.0, True, True, sw.world_min, sw.world_max, hdr, rows[q], sfc, ts,
                                      fp32=mode)
            its.append(it), xs.append(x), objs.append(obj), sts.append(st)
        res[mode] = (np.array(its), np.array(xs), np.array(objs), np.array(sts))
    ref = res[None]
    for mode in (None, 0, 1):
        it, x, obj, st = res[mode]
        ok = (st == 0) & (ref[3] == 0)
        print("factor %-8s  status!=0: %2d   iters mean %.2f max %2d   max|dx| %.2e   max rel dobj %.2e" % (
            "fp64" if mode is None else "fp32+%dr" % mode, int((st != 0).sum()), it.mean(), it.max(),
            np.abs(x - ref[1])[ok].max() if ok.any() else np.nan,
            (np.abs(obj - ref[2]) / np.maximum(1, np.abs(ref[2])))[ok].max() if ok.any() else np.nan))


if __name__ == "__main__":
    main()

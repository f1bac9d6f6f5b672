"""Development aid: the dual ACTIVE SET phase (csrc/lscqp_das.hip) on the GPU -- parity with the interior-point kernel alone and with the
CPU oracle, what share of a batch each path solves, and the time of one launch, on swarms of the bench's shapes.

usage (GPU box): python tools/das_check.py [quick]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from lsc_dr_planner_amd import api, synth
    from oracle import oracle as O

    O.build()
    dev = torch.device("cuda", 0)
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(64, 5, 3, 20, "forest", 1000), (10, 10, 2, 9, "forest", 3020), (128, 6, 3, 20, "maze", 3518), (32, 10, 3, 40, "forest", 3138),
              (512, 5, 3, 20, "forest", 7101), (24, 7, 3, 12, "maze", 6), (16, 5, 2, 12, "forest", 11), (24, 3, 3, 10, "forest", 4)]
    if quick:
        shapes = shapes[:2]
    for (N, M, dim, n_obs, style, seed) in shapes:
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        sols = {k: api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=v))
                for k, v in (("on", api.ACTIVE_SET_DEFAULT), ("off", api.ACTIVE_SET_OFF), ("only", api.ACTIVE_SET_ONLY))}
        cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
        for step in range(4):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            x0 = api.x_init_from_swarm(b, dim) if step > 0 else None
            G = {k: s.solve_host(hdr, rows, off, sfc, x_init=x0) for k, s in sols.items()}
            ag = np.zeros(N, O.AGENT_DTYPE)
            for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
                ag[f] = b[f]
            ag["vmax"], ag["amax"], ag["radius"], ag["nominal_velocity"], ag["n_obs"] = 1.0, 2.0, 0.15, 1.0, sw.n_obs
            R = O.solve_batch(cls, ag, np.ascontiguousarray(b["lsc"]).reshape(-1), np.arange(N) * sw.n_obs * M * 6, np.ascontiguousarray(b["sfc"]).reshape(-1), threads=8)
            on, offr, only = G["on"], G["off"], G["only"]
            as_solved = (on["info"]["flags"] & api.INFO_ACTIVE_SET) != 0
            only_ok = only["status"] == 0
            okb = (on["status"] == 0) & (R["status"] == 0)
            dx_or = np.abs(on["x"] - R["x"])[okb].max() if okb.any() else 0
            do_or = (np.abs(on["obj"] - R["obj"]) / np.maximum(1, np.abs(R["obj"])))[okb].max() if okb.any() else 0
            both = (on["status"] == 0) & (offr["status"] == 0)
            dx_off = np.abs(on["x"] - offr["x"])[both].max()
            st = on["info"]["iterations"][as_solved]
            print("%4d x M%d dim%d obs%d %-6s step %d: status on %s off %s oracle %s | active-set solved %d/%d (only-mode %d) steps mean %.2f max %d | "
                  "pdip iters (off) mean %.2f max %d | max|dx| vs oracle %.1e  rel dobj %.1e  vs pdip-only %.1e | res_p %.1e res_d %.1e" % (
                      N, M, dim, sw.n_obs, style, step, np.bincount(on["status"], minlength=1).tolist(), np.bincount(offr["status"], minlength=1).tolist(),
                      np.bincount(R["status"], minlength=1).tolist(), as_solved.sum(), N, only_ok.sum(), st.mean() if len(st) else 0, st.max() if len(st) else 0,
                      offr["info"]["iterations"].mean(), offr["info"]["iterations"].max(), dx_or, do_or, dx_off,
                      on["info"]["res_primal"].max(), on["info"]["res_dual"].max()), flush=True)
            xadv = on["x"].copy()
            bad = on["status"] != 0
            if x0 is not None:
                xadv[bad] = x0[bad]
            sw.advance(xadv)
        # timing of the device entry on the last batch (warm start), default stream
        t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, sols["on"].rows_in_format(rows), off, sfc)]
        d_xi = torch.from_numpy(np.ascontiguousarray(x0)).to(dev)
        nv = sols["on"].nv
        d_x = torch.zeros(N * nv, dtype=torch.float64, device=dev)
        d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
        d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
        d_info = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
        for k in ("off", "on", "only"):
            s = sols[k]
            for _ in range(20):
                s.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
            torch.cuda.synchronize()
            a = time.perf_counter()
            reps = 200
            for _ in range(reps):
                s.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - a) / reps * 1e3
            print("      %-5s %.4f ms per call (%d QPs: %.3e QP/s)" % (k, ms, N, N / ms * 1e3), flush=True)
        for s in sols.values():
            s.close()


if __name__ == "__main__":
    main()

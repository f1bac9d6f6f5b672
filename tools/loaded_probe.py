"""How LOADED is a swarm's batch as the mission goes on?  (development aid, round 6; VERDICT r05 "missing" 2, "next" 2)

Every BASELINE batch of bench.py is taken 3 replans after hover: most of its QPs hold no row at the optimum.  The reference's own missions
are not like that (forest10: 8.6 steps mean, 24 max; later replans 48-56).  This probe carries a swarm of a BASELINE shape through its
exchange and prints, per checkpoint (replans after hover): active-set steps histogram, instances the phase proves infeasible / hands over
and why, which kernel finished how many, and the time of one solve call with the phase off / on -- what bench.py's `*_loaded` blocks and
the launch policy (csrc/lscqp_api.hip) are set from.

    python tools/loaded_probe.py c1 3,10,25,40          # config key of bench.py, checkpoints

(The commit before this file's second version carried a record of each solve's active rows from replan to replan as a PRICING HINT for the
next -- lscqp_solve_batch_device_hinted.  Measured with this probe and dropped: the previous replan's active rows, shifted by a segment,
are 8 - 42 % of the next replan's (unshifted: 33 - 50 %), and trying them first RAISED the step counts: 64 x M5 41 -> 75 steps per batch,
512 x M6 761 -> 1051, forest10 150 -> 144.  NOTES.md section 14.)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench  # noqa: E402


def main():
    import torch

    from lsc_dr_planner_amd import api, synth

    key = sys.argv[1] if len(sys.argv) > 1 else "c1"
    marks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3,10,25,40").split(",")]
    over = dict(kv.split("=") for kv in sys.argv[3:])
    cfg = dict(bench.CONFIGS[key])
    for k, v in over.items():
        cfg[k] = type(cfg[k])(v) if k in cfg else v
    N, M, dim, n_obs = cfg["agents"], cfg["segments"], cfg["dim"], cfg["obs"]
    dev = torch.device("cuda", 0)
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=cfg["seed"], style=cfg["style"])
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    only = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    nv = sol.nv
    done, t_build = 0, 0.0
    for mark in marks:
        while done < mark:
            a = time.perf_counter()
            b = sw.build()
            t_build += time.perf_counter() - a
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            x0 = api.x_init_from_swarm(b, dim)
            r = sol.solve_host(hdr, rows, off, sfc, want_info=False, x_init=x0)
            bad = r["status"] != 0
            r["x"][bad] = x0[bad]
            sw.advance(r["x"])
            done += 1
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        dh, do, ds = (bench.to_dev(torch, a, dev) for a in (hdr, off, sfc))
        dr = bench.to_dev(torch, sol.rows_in_format(rows), dev)
        dxi = torch.from_numpy(api.x_init_from_swarm(b, dim)).to(dev)
        dx = torch.zeros(N * nv, dtype=torch.float64, device=dev)
        dob = torch.zeros(N, dtype=torch.float64, device=dev)
        dst = torch.zeros(N, dtype=torch.int32, device=dev)
        dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
        res = {}
        for mode in ("off", "on"):
            sol.set_knob("active_set_off", 1 if mode == "off" else 0)
            call = sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (e0.elapsed_time(e1) / 20, dinfo.cpu().numpy().view(api.INFO_DTYPE).copy(), dst.cpu().numpy().copy(), dx.cpu().numpy().copy())
        sol.set_knob("active_set_off", 0)
        only.solve_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)
        torch.cuda.synchronize()
        io, so = dinfo.cpu().numpy().view(api.INFO_DTYPE), dst.cpu().numpy()
        left = np.where((so != 0) & (so != api.STATUS_INFEASIBLE))[0]
        inf = np.where(so == api.STATUS_INFEASIBLE)[0]
        if len(left) or len(inf):
            print("    phase alone: proves %d infeasible (steps %s); leaves %d: (instance, why, steps) %s" % (
                len(inf), io["iterations"][inf].tolist()[:8], len(left), [(int(q), int(io["res_dual"][q]), int(io["gap"][q])) for q in left[:12]]))
        ms_off, info_off, st_off, x_off = res["off"]
        ms, info, st, x = res["on"]
        by_as = ((info["flags"] & api.INFO_ACTIVE_SET) != 0) & (st == 0)
        it = info["iterations"]
        ok = (st == 0) & (st_off == 0)
        hist = np.bincount(np.minimum(it[by_as], 40), minlength=1)
        print("%s after %3d replans: off %.4f ms | on %.4f ms (x%.2f): phase finished %d / %d, interior point %d, non-optimal %d (off %d), steps mean %.2f max %d, "
              "max|x_on - x_off| %.1e | hist %s | dist to goal %.2f m"
              % (key, mark, ms_off, ms, ms_off / ms, by_as.sum(), N, ((st == 0) & ~by_as).sum(), (st != 0).sum(), (st_off != 0).sum(),
                 it[by_as].mean() if by_as.any() else 0, it[by_as].max() if by_as.any() else 0,
                 np.abs(x.reshape(N, nv) - x_off.reshape(N, nv))[ok].max() if ok.any() else -1, hist.tolist()[:30],
                 np.linalg.norm(sw.final_goal - sw.pos, axis=1).mean()), flush=True)
    print("synth build: %.1f s for %d replans" % (t_build, done))


if __name__ == "__main__":
    main()

"""How LOADED is a swarm's batch as the mission goes on?  (development aid, round 6; VERDICT r05 "missing" 2)

Every BASELINE batch of bench.py is taken 3 replans after hover: most of its QPs hold no row at the optimum.  The reference's own missions
are not like that (forest10: 8.6 steps mean, 24 max; later replans 48-56).  This probe carries a swarm of a BASELINE shape through its
exchange and prints, per checkpoint (replans after hover): active-set steps histogram, instances the phase hands over, which kernel finished
how many, and the time of one solve call with the phase on and off -- what bench.py's `*_loaded` blocks and the large-batch launch policy
(csrc/lscqp_api.hip) are set from.

    python tools/loaded_probe.py c1 3,10,25,40          # config key of bench.py, checkpoints
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
    done = 0
    t_build = 0.0
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
        dx = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
        dob = torch.zeros(N, dtype=torch.float64, device=dev)
        dst = torch.zeros(N, dtype=torch.int32, device=dev)
        dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
        res = {}
        for mode in ("on", "off"):
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
            info = dinfo.cpu().numpy().view(api.INFO_DTYPE)
            st = dst.cpu().numpy()
            res[mode] = (e0.elapsed_time(e1) / 20, info.copy(), st.copy(), dx.cpu().numpy().copy())
        sol.set_knob("active_set_off", 0)
        ms_on, info, st, x_on = res["on"]
        ms_off, info_off, st_off, x_off = res["off"]
        by_as = (info["flags"] & api.INFO_ACTIVE_SET) != 0
        it = info["iterations"]
        hist = np.bincount(np.minimum(it[by_as], 40), minlength=1)
        far = np.linalg.norm(sw.final_goal - sw.pos, axis=1)
        print("%s after %3d replans: phase on %.4f ms, off %.4f ms (x%.2f) | finished by phase %d / %d, handed over %d, non-optimal %d (off: %d) | "
              "steps mean %.2f max %d, hist[0..] %s | IP iters of handed-over %s | max|x_on - x_off| %.2e | mean dist to goal %.2f m"
              % (key, mark, ms_on, ms_off, ms_off / ms_on, by_as.sum(), N, (~by_as).sum(), (st != 0).sum(), (st_off != 0).sum(),
                 it[by_as].mean() if by_as.any() else 0, it[by_as].max() if by_as.any() else 0, hist.tolist()[:26],
                 it[~by_as].tolist()[:8], np.abs(x_on - x_off).max(), far.mean()), flush=True)
    print("synth build: %.1f s for %d replans" % (t_build, done))


if __name__ == "__main__":
    main()

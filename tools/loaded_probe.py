"""How LOADED is a swarm's batch as the mission goes on?  (development aid, round 6; VERDICT r05 "missing" 2, "next" 2 and 6)

Every BASELINE batch of bench.py is taken 3 replans after hover: most of its QPs hold no row at the optimum.  The reference's own missions
are not like that (forest10: 8.6 steps mean, 24 max; later replans 48-56).  This probe carries a swarm of a BASELINE shape through its
exchange ON THE DEVICE (lscqp_solve_batch_device_hinted: every replan leaves the record of its active rows for the next) and prints, per
checkpoint (replans after hover): active-set steps histogram, instances the phase hands over and why, which kernel finished how many, and
the time of one solve call with the phase off / on / on with the previous replan's record as the hint -- what bench.py's `*_loaded` blocks
and the launch policy (csrc/lscqp_api.hip) are set from.

    python tools/loaded_probe.py c1 3,10,25,40 [neighbour_order=id]         # config key of bench.py, checkpoints
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
    order = over.pop("neighbour_order", "id")
    cfg = dict(bench.CONFIGS[key])
    for k, v in over.items():
        cfg[k] = type(cfg[k])(v) if k in cfg else v
    N, M, dim, n_obs = cfg["agents"], cfg["segments"], cfg["dim"], cfg["obs"]
    dev = torch.device("cuda", 0)
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=cfg["seed"], style=cfg["style"], neighbour_order=order)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    only = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    nv = sol.nv
    dx = torch.zeros(N * nv, dtype=torch.float64, device=dev)
    dob = torch.zeros(N, dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    act = torch.full((N, api.ACTIVE_SLOTS), -1, dtype=torch.int32, device=dev)  # the record carried from replan to replan
    done, t_build, steps_carried = 0, 0.0, []

    def upload(b):
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        dh, do, ds = (bench.to_dev(torch, a, dev) for a in (hdr, off, sfc))
        dr = bench.to_dev(torch, sol.rows_in_format(rows), dev)
        dxi = torch.from_numpy(api.x_init_from_swarm(b, dim)).to(dev)
        return dh, dr, do, ds, dxi

    for mark in marks:
        while done < mark:
            a = time.perf_counter()
            b = sw.build()
            t_build += time.perf_counter() - a
            dh, dr, do, ds, dxi = upload(b)
            sol.solve_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi, retry=1, d_active=act,
                             hint_mode=api.HINT_SHIFTED if done > 0 else api.HINT_NONE)
            torch.cuda.synchronize()
            x, st = dx.cpu().numpy().reshape(N, nv).copy(), dst.cpu().numpy()
            x0 = api.x_init_from_swarm(b, dim)
            x[st != 0] = x0[st != 0]
            steps_carried.append(int(dinfo.cpu().numpy().view(api.INFO_DTYPE)["iterations"].max()))
            sw.advance(x)
            done += 1
        b = sw.build()
        dh, dr, do, ds, dxi = upload(b)
        prev = act.clone()  # the record the previous replan left: the realistic hint of THIS replan
        res = {}
        for mode in ("off", "on", "hinted"):
            sol.set_knob("active_set_off", 1 if mode == "off" else 0)
            reps = 20
            if mode == "hinted":  # every timed call reads a fresh copy of the previous replan's record (the call overwrites it with its own)
                recs = [prev.clone() for _ in range(reps + 3)]
                calls = [sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi, d_active=r, hint_mode=api.HINT_SHIFTED) for r in recs]
            else:
                calls = [sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)] * (reps + 3)
            for c in calls[:3]:
                c()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in calls[3:]:
                c()
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (e0.elapsed_time(e1) / reps, dinfo.cpu().numpy().view(api.INFO_DTYPE).copy(), dst.cpu().numpy().copy(), dx.cpu().numpy().copy())
        sol.set_knob("active_set_off", 0)
        for tag, rec, hm in (("cold", None, api.HINT_NONE), ("hinted", prev.clone(), api.HINT_SHIFTED)):
            only.solve_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi, d_active=rec, hint_mode=hm)
            torch.cuda.synchronize()
            io, so = dinfo.cpu().numpy().view(api.INFO_DTYPE), dst.cpu().numpy()
            left = np.where((so != 0) & (so != api.STATUS_INFEASIBLE))[0]
            inf = np.where(so == api.STATUS_INFEASIBLE)[0]
            if len(left) or len(inf):
                print("    phase alone (%s): proves %d infeasible (steps %s); leaves %d: (instance, why, steps) %s" % (
                    tag, len(inf), io["iterations"][inf].tolist()[:8], len(left), [(int(q), int(io["res_dual"][q]), int(io["gap"][q])) for q in left[:12]]))
        ms_off, info_off, st_off, x_off = res["off"]
        line = "%s after %3d replans: off %.4f ms" % (key, mark, ms_off)
        for mode in ("on", "hinted"):
            ms, info, st, x = res[mode]
            by_as = ((info["flags"] & api.INFO_ACTIVE_SET) != 0) & (st == 0)
            it = info["iterations"]
            ok = (st == 0) & (st_off == 0)
            line += (" | %s %.4f ms (x%.2f): phase finished %d / %d, IP %d, non-optimal %d (off %d), steps mean %.2f max %d sum %d, dx vs off %.1e"
                     % (mode, ms, ms_off / ms, by_as.sum(), N, ((st == 0) & ~by_as).sum(), (st != 0).sum(), (st_off != 0).sum(),
                        it[by_as].mean() if by_as.any() else 0, it[by_as].max() if by_as.any() else 0, it[by_as].sum(),
                        np.abs(x.reshape(N, nv) - x_off.reshape(N, nv))[ok].max() if ok.any() else -1))
        # how much of this replan's active set did the previous replan's (shifted by a segment) predict?
        cur = torch.full((N, api.ACTIVE_SLOTS), -1, dtype=torch.int32, device=dev)
        sol.solve_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi, d_active=cur, hint_mode=api.HINT_NONE)
        torch.cuda.synchronize()
        P_, NX_ = 6 * M, dim * 6 * M
        oV, oA = NX_, NX_ + dim * 5 * M
        oC = oA + dim * 4 * M

        def shifted(code):
            if code < 0:
                return -1
            v = code & 0x3fffffff
            if not (code & 0x40000000):
                o, cp = divmod(v, P_)
                return o * P_ + cp - 6 if cp - 6 >= 3 else -1
            r, side = v >> 1, v & 1
            if r < oV:
                k_, cp = divmod(r, P_)
                rn = k_ * P_ + cp - 6 if cp - 6 >= 3 else -1
            elif r < oA:
                k_, rr = divmod(r - oV, 5 * M)
                rn = oV + k_ * 5 * M + rr - 5 if rr - 5 >= 0 else -1
            elif r < oC:
                k_, rr = divmod(r - oA, 4 * M)
                rn = oA + k_ * 4 * M + rr - 4 if rr - 4 >= 0 else -1
            else:
                rn = -1
            return (0x40000000 | (2 * rn + side)) if rn >= 0 else -1

        pv, cv = prev.cpu().numpy(), cur.cpu().numpy()
        n_new = n_hit = n_prev = n_same = 0
        for q in range(N):
            new = set(int(v) for v in cv[q] if v >= 0)
            old = set(shifted(int(v)) for v in pv[q] if v >= 0) - {-1}
            old_unshifted = set(int(v) for v in pv[q] if v >= 0)
            n_new += len(new); n_prev += len(old); n_hit += len(new & old); n_same += len(new & old_unshifted)
        print("    active rows now %d; previous replan's record shifted: %d rows, %d of them active now (%.0f %% of now's); unshifted ids in common: %d"
              % (n_new, n_prev, n_hit, 100.0 * n_hit / max(n_new, 1), n_same))
        ms, info, st, x = res["on"]
        by_as = ((info["flags"] & api.INFO_ACTIVE_SET) != 0) & (st == 0)
        hist = np.bincount(np.minimum(info["iterations"][by_as], 40), minlength=1)
        print(line + " | hist(on) %s | dist to goal %.2f m" % (hist.tolist()[:30], np.linalg.norm(sw.final_goal - sw.pos, axis=1).mean()), flush=True)
    print("synth build: %.1f s for %d replans; max steps of the carried (hinted) replans: %s" % (t_build, done, steps_carried))


if __name__ == "__main__":
    main()

"""Development aid: races show as launch-to-launch differences.  The same device-resident batch is solved `reps` times by the phase alone and by
the whole chain; every launch's statuses, step counts and control points are compared bit for bit with the first launch's.
usage (GPU box): python tools/das_soak.py [reps] [config ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from lsc_dr_planner_amd import api, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
for key in sys.argv[2:] or ["c0", "c3s", "c1", "c2"]:
    cfg = bench.CONFIGS[key]
    N, M, dim = cfg["agents"], cfg["segments"], cfg["dim"]
    for mode, aset in (("only", api.ACTIVE_SET_ONLY), ("on", api.ACTIVE_SET_DEFAULT)):
        sw, sol, build, (hdr, rows, off, sfc) = bench.make_batch(api, synth, lambda s: api.Solver(api.make_desc(M=M, dim=dim, world_min=s.world_min, world_max=s.world_max, active_set=aset)),
                                                                 N, M, dim, cfg["obs"], seed=cfg["seed"], style=cfg["style"], warm_steps=3)
        t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
        d_xi = torch.from_numpy(np.ascontiguousarray(api.x_init_from_swarm(build, dim))).to(dev)
        first = None
        bad = 0
        for r in range(reps):
            d_x = torch.full((N * sol.nv,), float("nan"), dtype=torch.float64, device=dev)
            d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
            d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
            d_info = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
            sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
            torch.cuda.synchronize()
            cur = (d_st.cpu().numpy().tobytes(), d_x.cpu().numpy().tobytes(), d_info.cpu().numpy().view(api.INFO_DTYPE)["iterations"].tobytes())
            if first is None:
                first = cur
                st0 = np.bincount(np.frombuffer(cur[0], np.int32) + 1, minlength=4).tolist()
            elif cur != first:
                bad += 1
                if bad <= 3:
                    s0, s1 = np.frombuffer(first[0], np.int32), np.frombuffer(cur[0], np.int32)
                    i0, i1 = np.frombuffer(first[2], np.int32), np.frombuffer(cur[2], np.int32)
                    x0, x1 = np.frombuffer(first[1]).reshape(N, -1), np.frombuffer(cur[1]).reshape(N, -1)
                    dq = np.nonzero((s0 != s1) | (i0 != i1) | (x0 != x1).any(axis=1))[0]
                    print("   launch %d differs at instances %s: status %s -> %s, steps %s -> %s, max|dx| %.2e" % (
                        r, dq[:6].tolist(), s0[dq][:6].tolist(), s1[dq][:6].tolist(), i0[dq][:6].tolist(), i1[dq][:6].tolist(),
                        np.nanmax(np.abs(x0[dq] - x1[dq])) if len(dq) else 0), flush=True)
        print("%s %-4s: %d launches, statuses(-1,0,1,2..) %s, launches that differ from the first: %d" % (key, mode, reps, st0, bad), flush=True)

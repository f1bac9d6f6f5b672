"""Development aid: one fresh process; the forest10 batch's warm-up replans with the phase alone and with the whole chain; prints statuses and,
for instances the phase handed over, what lscqp_info holds.  Run many times: every process must print the same line."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lsc_dr_planner_amd import api, synth  # noqa: E402

out = []
WANT_INFO = os.environ.get("FLAKY_INFO", "1") == "1"
OWN = os.environ.get("FLAKY_OWN", "0") == "1"  # the solver under test carries the swarm (bench.make_batch), not the reference path
for key in sys.argv[1:] or ["c0"]:
    cfg = bench.CONFIGS[key]
    N, M, dim = cfg["agents"], cfg["segments"], cfg["dim"]
    for mode, aset in (("on", api.ACTIVE_SET_DEFAULT), ("only", api.ACTIVE_SET_ONLY)):
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=cfg["obs"], seed=cfg["seed"], style=cfg["style"])
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=aset))
        ref = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_OFF))
        for step in range(4):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            x0 = api.x_init_from_swarm(b, dim)
            r = sol.solve_host(hdr, rows, off, sfc, x_init=x0, want_info=WANT_INFO)
            g = ref.solve_host(hdr, rows, off, sfc, x_init=x0)
            bad = np.nonzero(r["status"] != 0)[0]
            s = "%s%d:%s:%s" % (mode[:2], step, np.bincount(r["status"], minlength=3).tolist(), hashlib.md5(r["x"][r["status"] == 0].tobytes()).hexdigest()[:4])
            for q in (bad if WANT_INFO else []):
                s += "{q%d it%d fl%d gap%.0f}" % (q, r["info"]["iterations"][q], r["info"]["flags"][q], r["info"]["gap"][q])
            out.append(s)
            xg = g["x"].copy()
            xg[g["status"] != 0] = x0[g["status"] != 0]
            if OWN:
                xg = r["x"].copy()
                xg[r["status"] != 0] = x0[r["status"] != 0]
            sw.advance(xg)  # (the reference path carries the swarm: both modes see the same batches)
print(" ".join(out))

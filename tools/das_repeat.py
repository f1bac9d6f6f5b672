"""Development aid: one fresh process = one line.  The dual active-set phase ALONE (LSCQP_ACTIVE_SET_ONLY) on two small swarms whose launches use the
large-LDS form (table copy + staged rows above 64 KB); prints how many instances it finished, its step count and a hash of x, so that
`for i in $(seq 50); do python tools/das_repeat.py; done | sort | uniq -c` shows whether every process (and every box) agrees."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import api, synth  # noqa: E402

out = []
for (N, M, dim, n_obs, style, seed) in [(10, 10, 2, 9, "forest", 3020), (32, 10, 3, 40, "forest", 3138), (24, 7, 3, 12, "maze", 6)]:
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    ref = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_OFF))
    x0 = None
    for step in range(3):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        r = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
        g = ref.solve_host(hdr, rows, off, sfc, x_init=x0)
        out.append("%d/%d:%d:%s" % ((r["status"] == 0).sum(), N, r["info"]["iterations"].sum(), hashlib.md5(r["x"][r["status"] == 0].tobytes()).hexdigest()[:6]))
        x0 = api.x_init_from_swarm(b, dim)
        sw.advance(g["x"])
print(" ".join(out))

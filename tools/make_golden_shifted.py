"""Makes tests/golden/shifted_pivot.json (run on the GPU box): an instance whose interior-point solve loses a pivot late in the iteration and
repeats that iteration with a diagonal shift (LSCQP_INFO_SHIFTED).  Found by sweeping swarms of the M = 10 classes with the active-set
phase OFF (it would finish these instances before the interior-point kernel sees them); the fixture holds the inputs and what the kernel
reported (iterations, flags) so that the test can assert that a repeated iteration is counted once.

    python tools/make_golden_shifted.py [max_seeds]      (writes gpurun_out/shifted_pivot.json for the first instance found)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsc_dr_planner_amd import api, synth  # noqa: E402

max_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
found = None
for (N, M, dim, n_obs, style) in [(24, 10, 3, 40, "forest"), (16, 10, 2, 9, "forest"), (24, 8, 2, 12, "maze"), (24, 7, 3, 12, "maze")]:
    for seed in range(200, 200 + max_seeds):
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_OFF))
        x0 = None
        for step in range(4):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            r = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
            sh = np.nonzero((r["info"]["flags"] & api.INFO_SHIFTED) != 0)[0]
            for q in sh:
                print("shape", (N, M, dim, n_obs, style), "seed", seed, "replan", step, "agent", int(q), "status", int(r["status"][q]), "iterations", int(r["info"]["iterations"][q]),
                      "flags", int(r["info"]["flags"][q]), flush=True)
                if found is None and r["status"][q] == 0 and x0 is None:  # (a cold instance: the fixture needs no warm start)
                    rr = rows[off[q]:off[q + 1]]
                    found = {"source": "tools/make_golden_shifted.py: synth.Swarm(%d, M=%d, dim=%d, n_obs=%d, seed=%d, style=%r), replan %d, agent %d, active-set phase off, "
                                       "cold start: a pivot of a late iteration's matrix is lost and the iteration is repeated with a diagonal shift" % (N, M, dim, n_obs, seed, style, step, q),
                             "M": M, "dim": dim, "n_obs": int(hdr["n_obs"][q]), "world_min": [float(v) for v in sw.world_min], "world_max": [float(v) for v in sw.world_max],
                             "hdr": {f: (hdr[f][q].tolist() if np.ndim(hdr[f][q]) else float(hdr[f][q])) for f in ("p0", "v0", "a0", "goal", "next_waypoint", "vmax", "amax", "radius", "nominal_velocity")},
                             "rows": np.stack([rr["nx"], rr["ny"], rr["nz"], rr["b"]], axis=1).tolist(),
                             "sfc_min": sfc["bmin"][q].tolist(), "sfc_max": sfc["bmax"][q].tolist(),
                             "iterations": int(r["info"]["iterations"][q]), "flags": int(r["info"]["flags"][q]), "obj": float(r["obj"][q])}
            x0 = api.x_init_from_swarm(b, dim)
            xa = r["x"].copy()
            bad = r["status"] != 0
            xa[bad] = x0[bad]
            sw.advance(xa)
            x0 = None if os.environ.get("SHIFTED_COLD", "1") == "1" else x0
    if found:
        break
if found:
    out = os.path.join(ROOT, "gpurun_out", "shifted_pivot.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(found, open(out, "w"))
    print("written", out, found["source"])
else:
    print("no SHIFTED instance found")

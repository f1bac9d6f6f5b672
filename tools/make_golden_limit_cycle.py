"""Makes tests/golden/limit_cycle_dlsc.json (run on the GPU box): the instance of tools/stress_parity.py --dlsc --seed0 400 (shape
(24, 10, 3, 40, 'forest'), seed 402, replan 1) whose predictor-corrector iteration runs into a limit cycle of period four and ends at
the iteration limit on every kernel, from the default start.  Inputs only: the test solves it with the oracle itself.

    python tools/make_golden_limit_cycle.py        (writes the fixture if the first pass still ends at the iteration limit)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from lsc_dr_planner_amd import api, synth  # noqa: E402

N, M, dim, n_obs, style, seed = 24, 10, 3, 40, "forest", 402
sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=api.PLANNER_DLSC, world_min=sw.world_min, world_max=sw.world_max))
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
for step in range(2):
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    d_info = torch.zeros(N * api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    sol.solve_device(N, sw.n_obs, up(hdr), up(rows), up(off), up(sfc), d_x, d_obj, d_st, d_info=d_info)  # first pass only
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    print("replan", step, "first-pass statuses", np.bincount(st, minlength=5).tolist())
    if step == 0:
        assert (st == 0).all()
        sw.advance(d_x.cpu().numpy().reshape(N, -1))
bad = np.where(st == api.STATUS_ITER_LIMIT)[0]
assert len(bad) == 1, bad
q = int(bad[0])
info = d_info.cpu().numpy().view(api.INFO_DTYPE)[q]
print("instance", q, "iterations", info["iterations"], "gap", info["gap"], "n_obs", hdr["n_obs"][q])
r = rows[off[q]:off[q + 1]]
g = {"source": "tools/make_golden_limit_cycle.py: synth.Swarm(24, M=10, dim=3, n_obs=40, seed=402, style='forest'), planner mode DLSC, replan 1, agent %d: "
               "from the default start the predictor-corrector iteration cycles with period four (gap 6e-6 .. 6e-5) to the iteration limit" % q,
     "M": M, "dim": dim, "n_obs": int(hdr["n_obs"][q]), "planner_mode": "dlsc", "world_min": [float(v) for v in sw.world_min], "world_max": [float(v) for v in sw.world_max],
     "hdr": {f: (hdr[f][q].tolist() if np.ndim(hdr[f][q]) else float(hdr[f][q])) for f in ("p0", "v0", "a0", "goal", "next_waypoint", "vmax", "amax", "radius", "nominal_velocity")},
     "rows": np.stack([r["nx"], r["ny"], r["nz"], r["b"]], axis=1).tolist(),
     "sfc_min": sfc["bmin"][q].tolist(), "sfc_max": sfc["bmax"][q].tolist()}
out = os.path.join(ROOT, "gpurun_out", "limit_cycle_dlsc.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(g, open(out, "w"))
print("wrote", out, "(copy it to tests/golden/)")

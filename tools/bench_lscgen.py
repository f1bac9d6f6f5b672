"""Development aid / measurement: the LSC-generation kernel at BASELINE batch sizes (HBM-side kernel of SURVEY 8f-1).
    python tools/bench_lscgen.py [N ...]
Prints kernel time (HIP events on the launch stream), algorithmic GB/s against the 8 TB/s HBM peak, and the CPU oracle
(OpenMP) on the same inputs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lsc_dr_planner_amd import api, synth  # noqa: E402

Ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 512, 4096]
M, dim, n_obs = 5, 3, 20
dev = torch.device("cuda", 0)
out = []
for N in Ns:
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=1)
    init = sw.initial_traj()
    nbr = sw.neighbours().astype(np.int32)
    goal = np.ascontiguousarray(sw.pos + 0.5, dtype=np.float64)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    d_traj = torch.from_numpy(init.copy()).to(dev)
    d_nbr = torch.from_numpy(nbr).to(dev)
    d_r = torch.full((N,), sw.radius, dtype=torch.float64, device=dev)
    d_dw = torch.full((N,), sw.downwash, dtype=torch.float64, device=dev)
    d_goal = torch.from_numpy(goal).to(dev)
    d_rows = torch.zeros(N * n_obs * M * 6 * 4, dtype=torch.float64, device=dev)
    for _ in range(5):
        sol.generate_lsc_device(N, n_obs, 0, d_traj, d_nbr, d_r, d_dw, d_goal, d_rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        sol.generate_lsc_device(N, n_obs, 0, d_traj, d_nbr, d_r, d_dw, d_goal, d_rows)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = sol.generate_lsc_bytes(N, n_obs, N)
    rec = {"agents": N, "units": N * n_obs * M, "kernel_ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6,
           "hbm_frac": nbytes / (ms * 1e-3) / 8e12}
    if "--cpu" in sys.argv:
        from oracle import oracle as O

        t0 = time.perf_counter()
        O.generate_lsc(init, nbr, sw.radius, sw.downwash, goal, dim=dim)
        rec["cpu_oracle_ms_openmp"] = (time.perf_counter() - t0) * 1e3
    out.append(rec)
    print(json.dumps(rec))

"""Development aid: is there a HISTORY-FREE predictor of a QP's iteration count (for the first replan of a mission, where no previous solve exists)?
Correlates a few quantities of the inputs with the iteration counts of bench configs[3] and the configs[4] shape (NOTES.md section 11)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lsc_dr_planner_amd import api, synth
from scipy.stats import spearmanr
for key in ("c3", "c4_f64"):
    cfg = bench.CONFIGS[key]
    N, M, D, NOBS = cfg["agents"], cfg["segments"], cfg["dim"], cfg["obs"]
    factory = lambda sw: api.Solver(api.make_desc(M=M, dim=D, world_min=sw.world_min, world_max=sw.world_max))
    sw, sol, b, (hdr, rows, off, sfc) = bench.make_batch(api, synth, factory, N, M, D, NOBS, seed=cfg["seed"], style=cfg["style"], warm_steps=3)
    x0 = api.x_init_from_swarm(b, D)
    r = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    it = r["info"]["iterations"]
    P = 6 * M
    R = rows.reshape(N, sw.n_obs, M, 6)
    c = x0.reshape(N, D, M, 6)  # [k][m][i]
    nx, ny, nz, bb = R["nx"], R["ny"], R["nz"], R["b"]
    val = nx * c[:, None, 0] + ny * c[:, None, 1] + (nz * c[:, None, 2] if D == 3 else 0) - bb   # slack of every row at the initial trajectory
    live = (np.abs(nx) + np.abs(ny) + np.abs(nz)) > 0
    live[:, :, 0, :3] = False
    sl = np.where(live, val, 1e9)
    for name, prox in (("rows with slack < 0.02", (sl < 0.02).sum(axis=(1, 2, 3))), ("rows with slack < 0.1", (sl < 0.1).sum(axis=(1, 2, 3))),
                       ("-min slack", -sl.reshape(N, -1).min(axis=1)), ("-sum of 8 smallest slacks", -np.sort(sl.reshape(N, -1), axis=1)[:, :8].sum(axis=1)),
                       ("|goal - p0|", np.linalg.norm(hdr["goal"] - hdr["p0"], axis=1)), ("|v0|", np.linalg.norm(hdr["v0"], axis=1))):
        rho = spearmanr(prox, it).correlation
        slow = set(np.argsort(-it, kind="stable")[: N // 10].tolist())
        first = set(np.argsort(-prox, kind="stable")[: N // 4].tolist())
        print("%s: %-28s spearman %.2f, slowest 10%% inside the proxy's top 25%%: %.0f%%" % (key, name, rho, 100 * len(slow & first) / len(slow)))

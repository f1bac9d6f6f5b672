"""Development aid: randomized parity sweep of the HIP solver against the CPU oracle over seeds / shapes / styles.
    python tools/stress_parity.py [n_seeds]
Prints one line per configuration and every instance that is non-optimal or outside the parity tolerances."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import helpers as H  # noqa: E402
from lsc_dr_planner_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
WARM = "--warm" in sys.argv  # hand the initial trajectory (shifted previous plan) to the solver as primal start
shapes = [(48, 5, 3, 20, "forest"), (32, 6, 3, 20, "maze"), (16, 10, 2, 9, "forest"), (24, 7, 3, 12, "maze"), (32, 4, 3, 12, "forest"),
          (24, 10, 3, 40, "forest"), (40, 5, 2, 12, "forest"), (24, 8, 2, 12, "maze")]
bad_total = 0
for (N, M, dim, n_obs, style) in shapes:
    worst_dx = worst_do = 0.0
    iters = []
    nbad = 0
    for seed in range(100, 100 + n_seeds):
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
        for step in range(3):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            G = sol.solve_host(hdr, rows, off, sfc, x_init=api.x_init_from_swarm(b, dim) if WARM else None)
            ag, lsc, loff, sfco = H.swarm_oracle_inputs(O, sw, b)
            R = O.solve_batch(cls, ag, lsc, loff, sfco, threads=16)
            both = (G["status"] == 0) & (R["status"] == 0)
            dx = np.abs(G["x"] - R["x"]).max(axis=1)
            do = np.abs(G["obj"] - R["obj"]) / np.maximum(1.0, np.abs(R["obj"]))
            for q in range(N):
                if G["status"][q] != R["status"][q] or (both[q] and (dx[q] > 1e-6 or do[q] > 1e-8)):
                    nbad += 1
                    print("  MISMATCH %s seed %d step %d q %d: gpu status %d oracle %d dx %.2e dobj %.2e it %d" % (
                        (N, M, dim, n_obs, style), seed, step, q, G["status"][q], R["status"][q], dx[q], do[q], G["info"]["iterations"][q]))
            if both.any():
                worst_dx = max(worst_dx, dx[both].max())
                worst_do = max(worst_do, do[both].max())
            iters.append(G["info"]["iterations"])
            sw.advance(np.where((G["status"] == 0)[:, None], G["x"], R["x"]))
    it = np.concatenate(iters)
    bad_total += nbad
    print("%-28s seeds %d: mismatches %d, max dx %.2e, max rel dobj %.2e, iterations mean %.2f max %d" % (
        str((N, M, dim, n_obs, style)), n_seeds, nbad, worst_dx, worst_do, it.mean(), it.max()))
print("TOTAL mismatches", bad_total)

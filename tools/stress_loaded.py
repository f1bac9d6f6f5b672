"""Development aid (round 6): randomized parity sweep DEEP into the swarms' exchange -- where instances hold many rows, run out of the phase's
budgets, or have no feasible point -- of the HIP solver (default path: dual active-set phase + interior point behind it) against the CPU oracle
with its exact last step (tests/helpers.py: polish_primal).  tools/stress_parity.py looks at the first three replans after hover.

    python tools/stress_loaded.py [n_seeds] [--replans R] [--seed0 S]

Per shape: instances compared, status disagreements (GPU optimal vs oracle optimal), instances the phase PROVED infeasible and what the oracle
says of them, max |dx| of the phase's / the interior-point kernel's optima against the polished oracle, max relative objective difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from lsc_dr_planner_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.conftest import _TightOracle  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
REPLANS = int(sys.argv[sys.argv.index("--replans") + 1]) if "--replans" in sys.argv else 30
SEED0 = int(sys.argv[sys.argv.index("--seed0") + 1]) if "--seed0" in sys.argv else 700
shapes = [(48, 5, 3, 20, "forest"), (32, 6, 3, 20, "maze"), (16, 10, 2, 9, "forest"), (24, 7, 3, 12, "maze"), (24, 10, 3, 40, "forest"), (24, 8, 2, 12, "maze")]
T = _TightOracle(O)
tot = dict(n=0, dis=0, proven=0, proven_wrong=0)
for (N, M, dim, n_obs, style) in shapes:
    n = dis = proven = proven_wrong = n_ip = 0
    dx_as = dx_ip = dobj = 0.0
    steps_max = 0
    for seed in range(SEED0, SEED0 + n_seeds):
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
        for step in range(REPLANS):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            x0 = api.x_init_from_swarm(b, dim)
            G = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
            if step >= 3:  # (the first replans are tools/stress_parity.py's)
                ag, lsc, loff, sfco = H.swarm_oracle_inputs(O, sw, b)
                R = T.solve_batch(cls, ag, lsc, loff, sfco, threads=16)
                g_ok, o_ok = G["status"] == 0, R["status"] == 0
                by_as = (G["info"]["flags"] & api.INFO_ACTIVE_SET) != 0
                n += N
                bad = np.nonzero(g_ok != o_ok)[0]
                dis += len(bad)
                for q in bad:
                    print("  STATUS (%s) seed %d replan %d q %d: gpu %d (flags %d, it %d) oracle %d" % ((N, M, dim, n_obs, style), seed, step, q, G["status"][q],
                                                                                                   G["info"]["flags"][q], G["info"]["iterations"][q], R["status"][q]))
                pr = by_as & (G["status"] == api.STATUS_INFEASIBLE)
                proven += int(pr.sum())
                proven_wrong += int((pr & o_ok).sum())
                both = g_ok & o_ok
                d = np.abs(G["x"] - R["x"]).max(axis=1)
                if (both & by_as).any():
                    dx_as = max(dx_as, d[both & by_as].max())
                    steps_max = max(steps_max, int(G["info"]["iterations"][both & by_as].max()))
                if (both & ~by_as).any():
                    dx_ip = max(dx_ip, d[both & ~by_as].max())
                    n_ip += int((both & ~by_as).sum())
                if both.any():
                    dobj = max(dobj, (np.abs(G["obj"] - R["obj"]) / np.maximum(1.0, np.abs(R["obj"])))[both].max())
                for q in np.nonzero(both & by_as & (d > 1e-8))[0]:
                    print("  DX (%s) seed %d replan %d q %d: %.2e (steps %d)" % ((N, M, dim, n_obs, style), seed, step, q, d[q], G["info"]["iterations"][q]))
            x = G["x"].copy()
            x[G["status"] != 0] = x0[G["status"] != 0]  # what the planner does with a failed QP: it keeps the initial trajectory
            sw.advance(x)
    print("%-28s seeds %d x replans 3..%d: %6d QPs, status disagreements %d, proven infeasible by the phase %d (oracle optimal: %d), finished by the interior-point kernel %d, "
          "max dx phase %.2e / interior point %.2e, max rel dobj %.2e, steps max %d" % ((N, M, dim, n_obs, style), n_seeds, REPLANS - 1, n, dis, proven, proven_wrong, n_ip, dx_as, dx_ip, dobj, steps_max), flush=True)
    for k_, v_ in (("n", n), ("dis", dis), ("proven", proven), ("proven_wrong", proven_wrong)):
        tot[k_] += v_
print("TOTAL %(n)d QPs: status disagreements %(dis)d; proven infeasible by the phase %(proven)d, of which the oracle calls optimal %(proven_wrong)d" % tot)

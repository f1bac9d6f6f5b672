"""Development aid: randomized parity sweep of the HIP solver against the CPU oracle over seeds / shapes / styles.
    python tools/stress_parity.py [n_seeds] [--seed0 S] [--warm] [--tight] [--dlsc] [--nd] [--gen 0|1|2] [--shape i]
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
SEED0 = int(sys.argv[sys.argv.index("--seed0") + 1]) if "--seed0" in sys.argv else 100  # first seed of the sweep
GEN = None  # --gen 0|1|2: the rows come from the device generator (generateLSC / generateCLSC / generateBVC), not from synth
if "--gen" in sys.argv:
    GEN = int(sys.argv[sys.argv.index("--gen") + 1])
WARM = "--warm" in sys.argv  # hand the initial trajectory (shifted previous plan) to the solver as primal start
TIGHT = "--tight" in sys.argv  # lscqp_class_desc.warm_start = LSCQP_WARM_TIGHT
DLSC = "--dlsc" in sys.argv  # planner mode DLSC (no end stop) instead of LSC
shapes = [(48, 5, 3, 20, "forest"), (32, 6, 3, 20, "maze"), (16, 10, 2, 9, "forest"), (24, 7, 3, 12, "maze"), (32, 4, 3, 12, "forest"),
          (24, 10, 3, 40, "forest"), (40, 5, 2, 12, "forest"), (24, 8, 2, 12, "maze")]
if "--nd" in sys.argv:  # the shapes of round 3's nested-dissection instances (M = 8, 9, 10 in 3-D; with --dlsc the end-stop-free classes)
    shapes = [(24, 10, 3, 14, "forest"), (24, 10, 3, 36, "forest"), (24, 9, 3, 14, "forest"), (24, 9, 3, 40, "maze"), (24, 8, 3, 14, "maze"), (24, 8, 3, 32, "forest")]
if "--shape" in sys.argv:
    shapes = [shapes[int(sys.argv[sys.argv.index("--shape") + 1])]]
bad_total = tol_total = 0
n_kkt_ok = 0
for (N, M, dim, n_obs, style) in shapes:
    worst_dx = worst_do = 0.0
    iters = []
    nbad = n_both_bad = n_oracle_tol = 0
    for seed in range(SEED0, SEED0 + n_seeds):
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        cls = O.make_class(M=M, dim=dim, use_sfc=True, planner_lsc=not DLSC, world_min=sw.world_min, world_max=sw.world_max)
        sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=api.PLANNER_DLSC if DLSC else api.PLANNER_LSC, world_min=sw.world_min, world_max=sw.world_max,
                                       warm_start=api.WARM_TIGHT if TIGHT else api.WARM_DEFAULT))
        for step in range(3):
            b = sw.build()
            hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
            ag, lsc, loff, sfco = H.swarm_oracle_inputs(O, sw, b)
            if GEN is not None:
                dev = torch.device("cuda", 0)
                up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
                d_rows = torch.zeros(N * sw.n_obs * M * 6 * 4, dtype=torch.float64, device=dev)
                sol.generate_constraints_device(GEN, N, sw.n_obs, 0, up(b["init"]), up(b["nbr"].astype(np.int32)), up(np.full(N, sw.radius)),
                                                up(np.full(N, sw.downwash)), up(np.ascontiguousarray(b["goal"], dtype=np.float64)), d_rows)
                torch.cuda.synchronize()
                rows = d_rows.cpu().numpy().view(api.ROW_DTYPE).copy()
                lsc = np.zeros(rows.shape[0], O.LSC_DTYPE)  # packed row n.c >= b  ==  LSC with p_obs = 0, d = b
                lsc["nrm"][:, 0], lsc["nrm"][:, 1], lsc["nrm"][:, 2], lsc["d"] = rows["nx"], rows["ny"], rows["nz"], rows["b"]
            G = sol.solve_host(hdr, rows, off, sfc, x_init=api.x_init_from_swarm(b, dim) if WARM else None)
            R = O.solve_batch(cls, ag, lsc, loff, sfco, threads=16)
            both = (G["status"] == 0) & (R["status"] == 0)
            n_both_bad += int(((G["status"] != 0) & (R["status"] != 0)).sum())
            dx = np.abs(G["x"] - R["x"]).max(axis=1)
            do = np.abs(G["obj"] - R["obj"]) / np.maximum(1.0, np.abs(R["obj"]))
            far = both & ((dx > 1e-6) | (do > 1e-8))
            dx_tight = None
            if far.any():  # whose x is it?  the oracle once more, converged as far as fp64 goes (its default gap target is 1e-11)
                R2 = O.solve_batch(cls, ag, lsc, loff, sfco, tol=1e-14, max_iter=400, threads=16)
                dx_tight = np.where(R2["status"] == 0, np.abs(G["x"] - R2["x"]).max(axis=1), np.nan)
            for q in range(N):
                # a disagreement is one side optimal and the other not (the two solvers name their failures differently:
                # INFEASIBLE here, a stalled / numeric exit in the oracle), or an optimum outside the parity tolerances
                if (G["status"][q] == 0) != (R["status"][q] == 0) or (both[q] and (dx[q] > 1e-6 or do[q] > 1e-8)):
                    nbad += 1
                    print("  MISMATCH %s seed %d step %d q %d: gpu status %d oracle %d dx %.2e dobj %.2e it %d (res_p %.1e res_d %.1e gap %.1e)%s" % (
                        (N, M, dim, n_obs, style), seed, step, q, G["status"][q], R["status"][q], dx[q], do[q], G["info"]["iterations"][q],
                        G["info"]["res_primal"][q], G["info"]["res_dual"][q], G["info"]["gap"][q],
                        (" | against the oracle at tol 1e-14: dx %.2e" % dx_tight[q]) if (dx_tight is not None and both[q]) else ""))
                    if dx_tight is not None and both[q] and dx_tight[q] <= 1e-6:
                        n_oracle_tol += 1
                    if G["status"][q] == 0 and R["status"][q] != 0:
                        # the GPU returned a plan where the oracle gave up: is it a KKT point of the reference's row-for-row model?
                        lq = np.ascontiguousarray(lsc.reshape(N, -1)[q]); sq = np.ascontiguousarray(sfco.reshape(N, -1)[q])
                        try:
                            stat, eqv, iqv = H.kkt_from_primal(O, cls, ag[q:q + 1], lq, sq, G["x"][q])
                            n_kkt_ok += int(max(stat, eqv, iqv) <= 1e-8)
                            print("      ... the GPU point on the reference's model: stationarity %.1e, equality violation %.1e, inequality violation %.1e (flags %d)" % (
                                stat, eqv, iqv, G["info"]["flags"][q]))
                        except Exception as ex:  # noqa: BLE001
                            print("      ... KKT check failed:", type(ex).__name__, str(ex)[:120])
            if both.any():
                worst_dx = max(worst_dx, dx[both].max())
                worst_do = max(worst_do, do[both].max())
            iters.append(G["info"]["iterations"])
            sw.advance(np.where((G["status"] == 0)[:, None], G["x"], R["x"]))
    it = np.concatenate(iters)
    bad_total += nbad
    tol_total += n_oracle_tol
    print("%-28s seeds %d: mismatches %d (%d within 1e-6 m of the oracle at tol 1e-14), max dx %.2e, max rel dobj %.2e, iterations mean %.2f max %d, non-optimal on both sides %d" % (
        str((N, M, dim, n_obs, style)), n_seeds, nbad, n_oracle_tol, worst_dx, worst_do, it.mean(), it.max(), n_both_bad))
print("GPU-optimal / oracle-failed instances that are KKT points of the reference's model to 1e-8:", n_kkt_ok)
print("TOTAL mismatches", bad_total, "of which within 1e-6 m of the oracle converged to 1e-14 (the default-tolerance oracle's x was the loose one):", tol_total)

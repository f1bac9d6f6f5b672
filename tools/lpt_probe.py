"""Development aid: how much of a throughput launch is its TAIL, and what the work queue and the work order buy.

The hardware starts the workgroups of a grid in blockIdx order; a QP that needs 13 iterations and starts late keeps the launch alive on its own.
For a bench config's batch this probe times: the launch as given; with the order of lscqp_order_by_work_device (hint = the iteration counts
of the previous solve of the same batch: perfect); with a hint perturbed by +-1 iteration on half of the instances; shortest first (the worst
order); a random order.  LSCQP_NO_QUEUE=1 in the environment disables the persistent workgroups (one instance per workgroup, grid = n), so
the two mechanisms can be told apart.  Results are compared bit for bit with the as-given launch every time."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from lsc_dr_planner_amd import api, synth  # noqa: E402

for key in [a for a in sys.argv[1:] if not a.startswith("-")] or ([] if "--chain" in sys.argv else ["c3", "c4_f64", "c2"]):
    cfg = bench.CONFIGS[key]
    N, M, D, NOBS = cfg["agents"], cfg["segments"], cfg["dim"], cfg["obs"]

    def factory(sw):
        return api.Solver(api.make_desc(M=M, dim=D, world_min=sw.world_min, world_max=sw.world_max))

    sw, sol, b, (hdr, rows, off, sfc) = bench.make_batch(api, synth, factory, N, M, D, NOBS, seed=cfg["seed"], style=cfg["style"], warm_steps=3)
    dev = torch.device("cuda", 0)
    t = [bench.to_dev(torch, a, dev) for a in (hdr, rows, off, sfc)]
    dxi = torch.from_numpy(api.x_init_from_swarm(b, D)).to(dev)
    dx = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    dob = torch.zeros(N, dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)

    def run(d_order, reps=20):
        call = lambda: sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], dx, dob, dst, dinfo, d_x_init=dxi, d_order=d_order)  # noqa: E731
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, dx.cpu().numpy().copy(), dinfo.cpu().numpy().view(api.INFO_DTYPE)["iterations"].copy()

    ms0, x0, it = run(None)
    d_order = torch.zeros(N, dtype=torch.int32, device=dev)
    sol.order_by_work_device(N, dinfo, d_order)
    torch.cuda.synchronize()
    lpt = d_order.cpu().numpy()
    assert np.array_equal(lpt, np.argsort(-np.minimum(it, 63), kind="stable").astype(np.int32)), "order_by_work is not the stable descending sort"
    res = {"as given": ms0}
    rng = np.random.default_rng(0)
    noisy = it + rng.integers(-1, 2, N) * (rng.random(N) < 0.5)
    for name, order in (("longest first (lscqp_order_by_work_device)", lpt), ("noisy hint", np.argsort(-noisy, kind="stable")),
                        ("shortest first", lpt[::-1].copy()), ("random", rng.permutation(N))):
        ms, x, _ = run(torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(dev))
        assert np.array_equal(x, x0), name
        res[name] = ms
    print("%s (%d QPs, iterations mean %.2f max %d, queue %s): " % (key, N, it.mean(), it.max(), "off" if os.environ.get("LSCQP_NO_QUEUE") else "on") +
          " | ".join("%s %.4f ms" % kv for kv in res.items()))


def chain_hint_quality(N=256, replans=40):
    """How good is the hint in a real closed loop?  bench.py's 3-D replan chain (CLSC rows, corridors, goal LP, QP; agents on a sphere swapping
    sides) with N agents: per replan the iteration count of every agent's QP; reported: the rank correlation between consecutive replans and
    how many of the slowest 10 % of a replan were among the slowest 25 % of the previous one (what the first round of a sorted launch holds)."""
    rng = np.random.default_rng(7)
    radii = (10.0, 10.0, 4.0)
    i = np.arange(N) + 0.5
    phi, th = np.arccos(1 - 2 * i / N), np.pi * (1 + 5 ** 0.5) * i
    starts = np.round((np.c_[np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)] * list(radii) + [0, 0, radii[2] + 1.0]) * 4) / 4
    goals = np.c_[-starts[:, 0], -starts[:, 1], 2 * (radii[2] + 1.0) - starts[:, 2]]
    boxes = []
    while len(boxes) < 24:
        c = np.r_[rng.uniform(-0.7 * radii[0], 0.7 * radii[0], 2), rng.uniform(1.5, 2 * radii[2] + 0.5)]
        if np.abs(starts - c).max(axis=1).min() > 1.2:
            boxes.append([c[0], c[1], c[2], 0.8, 0.8, 0.8])
    wmin, wmax = [-radii[0] - 2.0, -radii[1] - 2.0, 0.0], [radii[0] + 2.0, radii[1] + 2.0, 2 * radii[2] + 2.0]
    sol = api.Solver(api.make_desc(M=5, dim=3, dt=0.2, world_min=wmin, world_max=wmax))
    wmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, 20, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True, closed_loop=True)
    plan.reset(starts)
    its = []
    for k in range(replans + 1):
        pos = plan.get(api.PLAN_STATE).reshape(N, 9)[:, :3]
        d = goals - pos
        dist = np.linalg.norm(d, axis=1, keepdims=True)
        plan.put(api.PLAN_WAYPOINT, np.float32(pos + d / np.maximum(dist, 1e-9) * np.minimum(dist, 0.75)).astype(np.float64))
        plan.step(graph=True)
        torch.cuda.synchronize()
        if k >= 1:
            its.append(plan.get(api.PLAN_INFO)["iterations"].copy())
    its = np.array(its)
    from scipy.stats import spearmanr

    rho = [spearmanr(its[k - 1], its[k]).correlation for k in range(1, len(its)) if its[k].std() > 0 and its[k - 1].std() > 0]
    hit = []
    for k in range(1, len(its)):
        slow = np.argsort(-its[k], kind="stable")[: max(1, N // 10)]
        first_round = set(np.argsort(-its[k - 1], kind="stable")[: N // 4].tolist())
        hit.append(np.mean([q in first_round for q in slow]))
    print("closed loop, %d agents x %d replans: iterations mean %.2f max %d; rank correlation between consecutive replans %.2f (median); "
          "of the slowest 10 %% of a replan %.0f %% were among the slowest 25 %% of the previous one" % (
              N, len(its), its.mean(), its.max(), float(np.median(rho)), 100 * float(np.mean(hit))))
    plan.close()
    wmap.close()


if "--chain" in sys.argv:
    chain_hint_quality()

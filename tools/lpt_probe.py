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

for key in sys.argv[1:] or ["c3", "c4_f64", "c2"]:
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

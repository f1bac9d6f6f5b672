"""Development aid (the experiment BEFORE lscqp_construct_sfc_device_ordered existed; tests/test_sfc.py covers the entry point): does the corridor launch (construct_sfc FROM_HULL, 4096 agents, tools/bench_next_rows.py's world) have a tail like the QP
launch had?  Per-agent cost = the duration of a one-agent launch; then the whole launch is timed with the agents as given, most expensive
first, cheapest first and in a random order (inputs permuted on the host; the boxes must not change)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lsc_dr_planner_amd import api, synth  # noqa: E402

N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 5
dev = torch.device("cuda", 0)
sw = synth.Swarm(N, M=M, dim=3, n_obs=20, seed=1)
sol = api.Solver(api.make_desc(M=M, dim=3, world_min=sw.world_min, world_max=sw.world_max))
rng = np.random.default_rng(2)
wmin, wmax = np.array(sw.world_min, dtype=np.float64), np.array(sw.world_max, dtype=np.float64)
nb = int(float(np.prod(wmax - wmin)) / 8.0)
boxes = np.concatenate([rng.uniform(wmin, wmax, (nb, 3)), rng.choice([0.3, 0.5, 0.8], (nb, 3))], axis=1)
wm = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
wm.prepare(float(np.max(sw.radius)))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
starts = np.float32(sw.pos).astype(np.float64)
P = np.repeat(starts[:, None, :], 3, axis=1)
rad = np.full(N, sw.radius)
d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
d_st = torch.zeros(N, dtype=torch.int32, device=dev)
sol.construct_sfc_device(wm, api.SFC_INIT, N, up(P.reshape(-1)), up(rad), d_sfc, d_st)
torch.cuda.synchronize()
base = d_sfc.clone().view(N, M * 6)
step = rng.normal(size=(N, 3))
step /= np.linalg.norm(step, axis=1, keepdims=True)
P2 = np.float32(np.stack([starts + 0.3 * step, starts + 0.5 * step, starts + 0.5 * step], axis=1)).astype(np.float64)


def run(order, reps=20):
    dP, dr, b0 = up(P2[order].reshape(-1)), up(rad[order]), base[torch.from_numpy(order).to(dev)].contiguous()
    work = b0.clone()
    st = torch.zeros(len(order), dtype=torch.int32, device=dev)

    def once():
        work.copy_(b0)
        sol.construct_sfc_device(wm, api.SFC_FROM_HULL, len(order), dP, dr, work.view(-1), st)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(reps):
        work.copy_(b0)
    e1.record()
    for _ in range(reps):
        once()
    e2.record()
    torch.cuda.synchronize()
    return (e1.elapsed_time(e2) - e0.elapsed_time(e1)) / reps, work.clone()


ident = np.arange(N)
ms0, box0 = run(ident)
cost = np.zeros(N)
for i in range(N):  # one-agent launches: the agent's own cost
    cost[i], _ = run(np.array([i]), reps=3)
lpt = np.argsort(-cost, kind="stable")
res = {"as given": ms0}
for name, order in (("most expensive first", lpt), ("cheapest first", lpt[::-1].copy()), ("random", rng.permutation(N))):
    ms, bx = run(order)
    assert torch.equal(bx, box0[torch.from_numpy(order).to(dev)]), name
    res[name] = ms
print("corridors FROM_HULL, %d agents: one-agent launch %.1f us median, %.1f us max, %.1f us min; " % (N, np.median(cost) * 1e3, cost.max() * 1e3, cost.min() * 1e3) +
      " | ".join("%s %.4f ms" % kv for kv in res.items()))

"""Development aid: the corridor launch of 4096 agents (tools/bench_next_rows.py's world) as given and in the order
lscqp_order_by_cost_device makes of the costs the previous launch recorded (lscqp_construct_sfc_device_ordered)."""
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
d_r = up(np.full(N, sw.radius))
d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
d_st = torch.zeros(N, dtype=torch.int32, device=dev)
sol.construct_sfc_device(wm, api.SFC_INIT, N, up(np.repeat(starts[:, None, :], 3, axis=1).reshape(-1)), d_r, d_sfc, d_st)
torch.cuda.synchronize()
base = d_sfc.clone()
step = rng.normal(size=(N, 3))
step /= np.linalg.norm(step, axis=1, keepdims=True)
d_P2 = up(np.float32(np.stack([starts + 0.3 * step, starts + 0.5 * step, starts + 0.5 * step], axis=1)).astype(np.float64).reshape(-1))
d_cost = torch.zeros(N, dtype=torch.int32, device=dev)
d_order = torch.zeros(N, dtype=torch.int32, device=dev)


def timed(order, reps=20):
    def once():
        d_sfc.copy_(base)
        sol.construct_sfc_device(wm, api.SFC_FROM_HULL, N, d_P2, d_r, d_sfc, d_st, d_order=order, d_cost=d_cost)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(reps):
        d_sfc.copy_(base)
    e[1].record()
    for _ in range(reps):
        once()
    e[2].record()
    torch.cuda.synchronize()
    return (e[1].elapsed_time(e[2]) - e[0].elapsed_time(e[1])) / reps, d_sfc.clone()


ms0, b0 = timed(None)
sol.order_by_cost_device(N, d_cost, d_order)
ms1, b1 = timed(d_order)
assert torch.equal(b0, b1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    sol.order_by_cost_device(N, d_cost, d_order)
e1.record()
torch.cuda.synchronize()
print("corridors FROM_HULL, %d agents: as given %.4f ms | most expensive previous corridor first %.4f ms (%.2fx) | the sort itself %.4f ms" % (
    N, ms0, ms1, ms0 / ms1, e0.elapsed_time(e1) / 50))

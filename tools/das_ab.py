"""Development aid: A/B of two builds of the library on one box.  usage: python tools/das_ab.py libA.so libB.so [config ...]
Each (library, config) pair runs in its own process (bench.measure-style timing of the phase alone and of the whole call), three rounds, interleaved."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from lsc_dr_planner_amd import api, synth
key = sys.argv[1]
cfg = bench.CONFIGS[key]
N, M, dim = cfg["agents"], cfg["segments"], cfg["dim"]
out = {}
for mode, aset in (("on", api.ACTIVE_SET_DEFAULT), ("only", api.ACTIVE_SET_ONLY)):
    try:
        sw, sol, build, (hdr, rows, off, sfc) = bench.make_batch(api, synth, lambda s: api.Solver(api.make_desc(M=M, dim=dim, world_min=s.world_min, world_max=s.world_max, active_set=aset)),
                                                                 N, M, dim, cfg["obs"], seed=cfg["seed"], style=cfg["style"], warm_steps=3)
    except RuntimeError as ex:
        out[mode] = "FAILED in make_batch: " + str(ex)
        continue
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
    d_xi = torch.from_numpy(np.ascontiguousarray(api.x_init_from_swarm(build, dim))).to(dev)
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev); d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev); d_info = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    if os.environ.get('AB_QUICK'):
        out[mode] = 'ok'
        continue
    for _ in range(20):
        sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, d_info, d_x_init=d_xi)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    out[mode] = round(best, 2)
print(json.dumps(out))
''' % ROOT

nl = int(os.environ.get("AB_NLIBS", "2"))
libs = sys.argv[1:1 + nl]
cfgs = sys.argv[1 + nl:] or ["c1"]
for key in cfgs:
    res = {l: [] for l in libs}
    for rnd in range(int(os.environ.get('AB_ROUNDS', '3'))):
        for l in libs:
            env = dict(os.environ, LSCQP_LIB=os.path.abspath(l))
            o = subprocess.run([sys.executable, "-c", CHILD, key], env=env, capture_output=True, text=True)
            line = [x for x in o.stdout.splitlines() if x.startswith("{")]
            res[l].append(json.loads(line[-1]) if line else o.stderr[-300:])
    for l in libs:
        print(key, os.path.basename(l), res[l], flush=True)

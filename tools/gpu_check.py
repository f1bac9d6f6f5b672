"""Exploratory GPU check (development aid): HIP solver vs CPU oracle on small synthetic swarms."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from lsc_dr_planner_amd import synth, api


def run(N, M, dim, n_obs, steps, seed, style="forest"):
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    for step in range(steps + 1):
        b = sw.build()
        ag = np.zeros(N, O.AGENT_DTYPE)
        for f in ('p0', 'v0', 'a0', 'goal', 'next_waypoint'):
            ag[f] = b[f]
        ag['vmax'] = 1; ag['amax'] = 2; ag['radius'] = 0.15; ag['nominal_velocity'] = 1; ag['n_obs'] = sw.n_obs
        lsc = np.ascontiguousarray(b['lsc']).reshape(-1)
        off = np.arange(N) * sw.n_obs * M * 6
        t0 = time.time()
        R = O.solve_batch(cls, ag, lsc, off, np.ascontiguousarray(b['sfc']).reshape(-1), threads=8)
        t1 = time.time()
        hdr, rows, roff, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        hdr['terminal_segments'] = [O.terminal_segments(cls, ag[q:q + 1]) for q in range(N)]
        G = sol.solve_host(hdr, rows, roff, sfc)
        t2 = time.time()
        ok = (G['status'] == 0) & (R['status'] == 0)
        dx = np.abs(G['x'] - R['x']).max(axis=1)
        do = np.abs(G['obj'] - R['obj']) / np.maximum(1, np.abs(R['obj']))
        print(f"[{style} N{N} M{M} d{dim} o{sw.n_obs}] step {step}: oracle bad {(R['status']!=0).sum()} it {R['iters'].mean():.1f}/{R['iters'].max()} "
              f"({(t1-t0)*1e3:.0f} ms) | gpu status {np.bincount(G['status'], minlength=4)} it {G['info']['iterations'].mean():.1f}/{G['info']['iterations'].max()} "
              f"({(t2-t1)*1e3:.0f} ms) | max|dx| {dx[ok].max() if ok.any() else -1:.2e} max rel dobj {do[ok].max() if ok.any() else -1:.2e} "
              f"| res p {G['info']['res_primal'].max():.1e} d {G['info']['res_dual'].max():.1e} gap {G['info']['gap'].max():.1e}", flush=True)
        if not ok.all():
            bad = np.where(~ok)[0][:5]
            print("   bad:", bad, G['status'][bad], G['info'][bad])
        sw.advance(R['x'])


if __name__ == "__main__":
    print(api.lib().lscqp_version().decode())
    run(16, 5, 3, 8, 3, 1)
    run(64, 5, 3, 20, 4, 1)
    run(10, 10, 2, 9, 3, 2)
    run(48, 6, 3, 20, 3, 3, "maze")

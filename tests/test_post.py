"""Post-solve epilogue: isSolValid + getStateAt + doStep on the device against the oracle."""
import numpy as np
import pytest

from tests import helpers as H


def test_validate_step_oracle_on_solved_swarm(oracle):
    """Solutions of the QP are valid by construction; pushing a control point out of its box or stretching the first
    segment beyond the velocity limit must flip the verdict (reference src/traj_planner.cpp:992-1042)."""
    from lsc_dr_planner_amd import synth

    N, M, dim = 12, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=6, seed=11)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    b = sw.build()
    ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, b)
    R = oracle.solve_batch(cls, ag, lsc, off, sfc, threads=4)
    assert (R["status"] == 0).all()
    for q in range(N):
        ok, st = oracle.validate_step(cls, ag[q], b["sfc"][q], R["x"][q], 0.1)
        assert ok == 1
        pos, vel, acc = oracle.state_at(cls, np.float32(R["x"][q]).astype(np.float64), 0.1)
        assert np.allclose(st[:3], np.float32(pos)) and np.allclose(st[3:6], np.float32(vel)) and np.allclose(st[6:], np.float32(acc))
        x2 = R["x"][q].copy()
        x2[0 * M * 6 + 6 * 2 + 4] = b["sfc"]["bmax"][q, 2, 0] + 1e-3  # control point (m=2, i=4) beyond its box in x
        assert oracle.validate_step(cls, ag[q], b["sfc"][q], x2, 0.1)[0] == 0
        x3 = R["x"][q].copy()
        x3[0 * M * 6 + 3:0 * M * 6 + 6] += 2.0  # 2 m within one segment: |v| far above 1.01 * vmax
        assert oracle.validate_step(cls, ag[q], None if False else b["sfc"][q], x3, 0.1)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,n_obs,seed", [(32, 5, 3, 12, 3), (10, 10, 2, 9, 2)])
def test_gpu_validate_step_matches_oracle(api, oracle, N, M, dim, n_obs, seed):
    import torch

    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    r = sol.solve_host(hdr, rows, off, sfc)
    assert (r["status"] == 0).all()
    x = r["x"].copy()
    # make a third of the batch invalid in the two ways isSolValid knows
    for q in range(0, N, 3):
        if q % 2 == 0:
            x[q, 0 * M * 6 + 6 * (M - 1) + 2] = b["sfc"]["bmin"][q, M - 1, 0] - 1e-3
        else:
            x[q, 1 * M * 6 + 3:1 * M * 6 + 6] += 2.0
    z2d = float(b["p0"][0][2])
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_x = torch.from_numpy(x).to(dev)
    d_valid = torch.full((N,), -1, dtype=torch.int32, device=dev)
    d_state = torch.zeros(N * 9, dtype=torch.float64, device=dev)
    for ts in (0.1, 0.2):  # multisim_time_step < dt and == dt (reference src/param.cpp:134-148)
        sol.validate_step_device(N, ts, d_x, up(hdr), up(sfc), d_valid, d_state, z_2d=z2d)
        torch.cuda.synchronize()
        got_v, got_s = d_valid.cpu().numpy(), d_state.cpu().numpy().reshape(N, 9)
        ag, _, _, _ = H.swarm_oracle_inputs(oracle, sw, b)
        for q in range(N):
            ok, st = oracle.validate_step(cls, ag[q], b["sfc"][q], x[q], ts, z2d)
            assert got_v[q] == ok, (q, ts)
            # same float32 control points, fp64 Bernstein evaluation on both sides, float32 result: at most one ulp
            assert np.allclose(got_s[q], st, rtol=2e-7, atol=1e-7), (q, got_s[q], st)
        assert got_v.sum() < N and got_v.sum() >= N - (N + 2) // 3


# ---- safety metrics (reference src/multi_sync_simulator.cpp:486-577) -----------------------------------------------
def _const_plans(pos, M, dim):
    """Plans whose control points all equal `pos` (n, 3): getStateAt(0) returns exactly that position."""
    n = pos.shape[0]
    x = np.zeros((n, dim, M, 6))
    for k in range(dim):
        x[:, k] = pos[:, k, None, None]
    return x.reshape(n, -1)


def test_safety_ratio_oracle_reproduces_reference_summary(oracle):
    """The reference's own run: positions of its 10 agents at every logged time (log/simulation_*_LSC_10agents.csv) must
    give the safety_ratio_agent of its summary CSV (1.02089) and zero velocity / acceleration excess."""
    g = H.load_golden("sim_log_states")
    M, dim = 10, 2
    cls = oracle.make_class(M=M, dim=dim, use_sfc=False, world_min=[-5, -5, 0], world_max=[5, 5, 2.5])
    ag = np.zeros(10, oracle.AGENT_DTYPE)
    ag["vmax"], ag["amax"] = g["vmax"], g["amax"]
    best = np.inf
    for p in g["pos"]:
        p = np.array(p)
        out = oracle.safety_metrics(cls, ag, _const_plans(p, M, dim), g["radius"], g["downwash"], 1, 0.1, z_2d=p[0, 2])
        best = min(best, out[:, 0].min())
    # the log prints 6 significant digits -> positions to 5e-6 m -> ratio to ~3e-5
    assert abs(best - g["summary"]["safety_ratio_agent"]) <= 5e-5, best
    v, a = np.array(g["vel"]), np.array(g["acc"])
    assert max(0.0, ((v - g["vmax"]) / g["vmax"]).max()) == g["summary"]["vel_excess_ratio"] == 0.0
    assert max(0.0, ((a - g["amax"]) / g["amax"]).max()) == g["summary"]["acc_excess_ratio"] == 0.0


def test_safety_metrics_oracle_against_numpy(oracle):
    from lsc_dr_planner_amd import synth

    N, M, dim = 14, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=6, seed=5)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    b = sw.build()
    ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, b)
    R = oracle.solve_batch(cls, ag, lsc, off, sfc, threads=4)
    ag["vmax"][:, 0] = 0.002  # for the metrics only: a limit the plans exceed in +x (the reference's ratio is signed)
    rad = np.full(N, sw.radius)
    dwv = np.full(N, sw.downwash)
    rad[2], dwv[2] = 0.25, 1.2
    out = oracle.safety_metrics(cls, ag, R["x"], rad, dwv, 3, 0.05)
    xf = np.float32(R["x"]).astype(np.float64)
    for a in range(N):
        best, bkey, vex = np.inf, None, 0.0
        for s in range(3):
            pa, va, _ = oracle.state_at(cls, xf[a], s * 0.05)
            vex = max(vex, (np.float32(va[0]) - 0.002) / 0.002)
            for j in range(N):
                if j == a:
                    continue
                pj, _, _ = oracle.state_at(cls, xf[j], s * 0.05)
                dwn = (dwv[a] * rad[a] + dwv[j] * rad[j]) / (rad[a] + rad[j])
                d = np.float32(pa) - np.float32(pj)
                d[2] = np.float32(d[2] / dwn)
                r = np.sqrt(float(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))) / (rad[a] + rad[j])
                if r < best:
                    best, bkey = r, (j, s)
        assert abs(out[a, 0] - best) <= 1e-7 * best and (out[a, 1], out[a, 2]) == bkey
        assert abs(out[a, 3] - max(vex, 0.0)) <= 1e-6
    assert out[:, 3].max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,n_obs,first,n_loc,seed", [(64, 5, 3, 12, 0, 64, 3), (300, 5, 3, 8, 100, 77, 4), (10, 10, 2, 9, 0, 10, 2)])
def test_gpu_safety_metrics_match_oracle(api, oracle, N, M, dim, n_obs, first, n_loc, seed):
    import torch

    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    r = sol.solve_host(hdr, rows, off, sfc)
    x = r["x"]
    hdr["vmax"][:, 0] = 0.002  # for the metrics only: a limit the plans exceed in +x (the reference's ratio is signed)
    rad = np.full(N, sw.radius)
    dwv = np.full(N, sw.downwash)
    rad[1], dwv[1] = 0.25, 1.2
    z2d = float(b["p0"][0][2])
    ag = np.zeros(n_loc, oracle.AGENT_DTYPE)
    ag["vmax"], ag["amax"] = hdr["vmax"][first:first + n_loc], hdr["amax"][first:first + n_loc]
    want = oracle.safety_metrics(cls, ag, x, rad, dwv, 2, 0.05, first=first, z_2d=z2d)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_out = torch.zeros(n_loc * api.SAFETY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    sol.safety_metrics_device(n_loc, first, N, 2, 0.05, torch.from_numpy(x.copy()).to(dev), torch.from_numpy(rad).to(dev),
                              torch.from_numpy(dwv).to(dev), up(hdr[first:first + n_loc]), d_out, z_2d=z2d)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(api.SAFETY_DTYPE)
    # same float32 states and float32 distance arithmetic on both sides: the Bernstein evaluation may differ in the last
    # fp64 bit before the float32 rounding of a position (one float32 ulp of a 10 m coordinate = 1e-6 m)
    assert np.abs(got["safety_ratio"] - want[:, 0]).max() <= 1e-5
    same = got["closest_agent"] == want[:, 1].astype(np.int32)
    assert same.mean() >= 0.98 and (got["sample"][same] == want[same, 2].astype(np.int32)).all()  # exact ties may swap
    assert np.abs(got["vel_excess_ratio"] - want[:, 3:6]).max() <= 1e-5 and np.abs(got["acc_excess_ratio"] - want[:, 6:9]).max() <= 1e-5
    assert got["vel_excess_ratio"][:, 0].max() > 0  # the tightened limit is really exceeded somewhere


def _obstacle_table(api, N, dim, world_min, world_max, rng):
    """A handful of non-agent obstacles as the constraint generator takes them (lscqp_obstacle), one of them "real" (skipped by the metric)."""
    obs = np.zeros(7, api.OBSTACLE_DTYPE)
    lo, hi = np.array(world_min, dtype=np.float64), np.array(world_max, dtype=np.float64)
    obs["position"] = lo + rng.uniform(0.1, 0.9, (7, 3)) * (hi - lo)
    obs["velocity"] = rng.uniform(-0.5, 0.5, (7, 3))
    obs["radius"] = rng.uniform(0.1, 0.6, 7)
    obs["downwash"] = rng.uniform(1.0, 3.0, 7)
    obs["max_acc"] = 1.0
    obs["type"] = 0
    obs["type"][3] = api.OBSTACLE_REAL  # :531-532
    obs["position"][3] = obs["position"][0]  # would win every minimum if it were not skipped ...
    obs["radius"][3] = 1e-3                  # (tiny radius sum -> huge ratio? no: make it the NEAREST in ratio terms below)
    return obs


def test_safety_obstacles_oracle_against_numpy(oracle, api):
    """safety_ratio_obs (reference src/multi_sync_simulator.cpp:527-557) restated in numpy, expression by expression: the mixed
    downwash (:538-540), ellipsoidalDistance in point3d = float32 arithmetic (include/util.hpp:155-159), the ratio in double, the first
    strict minimum in (sample, obstacle) order, "real" obstacles skipped, +inf / -1 when there is nothing to compare with."""
    from lsc_dr_planner_amd import synth

    N, M, dim = 9, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=4, seed=11)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    b = sw.build()
    ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, b)
    R = oracle.solve_batch(cls, ag, lsc, off, sfc, threads=4)
    rng = np.random.default_rng(5)
    obs = _obstacle_table(api, N, dim, sw.world_min, sw.world_max, rng)
    tab = np.c_[obs["position"], obs["radius"], obs["downwash"]]
    skip = (obs["type"] == api.OBSTACLE_REAL).astype(np.int32)
    rad, dwv = np.full(N, sw.radius), np.full(N, sw.downwash)
    rad[2], dwv[2] = 0.25, 1.2
    out = oracle.safety_obstacles(cls, N, R["x"], rad, dwv, tab, 3, 0.05, skip=skip)
    xf = np.float32(R["x"]).astype(np.float64)
    for a in range(N):
        best, key = np.inf, (-1, -1)
        for s in range(3):
            pa, _, _ = oracle.state_at(cls, xf[a], s * 0.05)
            for o in range(len(obs)):
                if skip[o]:
                    continue
                dwn = (obs["radius"][o] * obs["downwash"][o] + rad[a] * dwv[a]) / (rad[a] + obs["radius"][o])
                d = np.float32(pa) - np.float32(obs["position"][o])
                d[2] = np.float32(np.float64(d[2]) / dwn)
                r = np.sqrt(float(np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))) / (rad[a] + obs["radius"][o])
                if r < best:
                    best, key = r, (o, s)
        assert out[a, 0] == best and (out[a, 1], out[a, 2]) == key, (a, out[a], best, key)
    assert not (out[:, 1] == 3).any()
    # no obstacle that counts: SP_INFINITY, no index
    none = oracle.safety_obstacles(cls, N, R["x"], rad, dwv, tab[3:4], 3, 0.05, skip=np.ones(1, np.int32))
    assert np.isinf(none[:, 0]).all() and (none[:, 1] == -1).all() and (none[:, 2] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,first,n_loc,seed", [(64, 5, 3, 0, 64, 3), (300, 5, 3, 100, 77, 4), (10, 10, 2, 0, 10, 2)])
def test_gpu_safety_obstacles_match_oracle(api, oracle, N, M, dim, first, n_loc, seed):
    """lscqp_safety_obstacles_device against the oracle's restatement of src/multi_sync_simulator.cpp:527-557 on solved plans, with the
    obstacle table lscqp_generate_lsc_obstacles_device takes (a "real" entry included, which both sides skip)."""
    import torch

    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=8, seed=seed)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    x = sol.solve_host(hdr, rows, off, sfc)["x"]
    rng = np.random.default_rng(seed)
    obs = _obstacle_table(api, N, dim, sw.world_min, sw.world_max, rng)
    z2d = float(b["p0"][0][2])
    if dim == 2:
        obs["position"][:, 2] = z2d + rng.uniform(-0.3, 0.3, len(obs))
    rad, dwv = np.full(N, sw.radius), np.full(N, sw.downwash)
    rad[1], dwv[1] = 0.25, 1.2
    skip = (obs["type"] == api.OBSTACLE_REAL).astype(np.int32)
    want = oracle.safety_obstacles(cls, n_loc, x, rad, dwv, np.c_[obs["position"], obs["radius"], obs["downwash"]], 2, 0.05, first=first, z_2d=z2d, skip=skip)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_out = torch.zeros(n_loc * api.SAFETY_OBS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    sol.safety_obstacles_device(n_loc, first, N, 2, 0.05, torch.from_numpy(x.copy()).to(dev), torch.from_numpy(rad).to(dev), torch.from_numpy(dwv).to(dev),
                                len(obs), up(obs), d_out, z_2d=z2d)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(api.SAFETY_OBS_DTYPE)
    # (the Bernstein evaluation of a position may differ in the last fp64 bit before its float32 rounding: one float32 ulp of a 10 m coordinate)
    assert np.abs(got["safety_ratio_obs"] - want[:, 0]).max() <= 1e-5 and not (got["closest_obstacle"] == 3).any()
    same = got["closest_obstacle"] == want[:, 1].astype(np.int32)
    assert same.mean() >= 0.98 and (got["sample"][same] == want[same, 2].astype(np.int32)).all()
    exact = got["safety_ratio_obs"] == want[:, 0]
    assert exact.mean() >= 0.9  # IEEE division and square root on both sides: bit-for-bit wherever the float32 positions agree
    # nothing to compare with: SP_INFINITY and no index, like the reference's untouched running minimum
    sol.safety_obstacles_device(n_loc, first, N, 2, 0.05, torch.from_numpy(x.copy()).to(dev), torch.from_numpy(rad).to(dev), torch.from_numpy(dwv).to(dev),
                                0, None, d_out, z_2d=z2d)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(api.SAFETY_OBS_DTYPE)
    assert np.isinf(got["safety_ratio_obs"]).all() and (got["closest_obstacle"] == -1).all() and (got["sample"] == -1).all()

"""LSC generation: the oracle restatement against the REFERENCE's openGJK, and the HIP kernel
against the oracle.  Tolerances are stated where they are used."""
import os

import numpy as np
import pytest

from tests import helpers as H


def test_hull_closest_point_against_reference_gjk_golden(oracle):
    """Committed outputs of the reference's own openGJK (tools/make_golden_gjk.py): distance and closest point of 240
    six-point hulls (generic, planar, repeated points, float32 coordinates, origin inside)."""
    g = H.load_golden("gjk_hulls")
    worst = 0.0
    for c in g["cases"]:
        d, p = oracle.hull_closest_point(np.array(c["hull"]))
        worst = max(worst, abs(d - c["dist"]), np.abs(p - np.array(c["closest"])).max())
    assert worst <= 1e-9, worst  # both are fp64 solutions of the same unique projection


def test_hull_closest_point_against_reference_gjk_live(oracle):
    """Same check against oracle/_ref/libref_gjk.so (the reference's openGJK compiled where it lies), on fresh hulls."""
    if oracle.build_ref() is None:
        pytest.skip("oracle/_ref not built (no reference checkout on this machine); the golden test above covers it")
    rng = np.random.default_rng(7)
    worst = 0.0
    for t in range(3000):
        c = rng.normal(size=3) * rng.uniform(0.2, 3)
        pts = c + rng.normal(size=(6, 3)) * rng.uniform(0.01, 1.0)
        if t % 4 == 0:
            pts[:, 2] = 0
        if t % 9 == 0:
            pts[4:] = pts[:2]
        d, p = oracle.hull_closest_point(pts)
        dr, v = oracle.ref_gjk(pts)
        worst = max(worst, abs(d - dr), np.abs(p - v).max())
    assert worst <= 1e-9, worst


def _swarm_inputs(N, M, dim, n_obs, seed):
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    b = sw.build()
    return sw, b


@pytest.mark.parametrize("N,M,dim,n_obs,seed", [(24, 5, 3, 8, 1), (12, 10, 2, 5, 2)])
def test_generate_lsc_restatement_matches_independent_numpy_and_invariants(oracle, N, M, dim, n_obs, seed):
    sw, b = _swarm_inputs(N, M, dim, n_obs, seed)
    L = oracle.generate_lsc(b["init"], b["nbr"], sw.radius, sw.downwash, b["goal"], dim=dim)
    # (a) the workload generator's numpy restatement (written independently, vectorised): float32 staging -> 2e-7
    for f in ("p", "nrm", "d"):
        assert np.abs(L[f] - b["lsc"][f]).max() <= 2e-7, f
    # (b) the feasibility invariant of SURVEY.md section 8d: the agent's own initial control point satisfies its row with
    #     slack 1/2 (rel.n - (r_i + r_j)) >= 0 when the hulls are collision free
    own = b["init"][:, None]  # (N,1,M,6,3)
    slack = np.einsum("nkmic,nkmic->nkmi", own - L["p"], L["nrm"]) - L["d"]
    assert slack.min() >= -1e-6
    # (c) the neighbour's row is the mirror image: -normal (3-D normals are unit in downwash-scaled coordinates)
    for a in range(N):
        for oi, j in enumerate(b["nbr"][a]):
            back = np.where(b["nbr"][j] == a)[0]
            if len(back):
                assert np.abs(L["nrm"][a, oi] + L["nrm"][j, back[0]]).max() <= 2e-7


def test_generate_lsc_fallback_when_hulls_overlap(oracle):
    """Hull of the relative control points contains the origin -> normal = (goal - obstacle position) normalised,
    reference src/traj_planner.cpp:624-633."""
    M = 3
    traj = np.zeros((2, M, 6, 3))
    rng = np.random.default_rng(3)
    traj[0] = np.float32(rng.normal(size=(M, 6, 3)) * 0.5)
    traj[1] = np.float32(rng.normal(size=(M, 6, 3)) * 0.5)
    goal = np.array([[3.0, 4.0, 0.0]])
    L = oracle.generate_lsc(traj, np.array([[1]]), 0.15, 1.0, goal, dim=3)
    for m in range(M):
        rel = np.float32(traj[0, m]) - np.float32(traj[1, m])
        d, _ = oracle.hull_closest_point(rel.astype(np.float64))
        if d == 0.0:
            fb = goal[0] - traj[1, 0, 0]
            assert np.abs(L["nrm"][0, 0, m, 0] - fb / np.linalg.norm(fb)).max() <= 2e-7


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,n_obs,seed", [(64, 5, 3, 20, 1), (10, 10, 2, 9, 2), (48, 6, 3, 20, 3), (44, 10, 3, 40, 8)])
def test_gpu_generate_lsc_matches_oracle(api, oracle, N, M, dim, n_obs, seed):
    import torch

    assert torch.cuda.is_available()
    sw, b = _swarm_inputs(N, M, dim, n_obs, seed)
    assert sw.n_obs == n_obs
    # a few missing neighbours (-1) and one overlapping pair exercise the zero-row and the fallback branches
    nbr = b["nbr"].astype(np.int32).copy()
    nbr[0, -1] = -1
    init = b["init"].copy()
    init[1] = init[nbr[1, 0]]  # agent 1 sits exactly on its first neighbour
    L = oracle.generate_lsc(init, nbr, sw.radius, sw.downwash, b["goal"], dim=dim)
    want = api.pack_rows(L).reshape(N, n_obs, M, 6)
    dev = torch.device("cuda", 0)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    d_traj = torch.from_numpy(init.copy()).to(dev)
    d_nbr = torch.from_numpy(nbr).to(dev)
    d_r = torch.full((N,), sw.radius, dtype=torch.float64, device=dev)
    d_dw = torch.full((N,), sw.downwash, dtype=torch.float64, device=dev)
    d_goal = torch.from_numpy(np.ascontiguousarray(b["goal"], dtype=np.float64)).to(dev)
    d_rows = torch.full((N * n_obs * M * 6 * 4,), float("nan"), dtype=torch.float64, device=dev)
    sol.generate_lsc_device(N, n_obs, 0, d_traj, d_nbr, d_r, d_dw, d_goal, d_rows)
    torch.cuda.synchronize()
    got = d_rows.cpu().numpy().view(api.ROW_DTYPE).reshape(N, n_obs, M, 6)
    assert (got["nx"][0, -1] == 0).all() and (got["b"][0, -1] == 0).all()  # missing neighbour -> all-zero rows
    # Same float32 staging on both sides; the fp64 enumeration differs only in operation order (1/det vs /det, FMA
    # contraction), which can move a float32 rounding by one ulp: 2e-7 on the unit normal, 2e-6 on b = d + n.p (|p| <~ 30 m)
    for f, tol in (("nx", 2e-7), ("ny", 2e-7), ("nz", 2e-7), ("b", 2e-6)):
        assert np.abs(got[f] - want[f]).max() <= tol, (f, np.abs(got[f] - want[f]).max())
    assert np.isfinite(d_rows.cpu().numpy()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [3, 2])
def test_gpu_fallback_normal_uses_the_obstacles_current_position_on_moving_plans(api, oracle, dim):
    """Overlapping hulls on MOVING trajectories: the fallback normal is goal - obstacle POSITION (the first control point of the
    neighbour's shifted plan) for every segment (reference src/traj_planner.cpp:629-631), not goal - first point of segment m.
    With a stationary swarm (all control points equal) the two cannot be told apart; here every segment of the neighbour starts
    somewhere else."""
    import torch

    N, M, n_obs = 6, 5, 2
    rng = np.random.default_rng(17)
    init = np.zeros((N, M, 6, 3))
    for a in range(N):
        start = rng.uniform(-1, 1, 3)
        vel = rng.uniform(-1.0, 1.0, 3)
        t = (np.arange(M * 6) * 0.04).reshape(M, 6)
        init[a] = start + vel * t[..., None] + 0.02 * rng.normal(size=(M, 6, 3))
    if dim == 2:
        init[..., 2] = 1.0
    init = np.float32(init).astype(np.float64)
    nbr = np.array([[1, 2], [0, 2], [0, 1], [4, 5], [3, 5], [3, 4]], dtype=np.int32)
    init[1] = init[0] + np.float32(1e-3)   # pairs (0,1) and (3,4) overlap in every segment -> fallback everywhere
    init[4] = init[3]
    goal = np.float32(rng.uniform(-3, 3, (N, 3))).astype(np.float64)
    if dim == 2:
        goal[:, 2] = 1.0
    L = oracle.generate_lsc(init, nbr, 0.15, 2.0, goal, dim=dim)
    want = api.pack_rows(L).reshape(N, n_obs, M, 6)
    # the oracle itself: fallback direction = goal - first control point of the neighbour's plan, for the LAST segment too
    fb = goal[3] - init[4, 0, 0]
    if dim == 2:
        fb[2] = 0.0
    else:
        fb[2] /= 2.0
    fb = fb / np.linalg.norm(fb)
    nz_out = fb[2] / 2.0 if dim == 3 else 0.0
    assert np.abs(L["nrm"][3, 0, M - 1, 0] - np.array([fb[0], fb[1], nz_out])).max() <= 3e-7
    dev = torch.device("cuda", 0)
    sol = api.Solver(api.make_desc(M=M, dim=dim))
    d_rows = torch.full((N * n_obs * M * 6 * 4,), float("nan"), dtype=torch.float64, device=dev)
    sol.generate_lsc_device(N, n_obs, 0, torch.from_numpy(init.copy()).to(dev), torch.from_numpy(nbr).to(dev),
                            torch.full((N,), 0.15, dtype=torch.float64, device=dev), torch.full((N,), 2.0, dtype=torch.float64, device=dev),
                            torch.from_numpy(goal.copy()).to(dev), d_rows)
    torch.cuda.synchronize()
    got = d_rows.cpu().numpy().view(api.ROW_DTYPE).reshape(N, n_obs, M, 6)
    for f, tol in (("nx", 2e-7), ("ny", 2e-7), ("nz", 2e-7), ("b", 2e-6)):
        assert np.abs(got[f] - want[f]).max() <= tol, (f, np.abs(got[f] - want[f]).max())


@pytest.mark.gpu
def test_gpu_generated_rows_feed_the_solver(api, oracle):
    """shift -> generate -> solve entirely on the device equals the host pipeline (rows from the oracle restatement)."""
    import torch

    N, M, dim, n_obs = 32, 5, 3, 12
    sw, b = _swarm_inputs(N, M, dim, n_obs, 5)
    dev = torch.device("cuda", 0)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    r0 = sol.solve_host(hdr, rows, off, sfc)
    assert (r0["status"] == 0).all()
    sw.advance(r0["x"])
    b1 = sw.build()  # host pipeline: shift + LSC (numpy) of the next replan
    # device pipeline from the same solution
    d_x = torch.from_numpy(r0["x"].copy()).to(dev)
    d_traj = torch.zeros(N * M * 6 * 3, dtype=torch.float64, device=dev)
    sol.shift_traj_device(N, d_x, d_traj)
    torch.cuda.synchronize()
    assert np.array_equal(d_traj.cpu().numpy().reshape(N, M, 6, 3), b1["init"])  # bit-exact: float32 rounding of the same values
    d_nbr = torch.from_numpy(b1["nbr"].astype(np.int32)).to(dev)
    d_r = torch.full((N,), sw.radius, dtype=torch.float64, device=dev)
    d_dw = torch.full((N,), sw.downwash, dtype=torch.float64, device=dev)
    d_goal = torch.from_numpy(np.ascontiguousarray(b1["goal"], dtype=np.float64)).to(dev)
    d_rows = torch.zeros(N * n_obs * M * 6 * 4, dtype=torch.float64, device=dev)
    sol.generate_lsc_device(N, n_obs, 0, d_traj, d_nbr, d_r, d_dw, d_goal, d_rows)
    hdr1, rows1, off1, sfc1 = api.batch_from_swarm(b1, sw.n_obs, M)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_hdr, d_off, d_sfc = up(hdr1), up(off1), up(sfc1)
    d_xo = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    sol.solve_device(N, n_obs, d_hdr, d_rows, d_off, d_sfc, d_xo, d_obj, d_st)
    torch.cuda.synchronize()
    r1 = sol.solve_host(hdr1, rows1, off1, sfc1)
    assert (d_st.cpu().numpy() == 0).all() and (r1["status"] == 0).all()
    # rows agree to float32 rounding (previous test) -> optima agree far inside the parity tolerances
    assert np.abs(d_xo.cpu().numpy().reshape(N, -1) - r1["x"]).max() <= 1e-5
    assert np.abs(d_obj.cpu().numpy() - r1["obj"]).max() <= 1e-6 * max(1.0, np.abs(r1["obj"]).max())

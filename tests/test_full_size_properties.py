"""The kernels either side of the QP at BASELINE's largest batch (4096 agents), where the CPU oracle is too slow to be
the checker: size-independent properties of their outputs, verified with vectorised numpy / scipy (no oracle)."""
import numpy as np
import pytest


def _setup(api, N, M, dim, n_obs, seed):
    import torch

    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    init = sw.initial_traj()
    nbr = sw.neighbours().astype(np.int32)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    return torch, sw, init, nbr, sol


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_constraint_rows_at_4096_agents(api, mode):
    N, M, dim, n_obs = 4096, 5, 3, 20
    torch, sw, init, nbr, sol = _setup(api, N, M, dim, n_obs, 1)
    dev = torch.device("cuda", 0)
    nbr[::97, -1] = -1
    rng = np.random.default_rng(7)
    goal_all = np.float32(init[:, M - 1, 5] + rng.normal(size=(N, 3)) * 0.6).astype(np.float64)  # goal points in all directions
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d_rows = torch.full((N * n_obs * M * 6 * 4,), float("nan"), dtype=torch.float64, device=dev)
    sol.generate_constraints_device(mode, N, n_obs, 0, up(init), up(nbr), up(np.full(N, sw.radius)), up(np.full(N, sw.downwash)),
                                    up(goal_all), d_rows)
    torch.cuda.synchronize()
    R = d_rows.cpu().numpy().reshape(N, n_obs, M, 6, 4)
    assert np.isfinite(R).all()
    missing = nbr < 0
    assert (R[missing] == 0).all()                      # no neighbour -> all-zero rows
    n, b = R[..., :3], R[..., 3]
    dw, rsum = sw.downwash, 2 * sw.radius
    nt = n * np.array([1, 1, dw])                       # back in the downwash-scaled frame the normal is a unit vector
    ln = np.linalg.norm(nt, axis=-1)
    live = ~missing[:, :, None, None] & (ln > 1e-5)
    assert np.abs(ln[live] - 1).max() <= 1e-6
    own = init[:, None]                                 # (N, 1, M, 6, 3)
    obs = init[np.where(missing, 0, nbr)]               # (N, n_obs, M, 6, 3)
    S = np.array([1, 1, 1 / dw])
    if mode == 2:                                       # generateBVC: one normal from the start points, one margin (:716-727)
        diff = (own[:, :, :1, :1] - obs[:, :, :1, :1]) * S
        want_n = diff / np.maximum(np.linalg.norm(diff, axis=-1, keepdims=True), 1e-300)
        assert np.abs(np.where(live[..., None], nt - want_n, 0)).max() <= 2e-6
        want_b = 0.5 * (rsum + np.linalg.norm(diff, axis=-1)) + (n * obs).sum(-1)
        assert np.abs(np.where(live, b - want_b, 0)).max() <= 2e-5
        return
    seg = slice(0, M) if mode == 0 else slice(0, M - 1)  # CLSC: the last segment is a different construction
    rel = (own - obs) * S
    # margin identity of generateLSC / generateCLSC (:641-643, :683-686): d_i = (r_i + r_j + rel_i . n) / 2, packed b = d + n . p_obs
    want_b = 0.5 * (rsum + (rel * nt).sum(-1)) + (n * obs).sum(-1)
    assert np.abs(np.where(live, b - want_b, 0)[:, :, seg]).max() <= 2e-5
    # the normal is the direction of the hull's closest point: no relative control point lies behind the supporting plane
    proj = (rel * nt).sum(-1)                           # (N, n_obs, M, 6)
    dist = proj.min(-1, keepdims=True)                  # = distance of the hull from the origin when the normal is right
    inside = ~live
    assert (np.where(inside, 1, dist)[:, :, seg] > 0).all()
    # ... and that point is on the hull: the closest vertex cannot be nearer than the plane distance, and for a correct
    # normal some convex combination attains it: check via the dual bound |sum_i w_i rel_i| >= dist for the uniform weights
    cen = rel.mean(-2)
    assert (np.where(inside[..., 0], 1e9, np.linalg.norm(cen, axis=-1) - dist[..., 0])[:, :, seg] >= -1e-6).all()
    # own control points keep the slack (rel . n - (r_i + r_j)) / 2 -> non-negative for the collision-free swarm
    slack = (n * own).sum(-1) - b
    assert np.abs(np.where(live, slack - 0.5 * (proj - rsum), 0)[:, :, seg]).max() <= 2e-5
    if mode == 1:                                       # CLSC last segment: one plane per pair, mirrored for the pair (:691-703)
        last_n, last_b = nt[:, :, M - 1], b[:, :, M - 1]
        assert np.abs(last_n - last_n[:, :, :1]).max() == 0 and np.abs(last_b - last_b[:, :, :1]).max() == 0
        # mutual neighbours see opposite normals
        a_idx, o_idx = np.nonzero(~missing)
        j = nbr[a_idx, o_idx]
        back = (nbr[j] == a_idx[:, None])
        has = back.any(axis=1)
        ob = back.argmax(axis=1)
        na, nb_ = last_n[a_idx[has], o_idx[has], 0], last_n[j[has], ob[has], 0]
        both = (np.linalg.norm(na, axis=-1) > 1e-5) & (np.linalg.norm(nb_, axis=-1) > 1e-5)
        # the float32 line-line solve is as accurate as the two directions are far from parallel: compare where the sine of
        # their angle exceeds 0.3 (the reference's procedure is not symmetric in its two arguments, only its result is)
        da = goal_all[a_idx[has]] - init[a_idx[has], M - 1, 5]
        db = goal_all[j[has]] - init[j[has], M - 1, 5]
        sine = np.linalg.norm(np.cross(da, db), axis=-1) / (np.linalg.norm(da, axis=-1) * np.linalg.norm(db, axis=-1))
        sel = both & (sine > 0.3)
        err = np.abs(na[sel] + nb_[sel]).max(axis=-1)
        # closestPointsBetweenLineSegments clamps and re-projects once (include/geometry.hpp:228-256): when both parameters
        # leave [0, 1] the pair it returns depends on the order of its arguments, so a few per cent of the pairs are mirrored
        # only approximately -- in the reference as well
        assert sel.sum() > 1000 and (err <= 2e-5).mean() >= 0.95 and err.max() <= 0.2, ((err <= 2e-5).mean(), err.max())


@pytest.mark.gpu
def test_safety_ratio_at_4096_agents_against_kdtree(api):
    from scipy.spatial import cKDTree

    N, M, dim, n_obs = 4096, 5, 3, 20
    torch, sw, init, nbr, sol = _setup(api, N, M, dim, n_obs, 2)
    dev = torch.device("cuda", 0)
    x_all = np.ascontiguousarray(init.transpose(0, 3, 1, 2).reshape(N, -1))
    hdr = np.zeros(N, api.HEADER_DTYPE)
    hdr["vmax"], hdr["amax"] = 1.0, 2.0
    d_out = torch.zeros(N * api.SAFETY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    sol.safety_metrics_device(N, 0, N, 1, 0.1, torch.from_numpy(x_all).to(dev), torch.full((N,), sw.radius, dtype=torch.float64, device=dev),
                              torch.full((N,), sw.downwash, dtype=torch.float64, device=dev),
                              torch.from_numpy(hdr.view(np.uint8).reshape(-1).copy()).to(dev), d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(api.SAFETY_DTYPE)
    # equal radii and downwash: the ellipsoidal distance is Euclidean in the z-scaled frame -> nearest neighbour query
    p0 = np.float32(init[:, 0, 0]).astype(np.float64) * np.array([1, 1, 1 / sw.downwash])
    d, j = cKDTree(p0).query(p0, k=2)
    assert np.abs(got["safety_ratio"] - d[:, 1] / (2 * sw.radius)).max() <= 1e-5
    assert (got["closest_agent"] == j[:, 1]).mean() >= 0.999 and (got["sample"] == 0).all()
    assert got["safety_ratio"].min() >= 1.0  # the synthetic swarm is collision-free


@pytest.mark.gpu
def test_corridors_at_4096_agents(api):
    N, M, dim, n_obs = 4096, 5, 3, 20
    torch, sw, init, nbr, sol = _setup(api, N, M, dim, n_obs, 3)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    wmin, wmax = np.array(sw.world_min, dtype=np.float64), np.array(sw.world_max, dtype=np.float64)
    nb = int(np.prod(wmax - wmin) / 4.0)
    boxes = np.concatenate([rng.uniform(wmin, wmax, (nb, 3)), rng.choice([0.3, 0.5, 0.8], (nb, 3))], axis=1)
    wm = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    occ, near = wm.download()
    starts = np.float32(sw.pos).astype(np.float64)
    P = np.repeat(starts[:, None, :], 3, axis=1)
    d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    sol.construct_sfc_device(wm, api.SFC_INIT, N, torch.from_numpy(P.reshape(-1)).to(dev),
                             torch.full((N,), sw.radius, dtype=torch.float64, device=dev), d_sfc, d_st)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    B = d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(N, M)
    assert set(np.unique(st)) <= {0, 1} and st.sum() > 0.8 * N
    key0, res, reach = wm.key0, 0.1, sw.radius - 1e-4
    nz, ny, nx = occ.shape
    checked = full_clear = 0
    for a in np.nonzero(st == 1)[0]:
        lo, hi = B[a, 0]["bmin"], B[a, 0]["bmax"]
        assert (B[a]["bmin"] == lo).all() and (B[a]["bmax"] == hi).all()
        assert (starts[a] > lo - 1e-5).all() and (starts[a] < hi + 1e-5).all()
        # expandSFC stops at the world boundary; its margin compensation (src/collision_constraints.cpp:868-877) then moves a
        # face that is not ON the boundary outwards by 0.05, which crosses it when the world size is no multiple of the grid
        assert (lo >= wmin - 0.05 - 1e-5).all() and (hi <= wmax + 0.05 + 1e-5).all()
        # isObstacleInSFC measures the L-infinity distance to the cell that is nearest in the EUCLIDEAN sense (:796-803), so
        # it is guaranteed to see every occupied cell that touches a sample point (one cell = 0.1 m of clearance around the
        # grown box, 0.05 m after the margin compensation) and sees the cells up to the full radius unless a Euclidean-nearer
        # cell hides them: the first is asserted for every corridor, the second counted
        for r_, strict in ((0.05 - 1e-4, True), (reach, False)):
            i0 = np.maximum(np.floor((lo - r_) / res + 1e-6).astype(int) - key0, 0)
            i1 = np.minimum(np.ceil((hi + r_) / res - 1e-6).astype(int) - key0, [nx, ny, nz])
            hit = occ[i0[2]:i1[2], i0[1]:i1[1], i0[0]:i1[0]].any()
            if strict:
                assert not hit, a
            else:
                full_clear += not hit
        checked += 1
    # a start the construction rejects does sit inside an inflated obstacle: some occupied cell within radius of its grid cell
    for a in np.nonzero(st == 0)[0][:200]:
        lo = np.floor(starts[a] / res) * res
        hi = np.ceil(starts[a] / res) * res
        i0 = np.maximum(np.floor((lo - sw.radius - res) / res + 1e-6).astype(int) - key0, 0)
        i1 = np.minimum(np.ceil((hi + sw.radius + res) / res - 1e-6).astype(int) - key0, [nx, ny, nz])
        assert occ[i0[2]:i1[2], i0[1]:i1[1], i0[0]:i1[0]].any(), a
    assert checked > 3000 and full_clear >= 0.9 * checked, (checked, full_clear)
    wm.close()

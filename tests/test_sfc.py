"""Safe-flight-corridor construction: the oracle restatement (oracle/lscsfc_oracle.c) against the
face the reference's result log pins and against geometric invariants; the HIP map + construction kernels against the
oracle, bit for bit; and the chain map -> corridor -> goal LP -> QP against the reference log on the GPU."""
import numpy as np
import pytest

from tests import helpers as H


def _forest(oracle):
    g = H.load_golden("forest10_world")
    mp = oracle.Map(g["boxes"], g["world_min"], g["world_max"], g["resolution"], g["max_dist"])
    return g, mp


def _pts(first, second=None, third=None):
    n = len(first)
    P = np.zeros((n, 3, 3))
    P[:, 0] = first
    P[:, 1] = first if second is None else second
    P[:, 2] = first if third is None else third
    return P


def _clear_of_obstacles(occ, key0, res, lo, hi, reach):
    """True if no occupied cell intersects the open box (lo - reach, hi + reach) -- geometry only, no distance map."""
    nz, ny, nx = occ.shape
    idx = []
    for k, n in zip(range(3), (nx, ny, nz)):
        a = int(np.floor((lo[k] - reach) / res + 1e-6)) - key0[k]       # first cell whose interior reaches above lo - reach
        b = int(np.ceil((hi[k] + reach) / res - 1e-6)) - key0[k]        # one past the last cell below hi + reach
        idx.append((max(a, 0), min(b, n)))
    sub = occ[idx[2][0]:idx[2][1], idx[1][0]:idx[1][1], idx[0][0]:idx[0][1]]
    return not sub.any()


def test_forest10_corridors_pin_and_invariants(oracle):
    g, mp = _forest(oracle)
    starts = np.array(g["starts"])
    n, M = len(starts), 10
    sfc = np.zeros((n, M), oracle.BOX_DTYPE)
    st = mp.construct_sfc(oracle.SFC_INIT, _pts(starts), g["radius"], sfc)
    assert (st == 1).all()
    # the face the reference's own run pins (agent 1, -x): rasterised obstacle edge 2.4 + margin 0.15
    a = g["pinned"]["agent"]
    assert abs(sfc[a, 0]["bmin"][0] - g["pinned"]["value"]) <= 1e-6
    occ = mp.occ()
    wmin, wmax = np.array(g["world_min"]), np.array(g["world_max"])
    for q in range(n):
        lo, hi = sfc[q, 0]["bmin"], sfc[q, 0]["bmax"]
        assert (sfc[q]["bmin"] == lo).all() and (sfc[q]["bmax"] == hi).all()  # initializeSFC: all M boxes equal (:381-383)
        assert (starts[q] > lo - 1e-5).all() and (starts[q] < hi + 1e-5).all()
        assert (lo >= wmin - 1e-5).all() and (hi <= wmax + 1e-5).all()
        # an agent (L-infinity radius) whose centre is anywhere in the corridor touches no occupied cell
        assert _clear_of_obstacles(occ, mp.key0, g["resolution"], lo, hi, g["radius"] - 1e-4), q
        # maximal: one more cell in any direction that is not the world boundary runs into an obstacle
        for k in range(3):
            for side in (0, 1):
                if (side == 0 and lo[k] <= wmin[k] + 1e-5) or (side == 1 and hi[k] >= wmax[k] - 1e-5):
                    continue
                lo2, hi2 = lo.copy(), hi.copy()
                if side == 0:
                    lo2[k] -= g["resolution"]
                else:
                    hi2[k] += g["resolution"]
                assert not _clear_of_obstacles(occ, mp.key0, g["resolution"], lo2, hi2, g["radius"] - 1e-4), (q, k, side)


def test_reference_quirk_phantom_cell_at_the_world_origin(oracle):
    """Where no occupied cell lies within max_dist, DynamicEDTOctomap::getDistanceAndClosestObstacle leaves `closest_point` untouched;
    the reference's point3d is default-constructed, so isObstacleInSFC measures against a cell at the WORLD ORIGIN
    (src/collision_constraints.cpp:796-800).  In an EMPTY world the only thing that can stop a corridor short of the world boundary is
    that phantom: every corridor is cut so that none of its sample points comes within margin + res/2 (L-infinity) of the origin."""
    wmin, wmax = np.array([-3.0, -3.0, 0.0]), np.array([3.0, 3.0, 2.5])
    mp = oracle.Map(np.zeros((0, 6)), wmin, wmax, 0.1, 1.0)
    starts = np.float32([[1.5, 1.2, 1.0], [-1.0, 2.0, 0.6], [2.0, -2.0, 2.0], [0.5, 0.4, 0.5]]).astype(np.float64)
    sfc = np.zeros((len(starts), 3), oracle.BOX_DTYPE)
    st = mp.construct_sfc(oracle.SFC_INIT, _pts(starts), 0.15, sfc)
    assert (st == 1).all()
    full = 0
    for q in range(len(starts)):
        lo, hi = sfc[q, 0]["bmin"], sfc[q, 0]["bmax"]
        # the box's sample lattice (box_min + i * res) never comes within margin + res/2 of the origin in all three axes at once
        near = [np.any(np.abs(np.arange(lo[k], hi[k] + 1e-6, 0.1)) < 0.15 + 0.05 + 1e-5) for k in range(3)]
        assert not all(near), (q, lo, hi)
        full += int(np.all(lo <= wmin + 0.15 + 1e-4) and np.all(hi >= wmax - 0.15 - 1e-4))
    assert full == 0  # an empty world would otherwise give every agent the whole world box


def test_map_rasterisation_and_nearest_field(oracle):
    g, mp = _forest(oracle)
    occ, near = mp.occ(), mp.nearest()
    assert tuple(mp.dims) == (101, 101, 26) and tuple(mp.key0) == (-50, -50, 0)
    # updateOctreeFromCSV: cells round((c - s/2)/res) .. round((c + s/2)/res) - 1 per axis
    want = np.zeros_like(occ)
    for b in np.float32(g["boxes"]).astype(np.float64):
        lo = [int(np.floor((b[k] - 0.5 * b[3 + k]) / 0.1 + 0.5)) for k in range(3)]
        hi = [int(np.floor((b[k] + 0.5 * b[3 + k]) / 0.1 + 0.5)) for k in range(3)]
        x0, x1 = max(lo[0] + 50, 0), min(hi[0] + 50, 101)
        y0, y1 = max(lo[1] + 50, 0), min(hi[1] + 50, 101)
        want[max(lo[2], 0):min(hi[2], 26), y0:y1, x0:x1] = 1
    assert np.array_equal(occ, want)
    # nearest field against scipy's exact Euclidean distance transform (distances; the argmin may differ at ties)
    from scipy import ndimage

    dist, ind = ndimage.distance_transform_edt(occ == 0, return_indices=True)
    valid = (near >> 24) & 1
    off = np.stack([((near >> s) & 255) - 128 for s in (16, 8, 0)], axis=0)  # dz, dy, dx
    d_mine = np.sqrt((off.astype(np.float64) ** 2).sum(axis=0))
    assert np.array_equal(valid == 1, dist <= 10.0 + 1e-9)
    assert np.abs(d_mine[valid == 1] - dist[valid == 1]).max() <= 1e-9
    zz, yy, xx = np.nonzero(valid)
    assert occ[zz + off[0][zz, yy, xx], yy + off[1][zz, yy, xx], xx + off[2][zz, yy, xx]].all()  # the code points at an occupied cell


def test_corridor_updates_shift_and_contain(oracle):
    g, mp = _forest(oracle)
    starts, goals = np.array(g["starts"]), np.array(g["goals"])
    n, M = len(starts), 5
    sfc = np.zeros((n, M), oracle.BOX_DTYPE)
    mp.construct_sfc(oracle.SFC_INIT, _pts(starts), g["radius"], sfc)
    step = (goals - starts) / np.linalg.norm(goals - starts, axis=1, keepdims=True)
    for mode in (oracle.SFC_FROM_HULL, oracle.SFC_FROM_POINT):
        cur = sfc.copy()
        # make the boxes of a corridor distinguishable, then update with points 0.3 m / 0.5 m ahead
        for m in range(M):
            cur["bmax"][:, m, 2] -= 0.1 * m
        before = cur.copy()
        last, goal, wp = starts + 0.3 * step, starts + 0.5 * step, starts + 0.5 * step
        last, goal, wp = (np.float32(v).astype(np.float64) for v in (last, goal, wp))
        st = mp.construct_sfc(mode, _pts(last, goal, wp), g["radius"], cur)
        assert np.array_equal(cur[:, :M - 1], before[:, 1:])  # sfcs[m] = sfcs[m + 1] (:400-402, :418-420)
        for q in range(n):
            lo, hi = cur[q, M - 1]["bmin"], cur[q, M - 1]["bmax"]
            if st[q] == 0:
                assert cur[q, M - 1] == before[q, M - 1]  # "use previous one" (:406-409, :430-433)
                continue
            pts = [last[q]] if mode == oracle.SFC_FROM_POINT else [last[q], goal[q]]
            for p in pts:
                assert (p > lo - 1e-5).all() and (p < hi + 1e-5).all(), (mode, q)
        assert st.sum() >= n - 2


# ------------------------------------------------------------------------------------------------------------------
def _random_world(seed, n_boxes=120):
    rng = np.random.default_rng(seed)
    wmin, wmax = np.array([-6.0, -6.0, 0.0]), np.array([6.0, 6.0, 4.0])
    c = rng.uniform(wmin, wmax, (n_boxes, 3))
    s = rng.choice([0.3, 0.5, 0.8, 1.2], (n_boxes, 3))
    return np.concatenate([c, s], axis=1), wmin, wmax


@pytest.mark.gpu
def test_gpu_map_matches_oracle(api, oracle, tmp_path):
    g = H.load_golden("forest10_world")
    worlds = [(np.array(g["boxes"]), g["world_min"], g["world_max"])] + [_random_world(3)]
    for boxes, wmin, wmax in worlds:
        want = oracle.Map(boxes, wmin, wmax, 0.1, 1.0)
        got = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
        assert np.array_equal(got.dims, want.dims) and np.array_equal(got.key0, want.key0)
        occ, near = got.download()
        assert np.array_equal(occ, want.occ())
        assert np.array_equal(near, want.nearest())  # separable passes == brute force, including the tie rule
        # the CSV loader reads what the reference's world files hold
        p = tmp_path / "world.csv"
        np.savetxt(p, boxes, delimiter=",", fmt="%.17g")
        from_csv = api.WorldMap(csv_path=str(p), world_min=wmin, world_max=wmax)
        assert np.array_equal(from_csv.download()[0], occ)
        got.close()
        from_csv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prepared", [0.0, 0.25, 0.15], ids=["plain", "free_space_table", "table_for_smaller_agents"])
@pytest.mark.parametrize("world", ["forest10", "random3d", "sparse_origin"])
@pytest.mark.parametrize("variant", ["latency", "throughput"])
def test_gpu_corridors_match_oracle_bit_for_bit(api, oracle, world, prepared, variant, monkeypatch):
    """`variant`: the two builds of the corridor kernel (1024 threads per agent, one workgroup per CU / 512 threads with capped registers,
    two per CU: csrc/lscsfc_tp.hip), forced through LSCSFC_VARIANT -- a launch normally picks by agents > CUs.
    `prepared` > 0: lscqp_map_prepare's free-space table is in place -- tests of boxes it proves free are passed without sampling
    (for the agents whose radius it covers: 0.15 leaves the 0.25 m agents on the exact path), the corridors must not change by a bit."""
    import torch

    monkeypatch.setenv("LSCSFC_VARIANT", variant)
    rng = np.random.default_rng(11)
    if world == "sparse_origin":  # few obstacles: the phantom cell at the world origin (:796-800) decides many corridors
        wmin, wmax = np.array([-4.0, -4.0, 0.0]), np.array([4.0, 4.0, 2.5])
        boxes = np.array([[2.5, 2.5, 1.0, 0.5, 0.5, 2.0], [-3.0, 1.0, 0.5, 0.4, 0.4, 1.0]])
        starts = rng.uniform([-2.0, -2.0, 0.3], [2.0, 2.0, 2.2], (200, 3))
        M, dim = 5, 3
    elif world == "forest10":
        g = H.load_golden("forest10_world")
        boxes, wmin, wmax = np.array(g["boxes"]), np.array(g["world_min"]), np.array(g["world_max"])
        starts = np.concatenate([np.array(g["starts"]), np.c_[rng.uniform(-4.8, 4.8, (150, 2)), np.full(150, 0.6)]])
        M, dim = 10, 2
    else:
        boxes, wmin, wmax = _random_world(5)
        starts = rng.uniform(wmin + 0.2, wmax - 0.2, (300, 3))
        M, dim = 5, 3
    starts = np.float32(starts).astype(np.float64)
    n = len(starts)
    radius = np.where(np.arange(n) % 7 == 0, 0.25, 0.15)
    omap = oracle.Map(boxes, wmin, wmax, 0.1, 1.0)
    gmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    if prepared > 0:
        gmap.prepare(prepared)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    def gpu(mode, pts, sfc):
        d_sfc = torch.from_numpy(sfc.view(np.float64).reshape(-1).copy()).to(dev)
        d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
        sol.construct_sfc_device(gmap, mode, n, up(pts.reshape(-1)), up(radius), d_sfc, d_st)
        torch.cuda.synchronize()
        return d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(n, M), d_st.cpu().numpy()

    want = np.zeros((n, M), oracle.BOX_DTYPE)
    st_w = omap.construct_sfc(oracle.SFC_INIT, _pts(starts), radius, want)
    got, st_g = gpu(api.SFC_INIT, _pts(starts), np.zeros((n, M), api.BOX_DTYPE))
    assert np.array_equal(st_g, st_w) and 0 < st_w.sum()
    ok = st_w == 1
    assert np.array_equal(got["bmin"][ok], want["bmin"][ok]) and np.array_equal(got["bmax"][ok], want["bmax"][ok])
    assert (got["bmin"][~ok] == 0).all()  # failures (start inside an inflated obstacle) leave the boxes untouched
    # replan updates from the valid corridors, both generators
    idx = np.nonzero(ok)[0]
    base = want.copy()
    base[~ok] = base[idx[0]]
    direction = rng.normal(size=(n, 3))
    if dim == 2:
        direction[:, 2] = 0
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    last = np.float32(starts + 0.25 * direction).astype(np.float64)
    goal = np.float32(starts + 0.6 * direction).astype(np.float64)
    wp = np.float32(starts + 0.5 * direction).astype(np.float64)
    for mode_o, mode_g in ((oracle.SFC_FROM_HULL, api.SFC_FROM_HULL), (oracle.SFC_FROM_POINT, api.SFC_FROM_POINT)):
        w = base.copy()
        st_w = omap.construct_sfc(mode_o, _pts(last, goal, wp), radius, w)
        got, st_g = gpu(mode_g, _pts(last, goal, wp), base.copy())
        assert np.array_equal(st_g, st_w) and 0 < st_w.sum() < n + 1
        assert np.array_equal(got["bmin"], w["bmin"]) and np.array_equal(got["bmax"], w["bmax"]), mode_o
    gmap.close()


@pytest.mark.gpu
def test_map_prepare_is_idempotent_and_rebuilds_for_larger_agents(api):
    """lscqp_map_prepare: a table built for a radius serves every smaller one (a second call is a no-op), a larger radius rebuilds it;
    bad arguments are refused; the corridors of a 3-D world with few obstacles -- where almost every test is passed by the table --
    are the ones the plain map gives, after every call."""
    import torch

    rng = np.random.default_rng(4)
    wmin, wmax = np.array([-6.0, -6.0, 0.0]), np.array([6.0, 6.0, 5.0])
    boxes = np.array([[2.0, 1.0, 2.0, 0.6, 0.6, 0.6], [-2.5, -1.5, 3.0, 0.5, 0.8, 0.5], [0.5, -3.0, 1.0, 1.0, 0.4, 2.0]])
    n, M = 96, 5
    starts = np.float32(rng.uniform(wmin + 0.5, wmax - 0.5, (n, 3))).astype(np.float64)
    radius = np.where(np.arange(n) % 3 == 0, 0.25, 0.15)
    sol = api.Solver(api.make_desc(M=M, dim=3, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    pts = torch.from_numpy(np.repeat(starts[:, None, :], 3, axis=1).reshape(-1).copy()).to(dev)
    d_r = torch.from_numpy(radius).to(dev)

    def corridors(m):
        d_sfc = torch.zeros(n * M * 6, dtype=torch.float64, device=dev)
        d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
        sol.construct_sfc_device(m, api.SFC_INIT, n, pts, d_r, d_sfc, d_st)
        torch.cuda.synchronize()
        return d_sfc.cpu().numpy(), d_st.cpu().numpy()

    gmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    want = corridors(gmap)
    assert want[1].sum() > n // 2
    for r in (0.15, 0.1, 0.25, 0.25):  # build, no-op, rebuild, no-op
        gmap.prepare(r)
        got = corridors(gmap)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), r
    with pytest.raises(api.LscqpError):
        gmap.prepare(0.0)
    gmap.close()


@pytest.mark.gpu
def test_free_space_table_next_to_walls_floor_and_origin(api):
    """Corridors that lie along the world's boundary have a sample OUTSIDE the distance map on every column (the face sits on the
    boundary, its float coordinate a hair beyond it) -- measured against the phantom cell at the world origin.  The table-driven tests
    treat such samples as harmless where one coordinate of the sample, or a whole axis of the box, is far from 0, and leave the rest to
    the exact path: agents next to each wall, on the floor, under the ceiling, and around the world origin (which is ON the floor of
    this room) get, bit for bit, the corridors of the plain map -- first replan and hull update."""
    import torch

    wmin, wmax = np.array([-6.0, -6.0, 0.0]), np.array([6.0, 6.0, 4.0])
    boxes = np.array([[2.0, 1.5, 2.0, 0.6, 0.6, 0.6], [-3.0, -2.0, 1.0, 0.5, 0.5, 2.0], [0.8, -0.6, 0.4, 0.4, 0.4, 0.8]])
    spots = [[-5.8, 0.5, 2.0], [5.8, -1.0, 2.0], [0.5, -5.8, 1.5], [1.0, 5.8, 3.0], [3.0, 3.0, 0.2], [-4.0, 2.0, 3.8], [0.3, 0.2, 0.2], [-0.4, 0.3, 0.3],
             [0.0, 0.0, 1.5], [5.7, 5.7, 0.2], [-5.7, -5.7, 3.8], [0.2, -5.7, 0.2], [-5.8, 0.1, 0.2], [2.5, -3.0, 2.0]]
    starts = np.float32(np.array(spots)).astype(np.float64)
    n, M = len(starts), 5
    radius = np.full(n, 0.15)
    sol = api.Solver(api.make_desc(M=M, dim=3, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(radius).to(dev)
    step = np.array([0.3, -0.2, 0.1])
    inward = -np.sign(starts) * np.abs(step)
    inward[:, 2] = np.where(starts[:, 2] > 2.0, -0.1, 0.1)
    P1 = np.repeat(starts[:, None, :], 3, axis=1)
    P2 = np.float32(np.stack([starts + 0.5 * inward, starts + inward, starts + inward], axis=1)).astype(np.float64)

    def run(m):
        out = []
        d_sfc = torch.zeros(n * M * 6, dtype=torch.float64, device=dev)
        d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
        for mode, P in ((api.SFC_INIT, P1), (api.SFC_FROM_HULL, P2)):
            sol.construct_sfc_device(m, mode, n, torch.from_numpy(P.reshape(-1).copy()).to(dev), d_r, d_sfc, d_st)
            torch.cuda.synchronize()
            out.append((d_sfc.cpu().numpy().copy(), d_st.cpu().numpy().copy()))
        return out

    gmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    want = run(gmap)
    assert want[0][1].sum() >= n - 2 and want[1][1].sum() >= n // 2
    box0 = want[0][0].reshape(n, M, 6)[:, 0]
    assert (box0[:, 0] <= wmin[0] + 1e-3).any() and (box0[:, 3] >= wmax[0] - 1e-3).any() and (box0[:, 2] <= 1e-3).any() and (box0[:, 5] >= wmax[2] - 1e-3).any()
    gmap.prepare(0.15)
    got = run(gmap)
    for (a, sa), (b, sb) in zip(got, want):
        assert np.array_equal(sa, sb) and np.array_equal(a, b)
    gmap.close()


@pytest.mark.gpu
def test_gpu_chain_world_to_trajectory_reproduces_reference_log(api, oracle):
    """forest10, agent 1, first replan, everything on the device: world boxes -> voxel map -> initializeSFC -> goal LP ->
    trajectory QP.  The reference's own result log (tests/golden/kat_log.json) is the expected output."""
    import torch

    g = H.load_golden("forest10_world")
    kat = H.load_golden("kat_log")
    p = kat["params"]
    case = [c for c in kat["cases"] if c["agent"] == 1][0]
    M = p["M"]
    dev = torch.device("cuda", 0)
    gmap = api.WorldMap(g["boxes"], g["world_min"], g["world_max"], g["resolution"], g["max_dist"])
    sol = api.Solver(H.abi_desc(api, p, use_sfc=True))
    cls = H.oracle_class(oracle, p, use_sfc=True)
    start = np.array(g["starts"][1])
    d_sfc = torch.zeros(M * 6, dtype=torch.float64, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    sol.construct_sfc_device(gmap, api.SFC_INIT, 1, torch.from_numpy(_pts(start[None]).reshape(-1)).to(dev),
                             torch.tensor([g["radius"]], dtype=torch.float64, device=dev), d_sfc, d_st)
    torch.cuda.synchronize()
    assert d_st.item() == 1
    hdr = np.zeros(1, api.HEADER_DTYPE)
    hdr["p0"][0] = case["p0"]
    hdr["goal"][0] = [3.0, 2.5, 0.6]            # current goal point before the goal LP (one grid cell behind the waypoint)
    hdr["next_waypoint"][0] = case["next_waypoint"]
    hdr["vmax"][0], hdr["amax"][0] = p["vmax"], p["amax"]
    hdr["radius"][0], hdr["nominal_velocity"][0] = p["radius"], p["nominal_velocity"]
    d_hdr = torch.from_numpy(hdr.view(np.uint8).reshape(-1).copy()).to(dev)
    d_off = torch.zeros(2, dtype=torch.int64, device=dev)
    d_rows = torch.zeros(4, dtype=torch.float64, device=dev)
    d_gst = torch.full((1,), -1, dtype=torch.int32, device=dev)
    sol.optimize_goal_device(1, d_hdr, d_rows, d_off, d_sfc, d_gst)
    d_x = torch.zeros(sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(1, dtype=torch.float64, device=dev)
    d_qst = torch.full((1,), -1, dtype=torch.int32, device=dev)
    sol.solve_device(1, 0, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_qst)
    torch.cuda.synchronize()
    assert d_gst.item() == 0 and d_qst.item() == 0
    goal = d_hdr.cpu().numpy().view(api.HEADER_DTYPE)["goal"][0]
    assert abs(goal[0] - 2.55) <= 1e-6  # the SFC face bounds the goal LP
    x = d_x.cpu().numpy()
    for st in kat["agents"][1]["states"][1:]:
        pos, vel, acc = oracle.state_at(cls, x, st["t"])
        assert np.allclose(pos, st["p"][:2], rtol=0, atol=2e-5)
        assert np.allclose(vel, st["v"][:2], rtol=1e-4, atol=2e-6)
        assert np.allclose(acc, st["a"][:2], rtol=3e-4, atol=2e-5)
    gmap.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["latency", "throughput"])
@pytest.mark.parametrize("world", ["closet", "empty_room", "one_pillar", "long_hall"])
def test_gpu_corridor_look_ahead_special_paths(api, oracle, world, variant, monkeypatch):
    """The look-ahead of the corridor kernel continues behind tests the world boundary fails (a list of segments, one inner loop of the
    reference each) and ends a batch only at an obstacle or at its limits.  Worlds chosen for its corner cases, bit for bit against the
    oracle, with and without the free-space table:
      closet      1.2 x 1.0 x 0.8 m, no obstacle: all six directions fail on the boundary inside ONE batch (the expansion ends in it);
      empty_room  6 x 5 x 3 m, no obstacle: > 126 tests, every failure a boundary failure, face coordinates crossing 0 and binades;
      one_pillar  the same room with one box: obstacle failures between boundary failures;
      long_hall   110 m at 0.1 m = 1100 cells along x: beyond the cell-centre table, every test is decided on its own (sequentially)."""
    import torch

    monkeypatch.setenv("LSCSFC_VARIANT", variant)
    rng = np.random.default_rng(23)
    none = np.zeros((0, 6))
    if world == "closet":
        wmin, wmax, boxes, n = np.array([-0.6, -0.35, 0.0]), np.array([0.6, 0.65, 0.8]), none, 12
    elif world == "empty_room":
        wmin, wmax, boxes, n = np.array([-3.0, -2.2, 0.0]), np.array([3.0, 2.8, 3.0]), none, 24
    elif world == "one_pillar":
        wmin, wmax, boxes, n = np.array([-3.0, -2.2, 0.0]), np.array([3.0, 2.8, 3.0]), np.array([[0.6, 0.4, 1.5, 0.5, 0.5, 3.0]]), 32
    else:
        wmin, wmax, boxes, n = np.array([-55.0, -1.0, 0.0]), np.array([55.0, 1.0, 1.0]), np.array([[10.0, 0.5, 0.5, 0.4, 0.4, 1.0], [-20.0, -0.5, 0.5, 0.4, 0.4, 1.0]]), 6
    M = 5
    starts = np.float32(rng.uniform(wmin + 0.25, wmax - 0.25, (n, 3))).astype(np.float64)
    radius = np.full(n, 0.15)
    omap = oracle.Map(boxes, wmin, wmax, 0.1, 1.0)
    want = np.zeros((n, M), oracle.BOX_DTYPE)
    st_w = omap.construct_sfc(oracle.SFC_INIT, _pts(starts), radius, want)
    assert st_w.sum() >= n // 2
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    P2 = _pts(np.float32(starts + 0.1 * direction).astype(np.float64), np.float32(starts + 0.3 * direction).astype(np.float64),
              np.float32(starts + 0.2 * direction).astype(np.float64))
    base = want.copy()
    base[st_w != 1] = base[np.nonzero(st_w == 1)[0][0]]
    want2 = base.copy()
    st_w2 = omap.construct_sfc(oracle.SFC_FROM_HULL, P2, radius, want2)
    sol = api.Solver(api.make_desc(M=M, dim=3, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    d_r = torch.from_numpy(radius).to(dev)
    for prepared in (0.0, 0.15):
        gmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
        if prepared > 0:
            gmap.prepare(prepared)
        for mode, P, start, exp, exp_st in ((api.SFC_INIT, _pts(starts), np.zeros((n, M), api.BOX_DTYPE), want, st_w), (api.SFC_FROM_HULL, P2, base, want2, st_w2)):
            d_sfc = torch.from_numpy(start.view(np.float64).reshape(-1).copy()).to(dev)
            d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
            sol.construct_sfc_device(gmap, mode, n, torch.from_numpy(np.ascontiguousarray(P).reshape(-1).copy()).to(dev), d_r, d_sfc, d_st)
            torch.cuda.synchronize()
            got, st_g = d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(n, M), d_st.cpu().numpy()
            assert np.array_equal(st_g, exp_st), (world, prepared, mode)
            ok = exp_st == 1
            sel = ok if mode == api.SFC_INIT else np.ones(n, bool)
            assert np.array_equal(got["bmin"][sel], exp["bmin"][sel]) and np.array_equal(got["bmax"][sel], exp["bmax"][sel]), (world, prepared, mode)
        gmap.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["latency", "throughput"])
def test_gpu_corridor_seeded_outside_the_world(api, oracle, variant, monkeypatch):
    """A replan whose last point / goal lie OUTSIDE the world (a plan that left it): the hull box, clipped to the previous corridor, is
    INVERTED along that axis (upper face below the world's lower bound).  The reference grows such a box like any other -- no sample
    points, so every test passes until a boundary says no -- and the layer of the direction that grows back INTO the world fails the
    boundary on its inner face, which the look-ahead's list of boundary failures does not predict: the batch has to end in front of
    the disagreement (the safety net of expand_sfc).  Found by tools/sweep_corridors.py (seed 3, agent 197); bit for bit against the oracle."""
    import torch

    monkeypatch.setenv("LSCSFC_VARIANT", variant)
    wmin, wmax = np.array([-6.27, -6.0, 0.0]), np.array([6.0, 6.0, 3.31])
    boxes = np.array([[0.6, 0.4, 1.5, 0.5, 0.5, 3.0], [3.0, -4.0, 1.0, 0.8, 0.8, 0.8]])
    rng = np.random.default_rng(5)
    n, M = 24, 5
    starts = np.float32(np.c_[rng.uniform(-5, 5, n), rng.uniform(-5.9, -5.6, n), rng.uniform(0.5, 2.8, n)]).astype(np.float64)  # along the y = -6 wall
    out = np.arange(n) % 3 != 2  # two of three agents have left the world through that wall, the others along x / z
    radius = np.where(np.arange(n) % 2 == 0, 0.25, 0.15)
    omap = oracle.Map(boxes, wmin, wmax, 0.1, 1.0)
    base = np.zeros((n, M), oracle.BOX_DTYPE)
    st0 = omap.construct_sfc(oracle.SFC_INIT, _pts(starts), radius, base)
    assert (st0 == 1).all()
    last, goal = starts.copy(), starts.copy()
    last[:, 1] = np.where(out, wmin[1] - rng.uniform(0.05, 0.3, n), starts[:, 1] + 0.1)
    goal[:, 1] = np.where(out, wmin[1] - rng.uniform(0.4, 0.8, n), starts[:, 1] + 0.3)
    last[~out, 0] += 0.2
    last, goal = np.float32(last).astype(np.float64), np.float32(goal).astype(np.float64)
    P = _pts(last, goal, last)
    sol = api.Solver(api.make_desc(M=M, dim=3, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    for prepared in (0.0, 0.25):
        gmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
        if prepared > 0:
            gmap.prepare(prepared)
        for mode_o, mode_g in ((oracle.SFC_FROM_POINT, api.SFC_FROM_POINT), (oracle.SFC_FROM_HULL, api.SFC_FROM_HULL)):
            want = base.copy()
            st_w = omap.construct_sfc(mode_o, P, radius, want)
            d_sfc = torch.from_numpy(base.view(np.float64).reshape(-1).copy()).to(dev)
            d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
            sol.construct_sfc_device(gmap, mode_g, n, torch.from_numpy(np.ascontiguousarray(P).reshape(-1).copy()).to(dev), torch.from_numpy(radius).to(dev), d_sfc, d_st)
            torch.cuda.synchronize()
            got = d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(n, M)
            assert np.array_equal(d_st.cpu().numpy(), st_w), (prepared, mode_o)
            assert np.array_equal(got["bmin"], want["bmin"]) and np.array_equal(got["bmax"], want["bmax"]), (prepared, mode_o)
            inverted = (want["bmax"][:, M - 1] < want["bmin"][:, M - 1]).any(axis=1)
            if mode_o == oracle.SFC_FROM_POINT and prepared == 0.0:
                assert inverted.any(), "the premise: some corridor of this test is an inverted box"
        gmap.close()


@pytest.mark.gpu
def test_corridor_work_order_changes_when_a_corridor_is_built_never_the_corridor(api, torch_cuda):
    """Round 4: lscqp_construct_sfc_device_ordered builds agent d_order[k]'s corridor in workgroup k and records every agent's cost;
    lscqp_order_by_cost_device sorts by those costs, most expensive first.  1500 agents in a room with pillars (more than the chip takes
    at once, so the throughput build runs): INIT and FROM_HULL give bit for bit the same boxes and statuses as given, in the sorted order and
    in a random order; the recorded costs are positive and the order is a permutation that is non-increasing in the 16 cost levels, ties in index order."""
    torch = torch_cuda
    rng = np.random.default_rng(9)
    N, M = 1500, 5
    wmin, wmax = np.array([-12.0, -12.0, 0.0]), np.array([12.0, 12.0, 6.0])
    boxes = np.concatenate([rng.uniform(wmin + 1, wmax - 1, (160, 3)), rng.choice([0.4, 0.6, 1.0], (160, 3))], axis=1)
    wm = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    wm.prepare(0.15)
    sol = api.Solver(api.make_desc(M=M, dim=3, world_min=wmin, world_max=wmax))
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    starts = np.float32(rng.uniform(wmin + 0.5, wmax - 0.5, (N, 3))).astype(np.float64)
    P = np.repeat(starts[:, None, :], 3, axis=1)
    rad = np.full(N, 0.15)
    d_sfc = torch.zeros(N * M * 6, dtype=torch.float64, device=dev)
    d_st = torch.zeros(N, dtype=torch.int32, device=dev)
    d_cost = torch.zeros(N, dtype=torch.int32, device=dev)  # (uint32 on the device)
    sol.construct_sfc_device(wm, api.SFC_INIT, N, up(P.reshape(-1)), up(rad), d_sfc, d_st, d_cost=d_cost)
    torch.cuda.synchronize()
    base, st0 = d_sfc.clone(), d_st.clone()
    feasible = st0.cpu().numpy() == 1
    assert feasible.sum() > N // 2
    cost = d_cost.cpu().numpy().view(np.uint32)
    assert (cost > 0).all()
    d_order = torch.zeros(N, dtype=torch.int32, device=dev)
    sol.order_by_cost_device(N, d_cost, d_order)
    torch.cuda.synchronize()
    order = d_order.cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(N))
    bins = (cost[order].astype(np.uint64) * 15) // max(int(cost.max()), 1)
    assert (np.diff(bins.astype(np.int64)) <= 0).all() and bins[0] == 15
    for lv in np.unique(bins):  # stable: ties keep their index order
        assert (np.diff(order[bins == lv]) > 0).all()
    step = rng.normal(size=(N, 3))
    step /= np.linalg.norm(step, axis=1, keepdims=True)
    P2 = np.float32(np.stack([starts + 0.3 * step, starts + 0.5 * step, starts + 0.5 * step], axis=1)).astype(np.float64)
    ref = {}
    for mode, pts in ((api.SFC_INIT, P), (api.SFC_FROM_HULL, P2)):
        for name, od in (("as given", None), ("sorted", d_order), ("random", up(rng.permutation(N).astype(np.int32)))):
            work, st = base.clone(), torch.zeros(N, dtype=torch.int32, device=dev)
            sol.construct_sfc_device(wm, mode, N, up(pts.reshape(-1)), up(rad), work, st, d_order=od)
            torch.cuda.synchronize()
            if name == "as given":
                ref[mode] = (work, st)
            else:
                assert torch.equal(work, ref[mode][0]) and torch.equal(st, ref[mode][1]), (mode, name)
    wm.close()

"""The dual active-set phase (csrc/lscqp_das.hip, round 5) in front of the interior-point kernel.

CPU tests: the per-class tables (C = T (T'Hx T)^-1 T' per number of terminal segments) against an independent numpy construction from
the reference's Q_base (src/traj_optimizer.cpp:163-178) and continuity rows (:180-214, 318-368, 502-511).
GPU tests (-m gpu): the phase ALONE against the CPU oracle (which is an interior-point method: an algorithmically independent checker),
its hand-over to the interior-point kernel when the budget runs out, the statuses it must leave to that kernel (infeasible, capacity),
and bitwise repeatability.  Same bars as tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

from tests import helpers as H

OBJ_TOL = 1e-8
X_TOL = H.PathTol()  # 1e-8 m for the dual active-set phase, 1e-6 m for the interior-point kernel (tests/helpers.py)
KKT_TOL = 1e-8

Q_INT = np.array([[720, -1800, 1200, 0, 0, -120], [-1800, 4800, -3600, 0, 600, 0], [1200, -3600, 3600, -1200, 0, 0],
                  [0, 0, -1200, 3600, -3600, 1200], [0, 600, 0, -3600, 4800, -1800], [-120, 0, 0, 1200, -1800, 720]], dtype=float)
TB = np.array([[0.0, 0.0, 1.0], [0.0, -1.0, 2.0], [1.0, -4.0, 4.0]])


def _null_space_map(M, end_stop):
    """c = cfix + T z: the equality rows of the model (initial state, C0/C1/C2 junctions, end stop) eliminated."""
    nza = 3 * (M - 1) + (1 if end_stop else 3)
    T = np.zeros((6 * M, nza))
    for m in range(M):
        last = end_stop and m == M - 1
        for j in range(3):
            T[6 * m + 3 + j, 3 * m + (0 if last else j)] = 1.0
        if m >= 1:
            T[6 * m:6 * m + 3, 3 * (m - 1):3 * (m - 1) + 3] = TB
    return T


@pytest.mark.parametrize("M,es,dt,w_c,w_t", [(5, 1, 0.2, 0.01, 1.0), (10, 1, 0.2, 0.01, 1.0), (6, 0, 0.25, 0.02, 3.0), (12, 0, 0.2, 0.01, 1.0), (2, 1, 0.1, 1.0, 0.5)])
def test_tables_equal_an_independent_construction(api, M, es, dt, w_c, w_t):
    L = api.lib()
    L.lscqp_das_build_tables.restype = C.c_size_t
    L.lscqp_das_build_tables.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
    P = 6 * M
    n = L.lscqp_das_build_tables(M, es, dt, w_c, w_t, None)
    assert n == M * (3 + P) * P
    out = np.zeros(n)
    assert L.lscqp_das_build_tables(M, es, dt, w_c, w_t, out.ctypes.data) == n
    T = _null_space_map(M, bool(es))
    Q2 = 2 * w_c * Q_INT * dt ** -5
    for ts in range(1, M + 1):
        Hx = np.kron(np.eye(M), Q2)
        for m in range(M - ts, M):
            Hx[6 * m + 5, 6 * m + 5] += 2 * w_t
        K0 = T.T @ Hx @ T
        tb = out[(ts - 1) * (3 + P) * P:ts * (3 + P) * P]
        Cm = tb[3 * P:].reshape(P, P)
        assert np.array_equal(Cm, Cm.T) and not Cm[:3].any()  # symmetric; the initial state is fixed: its control points never move
        # the defining property, which does not need an inverse on the checker's side: T' Hx C = T'  (C restricted to the plan's degrees of freedom inverts Hx)
        resid = np.abs(T.T @ Hx @ Cm - T.T).max()
        assert resid <= 2e-9 * np.abs(K0).max() * np.abs(Cm).max(), (ts, resid)
        # ... and C moves control points only inside the feasible subspace of the equality rows: C = T X for some X
        coef = np.linalg.lstsq(T, Cm, rcond=None)[0]
        assert np.abs(T @ coef - Cm).max() <= 1e-12 * np.abs(Cm).max()
        e1 = np.zeros(P); e1[:6] = Q2[:, 1]
        e2 = np.zeros(P); e2[:6] = Q2[:, 2]
        assert np.allclose(tb[:P], Cm @ e1, rtol=0, atol=1e-12 * np.abs(Cm @ e1).max())
        assert np.allclose(tb[P:2 * P], Cm @ e2, rtol=0, atol=1e-12 * np.abs(Cm @ e2).max())
        assert np.allclose(tb[2 * P:3 * P], sum(Cm[:, 6 * m + 5] for m in range(M - ts, M)), rtol=0, atol=1e-13 * np.abs(Cm).max())


@pytest.mark.parametrize("M,dim,comm", [(5, 3, 1), (10, 2, 1), (6, 3, 0), (12, 3, 1), (2, 2, 1)])
def test_two_sided_rows_of_a_class_equal_an_independent_construction(api, M, dim, comm):
    """lscqp_das_build_pairs: what the kernel's two-sided rows are as far as they are the CLASS's (stencil type and entries; family, axis,
    segment for the bounds) -- built here from the reference's own loops (src/traj_optimizer.cpp: control-point intervals :252-265 /
    :372-397, velocity rows :448-453, acceleration rows :462-471, communication pairs :482-487), row by row."""
    L = api.lib()
    L.lscqp_das_build_pairs.restype = C.c_size_t
    L.lscqp_das_build_pairs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    P = 6 * M
    n = L.lscqp_das_build_pairs(M, dim, comm, None)
    want = []
    for k in range(dim):  # one interval per control point and axis; the initial state's three control points carry no row
        for cp in range(P):
            want.append((1 if cp >= 3 else 0, k * P + cp, 0, 0, k, cp // 6, int(cp % 6 == 5)))
    for k in range(dim):  # velocity: c[i+1] - c[i], i = 0..4 of every segment; the first two of segment 0 are fixed by the initial state
        for m in range(M):
            for i in range(5):
                want.append((0 if (m == 0 and i < 2) else 2, k * P + 6 * m + i, 0, 1, k, m, 0))
    for k in range(dim):  # acceleration: c[i+2] - 2 c[i+1] + c[i], i = 0..3
        for m in range(M):
            for i in range(4):
                want.append((0 if (m == 0 and i < 1) else 3, k * P + 6 * m + i, 0, 2, k, m, 0))
    for k in range(dim):  # communication range between the last control point of segment uu and the first of segment up + 1 (up < uu)
        for uu in range(1, M):
            for up in range(uu):
                want.append((4 if comm else 0, k * P + 6 * (up + 1), k * P + 6 * uu + 5, 3, k, 0, 0))
    assert n == len(want)
    out = np.zeros(2 * n, np.int32)
    assert L.lscqp_das_build_pairs(M, dim, comm, out.ctypes.data) == n
    for r, (typ, e0, e1, fam, k, m, last) in enumerate(want):
        w0, w1 = int(out[2 * r]), int(out[2 * r + 1])
        assert (w0 >> 24, (w0 >> 12) & 0xfff, w0 & 0xfff) == (typ, e0, e1), (r, w0)
        assert (w1 & 3, (w1 >> 2) & 3, (w1 >> 4) & 15, (w1 >> 8) & 1, (w1 >> 9) & 1) == (fam, k, m, last, int(fam == 0 and k == 2 and m == 0)), (r, w1)


def test_active_set_field_is_validated(api):
    with pytest.raises(api.LscqpError) as e:
        api.Solver(api.make_desc(M=5, dim=3, active_set=7))
    assert e.value.code == api.ERR_INVALID_ARGUMENT


def _oracle_batch(oracle, sw, b, cls):
    ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, b)
    return ag, oracle.solve_batch(cls, ag, lsc, off, sfc, threads=8)


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,n_obs,style,seed", [(64, 5, 3, 20, "forest", 1000), (10, 10, 2, 9, "forest", 3020), (96, 6, 3, 20, "maze", 3518),
                                                      (32, 10, 3, 40, "forest", 3138), (24, 7, 3, 12, "maze", 6), (16, 5, 2, 12, "forest", 11)])
def test_the_phase_alone_against_the_oracle(api, oracle, torch_cuda, solver_path, N, M, dim, n_obs, style, seed):
    """LSCQP_ACTIVE_SET_ONLY on replanning swarms of the bench's shapes: what the phase returns OPTIMAL is the oracle's optimum (x, objective,
    KKT residuals of the reference's row-for-row model), carries LSCQP_INFO_ACTIVE_SET, and reports residuals inside the bars."""
    if solver_path != "active_set":
        pytest.skip("the phase alone: one path")
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    only = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    both = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    solved = total = 0
    for step in range(4):
        b = sw.build()
        ag, R = _oracle_batch(oracle, sw, b, cls)
        hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
        hdr["terminal_segments"] = [oracle.terminal_segments(cls, ag[q:q + 1]) for q in range(N)]
        G = only.solve_host(hdr, rows, roff, sfcp)
        ok = G["status"] == 0
        assert ((G["status"] == 0) | (G["status"] == api.STATUS_ITER_LIMIT)).all()
        assert (R["status"] == 0).all()
        assert ((G["info"]["flags"][ok] & api.INFO_ACTIVE_SET) != 0).all() and (G["info"]["flags"][~ok] == 0).all()
        assert np.abs(G["x"][ok] - R["x"][ok]).max() <= X_TOL
        assert (np.abs(G["obj"][ok] - R["obj"][ok]) / np.maximum(1.0, np.abs(R["obj"][ok]))).max() <= OBJ_TOL
        assert G["info"]["res_primal"][ok].max() <= 1e-9 and G["info"]["res_dual"][ok].max() <= 1e-9 and (G["info"]["gap"][ok] == 0).all()
        for q in np.where(ok)[0][::max(1, N // 6)]:
            lq = np.ascontiguousarray(b["lsc"][q]); sq = np.ascontiguousarray(b["sfc"][q])
            stat, eqv, iqv = H.kkt_from_primal(oracle, cls, ag[q:q + 1], lq, sq, G["x"][q])
            assert stat <= KKT_TOL and eqv <= KKT_TOL and iqv <= KKT_TOL, (step, q, stat, eqv, iqv)
        solved += ok.sum(); total += N
        # the product's default path (phase + interior point behind it) returns the same plans for what the phase solved, bit for bit
        D = both.solve_host(hdr, rows, roff, sfcp)
        assert (D["status"] == 0).all() and np.array_equal(D["x"][ok], G["x"][ok]) and np.array_equal(D["obj"][ok], G["obj"][ok])
        sw.advance(D["x"])
    assert solved >= 0.9 * total, (solved, total)  # (every workload of BASELINE.json: all of them, see profiles/r05_*)


@pytest.mark.gpu
def test_budget_exhausted_instances_are_solved_by_the_interior_point_kernel(api, oracle, torch_cuda, solver_path, monkeypatch):
    """The bench's headline batch (64 x M5 x 20 after three replans: three of its QPs hold one active row) with a budget of ZERO steps: the
    phase finishes what is unconstrained and leaves the rest, those instances come back from the interior-point kernel behind it (no
    LSCQP_INFO_ACTIVE_SET, no LSCQP_INFO_REPAIRED: nothing was repaired), everything agrees with the oracle, and in LSCQP_ACTIVE_SET_ONLY
    the same instances are returned LSCQP_STATUS_ITER_LIMIT."""
    if solver_path != "active_set":
        pytest.skip("hand-over: one path")
    from lsc_dr_planner_amd import synth

    N, M, dim = 64, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=20, seed=1000, style="forest")
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    only = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    for _ in range(3):
        b = sw.build()
        hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
        sw.advance(sol.solve_host(hdr, rows, roff, sfcp, x_init=api.x_init_from_swarm(b, dim))["x"])
    b = sw.build()
    ag, R = _oracle_batch(oracle, sw, b, cls)
    hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
    x0 = api.x_init_from_swarm(b, dim)
    full = sol.solve_host(hdr, rows, roff, sfcp, x_init=x0)
    assert (full["status"] == 0).all() and ((full["info"]["flags"] & api.INFO_ACTIVE_SET) != 0).all() and full["info"]["iterations"].max() >= 1
    sol.set_knob("das_steps", 0)  # (library-internal switch of the handle: the launch's step budget)
    only.set_knob("das_steps", 0)
    G = sol.solve_host(hdr, rows, roff, sfcp, x_init=x0)
    O1 = only.solve_host(hdr, rows, roff, sfcp, x_init=x0)
    by_phase = (G["info"]["flags"] & api.INFO_ACTIVE_SET) != 0
    assert (G["status"] == 0).all() and 0 < by_phase.sum() < N
    assert np.array_equal(by_phase, full["info"]["iterations"] == 0)
    assert ((G["info"]["flags"][~by_phase] & api.INFO_REPAIRED) == 0).all() and (G["info"]["iterations"][~by_phase] >= 3).all()
    assert np.array_equal(O1["status"] == 0, by_phase) and (O1["status"][~by_phase] == api.STATUS_ITER_LIMIT).all()
    # (a batch finished by BOTH kernels: each instance is held to the bar of the kernel that returned it)
    tol_q = H.x_tol_by_instance(api, G["info"])
    assert (np.abs(G["x"] - R["x"]).max(axis=1) <= tol_q).all() and (np.abs(G["obj"] - R["obj"]) / np.maximum(1.0, np.abs(R["obj"]))).max() <= OBJ_TOL
    assert (np.abs(G["x"] - full["x"]).max(axis=1) <= tol_q).all()


@pytest.mark.gpu
def test_infeasible_and_refused_instances_keep_the_interior_point_kernels_statuses(api, oracle, torch_cuda, solver_path):
    """The phase never answers INFEASIBLE or CAPACITY itself: an instance whose rows admit no point and one with more obstacles than the
    launch's kernel instance holds get exactly the statuses they get without the phase, and their neighbours in the batch are solved."""
    import torch

    from lsc_dr_planner_amd import synth

    N, M, dim = 8, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=7, seed=3)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
    rows = rows.copy()
    # instance 2: two opposite half-spaces on the last control point that exclude each other
    r2 = rows[roff[2]:roff[3]].reshape(sw.n_obs, M, 6)
    r2["nx"][0, M - 1, 5], r2["ny"][0, M - 1, 5], r2["nz"][0, M - 1, 5], r2["b"][0, M - 1, 5] = 1.0, 0.0, 0.0, hdr["p0"][2][0] + 0.5
    r2["nx"][1, M - 1, 5], r2["ny"][1, M - 1, 5], r2["nz"][1, M - 1, 5], r2["b"][1, M - 1, 5] = -1.0, 0.0, 0.0, -(hdr["p0"][2][0] - 0.5)
    G = sol.solve_host(hdr, rows, roff, sfcp)
    assert G["status"][2] == api.STATUS_INFEASIBLE and (np.delete(G["status"], 2) == 0).all()
    # capacity: a bogus obstacle count on instance 5 through the device entry (as tests/test_mixed_precision.py does)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    hdr2 = hdr.copy()
    hdr2["n_obs"][5] = 25
    rows2 = np.concatenate([rows, np.zeros(25 * M * 6, api.ROW_DTYPE)])
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    sol.solve_device(N, sw.n_obs, up(hdr2), up(rows2), up(roff), up(sfcp), d_x, d_obj, d_st, None)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert st[5] == api.STATUS_CAPACITY and st[2] == api.STATUS_INFEASIBLE and (np.delete(st, [2, 5]) == 0).all()


@pytest.mark.gpu
def test_the_phase_is_repeatable_bit_for_bit(api, oracle, torch_cuda, solver_path):
    if solver_path != "active_set":
        pytest.skip("repeatability of the phase: one path")
    from lsc_dr_planner_amd import synth

    for (N, M, dim, n_obs, style, seed) in [(64, 5, 3, 20, "forest", 1000), (10, 10, 2, 9, "forest", 3020), (300, 6, 3, 20, "maze", 5)]:
        sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
        for step in range(2):
            b = sw.build()
            hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
            runs = [sol.solve_host(hdr, rows, roff, sfcp) for _ in range(3)]
            for r in runs[1:]:
                assert np.array_equal(r["x"], runs[0]["x"]) and np.array_equal(r["obj"], runs[0]["obj"])
                assert np.array_equal(r["info"]["iterations"], runs[0]["info"]["iterations"]) and np.array_equal(r["status"], runs[0]["status"])
            sw.advance(runs[0]["x"])


@pytest.mark.gpu
def test_nan_in_the_inputs_is_never_an_optimal_plan(api, oracle, torch_cuda, solver_path):
    """Garbage in: a NaN in an instance's header or in one of its rows must not come back LSCQP_STATUS_OPTIMAL (comparisons with NaN are
    false: a row whose slack is NaN would otherwise read as satisfied).  The neighbours in the batch are solved as usual."""
    from lsc_dr_planner_amd import synth

    N, M, dim = 12, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=8, seed=9)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
    hdr, rows = hdr.copy(), rows.copy()
    hdr["v0"][3][1] = np.nan                       # instance 3: its state
    rows["b"][int(roff[7]) + 2 * M * 6 + 17] = np.nan  # instance 7: one row of its third neighbour
    hdr["goal"][9][0] = np.inf                     # instance 9: its goal
    G = sol.solve_host(hdr, rows, roff, sfcp)
    bad = np.array([3, 7, 9])
    assert (G["status"][bad] != 0).all(), G["status"]
    assert (np.delete(G["status"], bad) == 0).all() and np.isfinite(np.delete(G["x"], bad, axis=0)).all()


@pytest.mark.gpu
def test_a_stale_work_order_is_refused_when_asked_to_check(api, oracle, torch_cuda, solver_path, monkeypatch):
    """lscqp_solve_batch_device_ordered trusts d_order to be a permutation of 0 .. n-1; LSCQP_CHECK_ORDER=1 (a debugging aid: it allocates and
    synchronises) verifies it on the device and refuses a duplicate or out-of-range entry instead of leaving an instance unsolved."""
    if solver_path != "active_set":
        pytest.skip("one path")
    torch = torch_cuda
    from lsc_dr_planner_amd import synth

    N, M, dim = 40, 5, 3
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=8, seed=4)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    args = (N, sw.n_obs, up(hdr), up(rows), up(roff), up(sfcp), d_x, d_obj, d_st, None)
    monkeypatch.setenv("LSCQP_CHECK_ORDER", "1")
    good = torch.from_numpy(np.random.default_rng(0).permutation(N).astype(np.int32)).to(dev)
    sol.solve_device(*args, d_order=good)
    torch.cuda.synchronize()
    assert (d_st.cpu().numpy() == 0).all()
    for bad in (np.r_[np.arange(N - 1), 0], np.r_[np.arange(N - 1), N]):
        with pytest.raises(api.LscqpError) as e:
            sol.solve_device(*args, d_order=torch.from_numpy(bad.astype(np.int32)).to(dev))
        assert e.value.code == api.ERR_INVALID_ARGUMENT and "permutation" in str(e.value)

"""LSCQP_ROWS_F32: the packed LSC rows stored as four floats (16 B), SURVEY.md section 8d's byte model for BASELINE
configs[4].  The arithmetic stays fp64, so every result on f32 rows must equal, bit for bit, the result on f64 rows that
hold the same float values; producers that write the format must write exactly the float32 rounding of their f64 output."""
import numpy as np
import pytest


def test_row_format_is_validated_and_counted(api):
    with pytest.raises(api.LscqpError) as e:
        api.Solver(api.make_desc(M=5, dim=3, row_format=2))
    assert e.value.code == api.ERR_INVALID_ARGUMENT and "row_format" in str(e.value)
    s32 = api.Solver(api.make_desc(M=5, dim=3, row_format=api.ROWS_F32))
    s64 = api.Solver(api.make_desc(M=5, dim=3))
    # 20 432 B for configs[1]; configs[4] with 16-byte rows: 9 600 + 240 + 256 + 720 + 16 (SURVEY's 10 712 also
    # stores the SFC boxes as floats, 120 B less; the boxes stay fp64 here)
    assert s32.algorithmic_bytes(20) == 10832 and s64.algorithmic_bytes(20) == 20432
    assert s64.generate_lsc_bytes(64, 20, 64) - s32.generate_lsc_bytes(64, 20, 64) == 64 * 20 * 30 * 16
    assert api.ROW_F32_DTYPE.itemsize == 16
    rows = np.zeros(7, api.ROW_DTYPE)
    rows["b"] = 1.0 + 1e-9
    assert s32.rows_in_format(rows).dtype == api.ROW_F32_DTYPE and s32.rows_in_format(rows)["b"][0] == np.float32(1.0)
    assert s64.rows_in_format(rows).dtype == api.ROW_DTYPE


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,dim,n_obs,seed", [(64, 5, 3, 20, 1), (10, 10, 2, 9, 2), (24, 10, 3, 40, 8)])
def test_solve_on_f32_rows_equals_solve_on_the_same_values_in_f64(api, N, M, dim, n_obs, seed):
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    s32 = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, row_format=api.ROWS_F32))
    s64 = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    r32 = s32.rows_in_format(rows)
    widened = np.zeros(r32.shape, api.ROW_DTYPE)
    for f in ("nx", "ny", "nz", "b"):
        widened[f] = r32[f]
    x0 = api.x_init_from_swarm(b, dim)
    A = s32.solve_host(hdr, r32, off, sfc, x_init=x0)
    B = s64.solve_host(hdr, widened, off, sfc, x_init=x0)
    assert (A["status"] == 0).all()
    assert np.array_equal(A["x"], B["x"]) and np.array_equal(A["obj"], B["obj"]) and np.array_equal(A["status"], B["status"])
    assert np.array_equal(A["info"]["iterations"], B["info"]["iterations"])
    # against the fp64 rows of the generator the solution moves by the float32 rounding of b (|b| <~ 30 m: 2e-6 m), not more
    Cc = s64.solve_host(hdr, rows, off, sfc, x_init=x0)
    assert np.abs(A["x"] - Cc["x"]).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_generators_write_the_float32_rounding_of_their_f64_rows(api, mode):
    import torch

    from lsc_dr_planner_amd import synth

    N, M, dim, n_obs = 96, 5, 3, 12
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=5)
    init, nbr = sw.initial_traj(), sw.neighbours().astype(np.int32)
    nbr[3, -1] = -1
    goal_all = np.float32(init[:, M - 1, 5] + 0.4).astype(np.float64)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    outs = {}
    for fmt, dt in ((api.ROWS_F64, api.ROW_DTYPE), (api.ROWS_F32, api.ROW_F32_DTYPE)):
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, row_format=fmt))
        d_rows = torch.full((N * n_obs * M * 6 * dt.itemsize,), 0xAB, dtype=torch.uint8, device=dev)
        sol.generate_constraints_device(mode, N, n_obs, 0, up(init), up(nbr), up(np.full(N, sw.radius)), up(np.full(N, sw.downwash)),
                                        up(goal_all), d_rows)
        torch.cuda.synchronize()
        outs[fmt] = d_rows.cpu().numpy().view(dt)
    for f in ("nx", "ny", "nz", "b"):
        assert np.array_equal(outs[api.ROWS_F32][f], outs[api.ROWS_F64][f].astype(np.float32)), f


@pytest.mark.gpu
def test_goal_lp_on_f32_rows(api):
    from lsc_dr_planner_amd import synth

    N, M, dim, n_obs = 48, 5, 3, 10
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=9)
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    hdr["next_waypoint"] = hdr["goal"] + np.array([0.5, -0.3, 0.1])
    s32 = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, row_format=api.ROWS_F32))
    s64 = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    r32 = s32.rows_in_format(rows)
    widened = np.zeros(r32.shape, api.ROW_DTYPE)
    for f in ("nx", "ny", "nz", "b"):
        widened[f] = r32[f]
    h32, st32 = s32.optimize_goal_host(hdr.copy(), r32, off, sfc)
    h64, st64 = s64.optimize_goal_host(hdr.copy(), widened, off, sfc)
    assert np.array_equal(st32, st64) and np.array_equal(h32["goal"], h64["goal"])

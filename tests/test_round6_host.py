"""Round 6, host side of the C ABI (csrc/lscqp_api.hip): switches that live in the handle, table buffers that an update cannot pull from under a
launch, small host-pointer calls on the mapped pinned mirror, the interior-point passes enqueued only when the phase left something, and
infeasibility proven inside the dual active-set phase on the bench's own construction."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import helpers as H  # noqa: F401


def _batch(api, N=16, M=5, dim=3, n_obs=8, seed=11, replans=2):
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    for _ in range(replans):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        sw.advance(sol.solve_host(hdr, rows, off, sfc, x_init=api.x_init_from_swarm(b, dim))["x"])
    b = sw.build()
    return sw, sol, b, api.batch_from_swarm(b, sw.n_obs, M), api.x_init_from_swarm(b, dim)


@pytest.mark.gpu
def test_the_library_reads_its_environment_only_when_a_handle_is_created(api, torch_cuda, monkeypatch):
    """A live handle does not change algorithm because the process environment changed under it: called straight through ctypes (not through
    api.Solver, whose wrapper re-reads the switches for the tests), a solve after setenv(LSCQP_ACTIVE_SET_NOW=0) is still the phase's; a handle
    created afterwards runs the interior-point kernel alone; lscqp_debug_reload_knobs_ is what flips a live one."""
    monkeypatch.delenv("LSCQP_ACTIVE_SET_NOW", raising=False)
    monkeypatch.delenv("LSCQP_ACTIVE_SET", raising=False)
    sw, sol, b, (hdr, rows, off, sfc), x0 = _batch(api)
    L = api.lib()
    n, nv = len(hdr), sol.nv

    def raw(handle):
        x, obj, st, info = np.zeros((n, nv)), np.zeros(n), np.full(n, -1, np.int32), np.zeros(n, api.INFO_DTYPE)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        r_ = sol.rows_in_format(rows)
        assert L.lscqp_solve_batch(handle, n, p(hdr), p(r_), p(off), p(sfc), p(x0), p(x), p(obj), p(st), p(info)) == 0
        return st, info

    os.environ["LSCQP_ACTIVE_SET_NOW"] = "0"
    try:
        st, info = raw(sol._h)
        assert (st == 0).all() and ((info["flags"] & api.INFO_ACTIVE_SET) != 0).all()  # still the phase: the handle was created before
        late = api.Solver(api.make_desc(M=5, dim=3, world_min=sw.world_min, world_max=sw.world_max))
        st2, info2 = raw(late._h)
        assert (st2 == 0).all() and ((info2["flags"] & api.INFO_ACTIVE_SET) == 0).all()  # created with the switch set: interior point alone
        assert L.lscqp_debug_reload_knobs_(sol._h) == 0
        st3, info3 = raw(sol._h)
        assert ((info3["flags"] & api.INFO_ACTIVE_SET) == 0).all()
    finally:
        os.environ.pop("LSCQP_ACTIVE_SET_NOW", None)
        L.lscqp_debug_reload_knobs_(sol._h)


@pytest.mark.gpu
def test_small_host_pointer_calls_on_the_mapped_mirror_return_the_copied_calls_bits(api, torch_cuda):
    sw, sol, b, (hdr, rows, off, sfc), x0 = _batch(api, N=24)
    sol.set_knob("zero_copy_bytes", 0)
    ref = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    sol.set_knob("zero_copy_bytes", 4 << 20)
    for _ in range(3):
        got = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
        for k in ("x", "obj", "status"):
            assert np.array_equal(got[k], ref[k]), k
        assert np.array_equal(got["info"]["iterations"], ref["info"]["iterations"])
    assert (ref["status"] == 0).all()


@pytest.mark.gpu
def test_an_update_cannot_pull_the_tables_from_under_a_launch_in_flight(api, torch_cuda):
    """lscqp_update with a changed dt gives the device a fresh table buffer; launches enqueued before it keep the old one (the class travels by
    value, the buffer is immutable): results of asynchronous solves enqueued right before an update are the un-updated class's, bit for bit."""
    torch = torch_cuda
    sw, sol, b, (hdr, rows, off, sfc), x0 = _batch(api, N=64, n_obs=12)
    n, nv, dev = len(hdr), sol.nv, torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    dh, dr, do, ds, dxi = up(hdr), up(sol.rows_in_format(rows)), up(off), up(sfc), torch.from_numpy(x0).to(dev)
    outs = [(torch.zeros(n * nv, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
            for _ in range(40)]
    ref = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    d_old = api.make_desc(M=5, dim=3, world_min=sw.world_min, world_max=sw.world_max)
    d_new = api.make_desc(M=5, dim=3, dt=0.25, world_min=sw.world_min, world_max=sw.world_max)
    st = torch.cuda.Stream()
    for i, (x, o, s) in enumerate(outs):
        sol.solve_device(n, sw.n_obs, dh, dr, do, ds, x, o, s, None, stream=st, d_x_init=dxi)
        if i % 2 == 1:
            sol.update(d_new)  # (a synchronous table build + upload on the host thread while the stream still works)
            sol.update(d_old)
    torch.cuda.synchronize()
    for x, o, s in outs:
        assert np.array_equal(x.cpu().numpy().reshape(n, nv), ref["x"]) and np.array_equal(s.cpu().numpy(), ref["status"])
    assert api.lib().lscqp_prepare_device(sol._h) == 0  # idempotent: the tables are there


@pytest.mark.gpu
def test_the_phase_proves_the_benchs_infeasible_instances_and_nothing_else(api, oracle, torch_cuda):
    import bench

    sw, sol, b, (hdr, rows, off, sfc), x0 = _batch(api, N=64, n_obs=20, seed=1000, replans=3)
    rows_bad, sel = bench.make_infeasible(api, rows, hdr, sw.n_obs, 5, 3 / 64, 5)
    G = sol.solve_host(hdr, rows_bad, off, sfc, x_init=x0)
    clean = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    assert (clean["status"] == 0).all()
    others = np.setdiff1d(np.arange(64), sel)
    assert (G["status"][sel] != 0).all() and (G["status"][others] == 0).all()
    assert np.array_equal(G["x"][others], clean["x"][others])  # the neighbours in the batch are untouched by the failures
    # (the phase proves what it reaches inside its step budget -- on this construction 74 steps and more -- and hands the rest over: whichever
    # kernel gives the verdict, an instance the PHASE answers is INFEASIBLE with its violation, nothing else)
    proven = (G["info"]["flags"][sel] & api.INFO_ACTIVE_SET) != 0
    assert (G["status"][sel][proven] == api.STATUS_INFEASIBLE).all() and (G["info"]["res_primal"][sel][proven] > 1e-6).all()
    # the phase alone: an infeasible instance is PROVEN so (INFEASIBLE) or left with its reason (ITER_LIMIT + LSCQP_DAS_WHY_*), never anything else
    only = api.Solver(api.make_desc(M=5, dim=3, world_min=sw.world_min, world_max=sw.world_max, active_set=api.ACTIVE_SET_ONLY))
    O1 = only.solve_host(hdr, rows_bad, off, sfc, x_init=x0)
    assert (O1["status"][others] == 0).all()
    for q in sel:
        if O1["status"][q] == api.STATUS_INFEASIBLE:
            assert O1["info"]["flags"][q] & api.INFO_ACTIVE_SET and O1["info"]["res_primal"][q] > 1e-6
        else:
            assert O1["status"][q] == api.STATUS_ITER_LIMIT and int(O1["info"]["res_dual"][q]) in (api.DAS_WHY_ROWS, api.DAS_WHY_STEPS, api.DAS_WHY_NO_STEP, api.DAS_WHY_PIVOT)
    # the checker agrees: no point satisfies those rows
    cls = oracle.make_class(M=5, dim=3, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    ag, lsc, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
    per = sw.n_obs * 5 * 6
    rb = rows_bad.reshape(64, -1)
    for q in sel:
        lq = np.zeros(per, oracle.LSC_DTYPE)
        lq["nrm"][:, 0], lq["nrm"][:, 1], lq["nrm"][:, 2], lq["d"] = rb["nx"][q], rb["ny"][q], rb["nz"][q], rb["b"][q]
        assert oracle.solve(cls, ag[q:q + 1], lq, np.ascontiguousarray(sfco[q * 5:(q + 1) * 5]), max_iter=100)["status"] != 0
    # with the phase off the same instances fail too (whatever the kernel calls the failure), the others solve to the same optimum
    sol.set_knob("active_set_off", 1)
    try:
        P = sol.solve_host(hdr, rows_bad, off, sfc, x_init=x0)
    finally:
        sol.set_knob("active_set_off", 0)
    assert np.array_equal(P["status"] != 0, G["status"] != 0) and np.abs(P["x"][others] - G["x"][others]).max() <= 1e-6

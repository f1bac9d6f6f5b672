"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same inputs,
against the committed golden fixtures, and - at BASELINE.json's full sizes - through size-independent properties.

Tolerances (fp64, stated by north_star as <= 1e-8):
  objective     |obj_gpu - obj_oracle| <= 1e-8 * max(1, |obj|)
  KKT           scaled stationarity, equality and inequality violation of the GPU point on the reference's own
                row-for-row model <= 1e-8
  control points (the optimum is unique)  max |dx| <= 1e-6 m   (float32 output precision of the reference is ~1e-7 m)
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

OBJ_TOL = 1e-8
KKT_TOL = 1e-8
X_TOL = H.PathTol()  # 1e-8 m for the dual active-set phase, 1e-6 m for the interior-point kernel (tests/helpers.py)


def _check_against_oracle(O, cls, G, R, sel=None):
    ok = (R["status"] == 0)
    assert ok.all(), "oracle failed on %s" % np.where(~ok)[0]
    assert (G["status"] == 0).all(), "gpu status %s" % np.bincount(G["status"], minlength=4)
    dx = np.abs(G["x"] - R["x"]).max()
    do = (np.abs(G["obj"] - R["obj"]) / np.maximum(1.0, np.abs(R["obj"]))).max()
    assert dx <= X_TOL, dx
    assert do <= OBJ_TOL, do
    return dx, do


@pytest.mark.parametrize("N,M,dim,n_obs,style,seed,steps", [
    (64, 5, 3, 20, "forest", 1, 3),    # BASELINE configs[1]
    (10, 10, 2, 9, "forest", 2, 3),    # forest10 replica (reference launch files: M=10, dim=2)
    (10, 5, 2, 9, "forest", 5, 2),     # forest10 with the M=5 default of src/param.cpp:71
    (48, 6, 3, 20, "maze", 3, 3),      # dense-maze set, M=6 (configs[2] shape, fewer agents)
    (24, 3, 3, 10, "forest", 4, 2),
    (20, 7, 3, 12, "maze", 6, 2),      # largest dim-3 horizon of the one-wavefront kernel (nz = 57)
    (24, 10, 3, 40, "forest", 8, 2),   # configs[3] shape: M=10, dim 3, 40 neighbours + SFC (nz = 84: two wavefronts per QP)
])
def test_swarm_parity(api, oracle, torch_cuda, N, M, dim, n_obs, style, seed, steps):
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    for step in range(steps + 1):
        b = sw.build()
        ag, lsc, off, sfc = H.swarm_oracle_inputs(oracle, sw, b)
        R = oracle.solve_batch(cls, ag, lsc, off, sfc, threads=8)
        hdr, rows, roff, sfcp = api.batch_from_swarm(b, sw.n_obs, M)
        hdr["terminal_segments"] = [oracle.terminal_segments(cls, ag[q:q + 1]) for q in range(N)]
        G = sol.solve_host(hdr, rows, roff, sfcp)
        _check_against_oracle(oracle, cls, G, R)
        # KKT residuals of the GPU point on the reference's row-for-row model: a few agents per step, EVERY agent for the nz = 84 class
        # (where the solver's own stationarity test has a documented rounding floor: this is the check that does not)
        for q in range(0, N, 1 if dim * (3 * M - 2) > 64 else max(1, N // 4)):
            lq = np.ascontiguousarray(b["lsc"][q]); sq = np.ascontiguousarray(b["sfc"][q])
            stat, eqv, iqv = H.kkt_from_primal(oracle, cls, ag[q:q + 1], lq, sq, G["x"][q])
            assert stat <= KKT_TOL and eqv <= KKT_TOL and iqv <= KKT_TOL, (step, q, stat, eqv, iqv)
        # the solver's own scaled stationarity residual: <= 1e-8, except that at nz = 84 (cond(Hred) ~ 1e7) single
        # QPs stop at its rounding floor (<= 1e-6, documented in lscqp_kernel.hpp); the KKT residuals on the reference's
        # model, checked above with the 1e-8 bar, are not affected
        rd_bar = 1e-8  # (round 5: every shape; the rounding floor of the nz = 84 class went with round 4's centring floor, tests/test_floor_audit.py)
        assert G["info"]["res_primal"].max() <= 1e-9 and G["info"]["res_dual"].max() <= rd_bar
        sw.advance(G["x"])  # the swarm is carried forward by the GPU solution


@pytest.mark.parametrize("name", ["scipy_m5d3", "scipy_m10d2", "scipy_m6d3_maze"])
def test_golden_fixtures(api, oracle, torch_cuda, name):
    g = H.load_golden(name)
    p = g["params"]
    cls = H.oracle_class(oracle, p)
    sol = api.Solver(H.abi_desc(api, p))
    ags, lscs, sfcs = zip(*[H.golden_case_arrays(oracle, p, c) for c in g["cases"]])
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, lscs, sfcs, p["M"])
    G = sol.solve_host(hdr, rows, off, sfc)
    assert (G["status"] == 0).all()
    for q, c in enumerate(g["cases"]):
        assert np.abs(G["x"][q] - np.array(c["x"])).max() < 1e-7       # 80-bit polished optimum
        assert np.abs(G["x"][q] - np.array(c["scipy_x"])).max() < 1e-5  # raw scipy trust-constr iterate
        assert abs(G["obj"][q] - c["obj"]) <= OBJ_TOL * max(1.0, abs(c["obj"]))


def test_reference_log_known_answers(api, oracle, torch_cuda):
    """First replan of forest10_10 from the reference's own result log, solved on the GPU."""
    g = H.load_golden("kat_log")
    p = g["params"]
    for case in g["cases"]:
        use_sfc = case["sfc"] is not None
        cls = H.oracle_class(oracle, p, use_sfc=use_sfc)
        sol = api.Solver(H.abi_desc(api, p, use_sfc=use_sfc))
        ag = oracle.make_agent(p0=case["p0"], goal=case["goal"], next_waypoint=case["next_waypoint"], vmax=p["vmax"],
                               amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"])
        sfcs = None
        if use_sfc:
            s = np.zeros(p["M"], oracle.BOX_DTYPE)
            s["bmin"], s["bmax"] = case["sfc"]["bmin"], case["sfc"]["bmax"]
            sfcs = [s]
        hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, [ag], [None], sfcs, p["M"])
        G = sol.solve_host(hdr, None, None, sfc)
        assert G["status"][0] == 0
        for st in g["agents"][case["agent"]]["states"][1:]:
            pos, vel, acc = oracle.state_at(cls, G["x"][0], st["t"])
            assert np.allclose(pos, st["p"][:2], rtol=0, atol=2e-5)
            assert np.allclose(vel, st["v"][:2], rtol=1e-4, atol=2e-6)
            assert np.allclose(acc, st["a"][:2], rtol=3e-4, atol=2e-5)


def test_edge_cases(api, oracle, torch_cuda):
    """No obstacles, ragged obstacle counts, zero normals, DLSC mode (no end stop), no SFC, no comm range."""
    from lsc_dr_planner_amd import synth

    M, dim, N = 5, 3, 12
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=8, seed=9)
    b = sw.build()
    x0 = None
    for variant in ("ragged", "zero_normals", "dlsc", "no_sfc", "no_comm"):
        planner_lsc = variant != "dlsc"
        use_sfc = variant != "no_sfc"
        comm = 0.0 if variant == "no_comm" else 3.0
        cls = oracle.make_class(M=M, dim=dim, use_sfc=use_sfc, planner_lsc=planner_lsc, comm_range=comm,
                                world_min=sw.world_min, world_max=sw.world_max)
        sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=use_sfc, comm_range=comm,
                                       planner_mode=api.PLANNER_LSC if planner_lsc else api.PLANNER_DLSC,
                                       world_min=sw.world_min, world_max=sw.world_max))
        ags, lscs, sfcs = [], [], []
        for q in range(N):
            nob = (q % 9) if variant == "ragged" else 8      # includes n_obs == 0
            lq = np.ascontiguousarray(b["lsc"][q][:nob]).copy() if nob else None
            if variant == "zero_normals" and lq is not None:
                lq["nrm"][q % nob, :, :] = 0.0                # dropped by the ||n|| < 1e-5 rule
                lq["nrm"][(q + 1) % nob, 2, 3] = (3e-6, 0, 0)
            ags.append(oracle.make_agent(p0=b["p0"][q], v0=b["v0"][q], a0=b["a0"][q], goal=b["goal"][q],
                                         next_waypoint=b["next_waypoint"][q], n_obs=nob))
            lscs.append(lq)
            sfcs.append(np.ascontiguousarray(b["sfc"][q]))
        hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, lscs, sfcs if use_sfc else None, M)
        G = sol.solve_host(hdr, rows, off, sfc)
        assert (G["status"] == 0).all(), (variant, G["status"])
        for q in range(N):
            r = oracle.solve(cls, ags[q], lscs[q], sfcs[q] if use_sfc else None)
            assert r["status"] == 0
            assert np.abs(G["x"][q] - r["x"]).max() <= X_TOL, (variant, q)
            assert abs(G["obj"][q] - r["obj"]) <= OBJ_TOL * max(1, abs(r["obj"])), (variant, q)


def test_infeasible_instances_are_reported(api, oracle, torch_cuda):
    """Status per instance; the shim turns != OPTIMAL into `throw PlanningReport::QPFAILED`
    (src/traj_optimizer.cpp:143) and the caller falls back to initial_traj (src/traj_planner.cpp:767-797)."""
    from lsc_dr_planner_amd import synth

    M, dim, N = 5, 3, 16
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=6, seed=21)
    b = sw.build()
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    ags, lscs, sfcs = [], [], []
    bad = {3: "lsc", 7: "box", 11: "lsc_pair"}
    for q in range(N):
        lq = np.ascontiguousarray(b["lsc"][q]).copy()
        sq = np.ascontiguousarray(b["sfc"][q]).copy()
        if bad.get(q) == "lsc":
            lq["d"][0] += 100.0                                 # half-space far outside the world box
        if bad.get(q) == "box":
            sq["bmin"][2] = sq["bmax"][2] + 0.5                  # empty SFC box on one segment
        if bad.get(q) == "lsc_pair":
            lq["nrm"][1] = -lq["nrm"][0]; lq["p"][1] = lq["p"][0]; lq["d"][1] = 0.2; lq["d"][0] = 0.2  # opposing slabs
        ags.append(oracle.make_agent(p0=b["p0"][q], goal=b["goal"][q], next_waypoint=b["next_waypoint"][q], n_obs=6))
        lscs.append(lq); sfcs.append(sq)
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, lscs, sfcs, M)
    G = sol.solve_host(hdr, rows, off, sfc)
    for q in range(N):
        if q in bad:
            assert G["status"][q] != 0, (q, bad[q], G["info"][q])
            # reported early (a stalled primal residual), not at the iteration limit of 60: a launch lasts as long as its
            # slowest QP, and the caller falls back to the initial trajectory anyway
            if G["info"]["flags"][q] & api.INFO_ACTIVE_SET:
                # round 6: PROVEN inside the dual active-set phase (a Farkas certificate; csrc/lscqp_das.hip) -- its count is of active-set steps,
                # bounded by the launch's budget, and the verdict is INFEASIBLE, nothing else
                assert G["status"][q] == api.STATUS_INFEASIBLE and G["info"]["iterations"][q] <= 96 and G["info"]["res_primal"][q] > 1e-6, (q, bad[q], G["info"][q])
            else:
                assert G["info"]["iterations"][q] <= 24, (q, bad[q], G["info"][q])
            assert oracle.solve(cls, ags[q], lscs[q], sfcs[q], max_iter=100)["status"] != 0
        else:
            assert G["status"][q] == 0, (q, G["info"][q])


def test_device_resident_path_and_batch_of_one(api, oracle, torch_cuda):
    torch = torch_cuda
    from lsc_dr_planner_amd import synth

    M, dim, N = 5, 3, 32
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=12, seed=33)
    b = sw.build()
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    ref = sol.solve_host(hdr, rows, off, sfc)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) for a in (hdr, rows, off, sfc)]
    d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        sol.solve_device(N, sw.n_obs, t[0], t[1], t[2], t[3], d_x, d_obj, d_st, None)
    stream.synchronize()
    assert (d_st.cpu().numpy() == 0).all()
    # deterministic: bitwise equal to the host-pointer path (same kernel, same inputs)
    assert np.array_equal(d_x.cpu().numpy().reshape(N, -1), ref["x"])
    assert np.array_equal(d_obj.cpu().numpy(), ref["obj"])
    # batch of one == the sequential loop of the unchanged simulator (src/multi_sync_simulator.cpp:357-362)
    for q in (0, 5, N - 1):
        lo, hi = int(off[q]), int(off[q + 1])
        one = sol.solve_host(hdr[q:q + 1], rows[lo:hi], np.array([0, hi - lo], dtype=np.uint64), sfc[q:q + 1])
        assert np.array_equal(one["x"][0], ref["x"][q])


@pytest.mark.parametrize("N,M,dim,n_obs,style", [(512, 6, 3, 20, "maze"), (4096, 5, 3, 20, "forest"), (128, 10, 3, 40, "forest")])
def test_full_size_properties(api, oracle, torch_cuda, N, M, dim, n_obs, style):
    """BASELINE configs[2] / configs[4] / configs[3] shapes (per GPU: 512, 4096 and 1024/8 agents): properties that do not
    need the oracle on every instance."""
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=100 + N, style=style)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    for step in range(2):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        G = sol.solve_host(hdr, rows, off, sfc)
        assert (G["status"] == 0).all(), np.bincount(G["status"], minlength=4)
        # rounding floor of the nz = 84 class, see test_swarm_parity; the flag is set at <= 1e-6 and the iterate that is
        # finally returned after the stall may sit marginally above it
        rd_bar = 1e-8  # (round 5: every shape; the rounding floor of the nz = 84 class went with round 4's centring floor, tests/test_floor_audit.py)
        assert G["info"]["res_primal"].max() <= 1e-9 and G["info"]["res_dual"].max() <= rd_bar and G["info"]["gap"].max() <= 1e-9
        x = G["x"].reshape(N, dim, M, 6)
        # 1. eliminated equalities hold: initial state, C0/C1/C2 joins, end stop (src/traj_optimizer.cpp:318-368,502-511)
        assert np.abs(x[:, :, 0, 0] - b["p0"][:, :dim]).max() < 1e-12
        assert np.abs((x[:, :, 0, 1] - x[:, :, 0, 0]) * 25 - b["v0"][:, :dim]).max() < 1e-9
        assert np.abs(x[:, :, 1:, 0] - x[:, :, :-1, 5]).max() < 1e-12
        assert np.abs((x[:, :, 1:, 1] - x[:, :, 1:, 0]) - (x[:, :, :-1, 5] - x[:, :, :-1, 4])).max() < 1e-12
        assert np.abs(x[:, :, -1, 5] - x[:, :, -1, 3]).max() < 1e-12
        # 2. every inequality of the reference model holds (vectorised restatement of the row families)
        v = np.abs(np.diff(x, axis=3)) * 25
        assert (v[:, :, 1:] <= 1 + 1e-8).all() and (v[:, :, 0, 2:] <= 1 + 1e-8).all()
        a = np.abs(np.diff(x, 2, axis=3)) * 500
        assert (a[:, :, 1:] <= 2 + 1e-7).all() and (a[:, :, 0, 1:] <= 2 + 1e-7).all()
        cp = x.transpose(0, 2, 3, 1)                                            # (N, M, 6, dim)
        lsc = b["lsc"]
        marg = ((cp[:, None, :, :, :] - lsc["p"][..., :dim]) * lsc["nrm"][..., :dim]).sum(-1) - lsc["d"]
        marg[:, :, 0, :3] = 1.0
        assert marg.min() >= -1e-8
        assert (cp >= b["sfc"]["bmin"][:, :, None, :dim] - 1e-8).all() and (cp <= b["sfc"]["bmax"][:, :, None, :dim] + 1e-8).all()
        # 3. determinism and permutation equivariance of the batch
        perm = np.random.default_rng(step).permutation(N)
        rows2 = rows.reshape(N, -1)[perm].reshape(-1)
        G2 = sol.solve_host(hdr[perm], rows2, off, sfc[perm])
        assert np.array_equal(G2["x"], G["x"][perm]) and np.array_equal(G2["obj"], G["obj"][perm])
        # 4. spot parity against the oracle on a bounded sample
        ag, lsco, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
        for q in range(0, N, N // 16):
            r = oracle.solve(cls, ag[q:q + 1], np.ascontiguousarray(b["lsc"][q]), np.ascontiguousarray(b["sfc"][q]))
            assert r["status"] == 0
            assert np.abs(G["x"][q] - r["x"]).max() <= X_TOL
            assert abs(G["obj"][q] - r["obj"]) <= OBJ_TOL * max(1, abs(r["obj"]))
        sw.advance(G["x"])


def test_translation_invariance(api, oracle, torch_cuda):
    """Shifting the whole world shifts the solution and leaves the objective unchanged (to the reference model's
    own ~1e-9 |x|^2 rounding of Q_base)."""
    from lsc_dr_planner_amd import synth

    M, dim, N = 5, 3, 16
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=8, seed=77)
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    G = sol.solve_host(hdr, rows, off, sfc)
    shift = np.array([3.0, -2.0, 1.0])   # exactly representable
    hdr2 = hdr.copy()
    for f in ("p0", "goal", "next_waypoint"):
        hdr2[f] += shift
    rows2 = rows.copy()
    rows2["b"] += rows["nx"] * shift[0] + rows["ny"] * shift[1] + rows["nz"] * shift[2]
    sfc2 = sfc.copy(); sfc2["bmin"] += shift; sfc2["bmax"] += shift
    sol2 = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min + shift, world_max=sw.world_max + shift))
    G2 = sol2.solve_host(hdr2, rows2, off, sfc2)
    assert (G["status"] == 0).all() and (G2["status"] == 0).all()
    assert np.abs(G2["x"] - (G["x"] + np.repeat(shift, M * 6))).max() < 1e-9
    assert np.abs(G2["obj"] - G["obj"]).max() < 1e-7


def _compiled_instances():
    import os
    import re

    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lsc_dr_planner_amd", "csrc",
                            "lscqp_launch.hpp")).read()
    body = txt[txt.index("#define LSCQP_INSTANCES"):]
    inst = [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", body)]
    assert len(inst) >= 20, "the instance list of lscqp_launch.hpp was not parsed"
    return [i[:5] for i in inst if i[5] == 0]  # (the mixed-precision instances have their own tests)


@pytest.mark.pdip_only
@pytest.mark.parametrize("M,dim,es,nslot,waves", _compiled_instances())
def test_every_compiled_instance_is_deterministic_and_exact(api, oracle, torch_cuda, request, M, dim, es, nslot, waves):
    """Every kernel instance, at its full obstacle capacity: bitwise repeatable and equal to the oracle.
    (Guards against the exec-masked register-spill hazard described in lscqp_kernel.hpp: a miscompiled instance shows
    up as run-to-run differences long before it shows up as a wrong answer.)"""
    from lsc_dr_planner_amd import synth

    import os

    os.environ["LSCQP_WAVES"] = str(waves)  # pin the launch policy to the instance under test
    request.addfinalizer(lambda: os.environ.pop("LSCQP_WAVES", None))
    G = max(1, 64 * waves // (6 * M - 3))
    n_obs = nslot * G
    N = max(24, n_obs + 2)
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=17 + M)
    assert sw.n_obs == n_obs
    pm = api.PLANNER_LSC if es else api.PLANNER_DLSC
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, planner_lsc=bool(es), world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=pm, world_min=sw.world_min, world_max=sw.world_max))
    for step in range(2):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        runs = [sol.solve_host(hdr, rows, off, sfc) for _ in range(3)]
        assert (runs[0]["status"] == 0).all(), np.bincount(runs[0]["status"], minlength=4)
        for r in runs[1:]:
            assert np.array_equal(r["x"], runs[0]["x"]) and np.array_equal(r["obj"], runs[0]["obj"])
            assert np.array_equal(r["info"]["iterations"], runs[0]["info"]["iterations"])
        ag, lsc, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
        R = oracle.solve_batch(cls, ag, lsc, loff, sfco, threads=8)
        _check_against_oracle(oracle, cls, runs[0], R)
        sw.advance(runs[0]["x"])


@pytest.mark.pdip_only
@pytest.mark.parametrize("N,M,dim,n_obs,style,seed", [(48, 5, 3, 20, "forest", 21), (16, 10, 2, 9, "forest", 22), (24, 10, 3, 40, "forest", 23)])
def test_initial_trajectory_as_primal_start(api, oracle, torch_cuda, N, M, dim, n_obs, style, seed):
    """x_init (TrajOptimizer::solve's initial_traj, the shifted previous plan) only moves the starting point of the
    interior-point iteration: same optimum as the cold start and as the oracle, fewer iterations in steady state."""
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    it_cold = it_warm = 0
    for step in range(4):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        cold = sol.solve_host(hdr, rows, off, sfc)
        warm = sol.solve_host(hdr, rows, off, sfc, x_init=api.x_init_from_swarm(b, dim))
        assert (cold["status"] == 0).all() and (warm["status"] == 0).all()
        assert np.abs(warm["x"] - cold["x"]).max() <= X_TOL
        assert (np.abs(warm["obj"] - cold["obj"]) / np.maximum(1.0, np.abs(cold["obj"]))).max() <= OBJ_TOL
        ag, lsc, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
        R = oracle.solve_batch(cls, ag, lsc, loff, sfco, threads=8)
        _check_against_oracle(oracle, cls, warm, R)
        if step >= 1:  # step 0 starts from hover: the initial trajectory is the hover itself
            it_cold += cold["info"]["iterations"].sum()
            it_warm += warm["info"]["iterations"].sum()
        sw.advance(warm["x"])
    assert it_warm <= it_cold


@pytest.mark.pdip_only
def test_jammed_warm_start_is_solved_from_the_default_start(api, oracle, torch_cuda):
    """An instance from the 64-agent closed loop (tests/golden/warm_start_jam.json) whose iteration, started from the shifted
    previous plan, stalls with the gap near 6e-7 unless the row state is re-centred: both entry points return the optimum (the
    oracle's, to the parity bar)."""
    torch = torch_cuda
    g = H.load_golden("warm_start_jam")
    M, dim, n_obs = g["M"], g["dim"], g["n_obs"]
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=g["world_min"], world_max=g["world_max"]))
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=g["world_min"], world_max=g["world_max"])
    hdr = np.zeros(1, api.HEADER_DTYPE)
    for f, v in g["hdr"].items():
        hdr[f][0] = v
    hdr["n_obs"][0] = n_obs
    R = np.array(g["rows"])
    rows = np.zeros(len(R), api.ROW_DTYPE)
    rows["nx"], rows["ny"], rows["nz"], rows["b"] = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
    sfc = np.zeros(M, api.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = g["sfc_min"], g["sfc_max"]
    off = np.array([0, len(R)], dtype=np.uint64)
    x0 = np.array(g["x_init"])[None]
    G = sol.solve_host(hdr, rows, off, sfc, x_init=x0)
    assert G["status"][0] == 0
    ag = oracle.make_agent(n_obs=n_obs, **{k: v for k, v in g["hdr"].items()})
    lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
    lsc["nrm"] = R[:, :3].reshape(n_obs, M, 6, 3)
    lsc["d"] = R[:, 3].reshape(n_obs, M, 6)
    o = oracle.solve(cls, ag, lsc, sfc)
    assert o["status"] == 0
    assert abs(o["obj"] - G["obj"][0]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][0]).max() <= 1e-6
    # the device entry point cannot look at the status without synchronising: it reports ITER_LIMIT, and a cold re-launch solves it
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_x = torch.zeros(sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(1, dtype=torch.float64, device=dev)
    d_st = torch.full((1,), -1, dtype=torch.int32, device=dev)
    args = (1, n_obs, up(hdr), up(rows), up(off), up(sfc), d_x, d_obj, d_st)
    d_info = torch.zeros(api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    sol.solve_device(*args, d_info=d_info, d_x_init=torch.from_numpy(x0.copy()).to(dev))
    torch.cuda.synchronize()
    # the kernel notices the jam (no tenfold improvement of the gap within six converged-residual iterations), re-centres the
    # row state once at the current point and finishes: far fewer than the 60 + 8 iterations of a failed attempt plus a cold one
    assert d_st.item() == 0 and abs(d_obj.item() - o["obj"]) <= OBJ_TOL * max(1.0, abs(o["obj"]))
    assert d_info.cpu().numpy().view(api.INFO_DTYPE)["iterations"][0] <= 40
    sol.solve_device(*args)  # and the default start needs no help
    torch.cuda.synchronize()
    assert d_st.item() == 0 and abs(d_obj.item() - o["obj"]) <= OBJ_TOL * max(1.0, abs(o["obj"]))


@pytest.mark.pdip_only
def test_limit_cycle_of_the_iteration_is_ended_by_the_rescue_pass(api, oracle, torch_cuda):
    """tests/golden/limit_cycle_dlsc.json: a DLSC instance (M = 10, 3-D, 23 neighbours) on which the predictor-corrector iteration runs
    into a limit cycle from the default start -- period four, gap 6e-6 .. 6e-5, the re-centring does not break it -- and ends at the
    iteration limit on the compiled instances and on the run-time-shaped kernel alike.  The host-pointer entry (what the shim calls) and
    retry = 2 of the device entry re-solve such an instance in the rescue pass (run-time-shaped kernel, the corrector's second-order
    term weighted by the blocked affine step length): the optimum is the oracle's, lscqp_info says LSCQP_INFO_RESCUED."""
    torch = torch_cuda
    g = H.load_golden("limit_cycle_dlsc")
    M, dim, n_obs = g["M"], g["dim"], g["n_obs"]
    sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=api.PLANNER_DLSC, world_min=g["world_min"], world_max=g["world_max"]))
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, planner_lsc=False, world_min=g["world_min"], world_max=g["world_max"])
    hdr = np.zeros(1, api.HEADER_DTYPE)
    for f, v in g["hdr"].items():
        hdr[f][0] = v
    hdr["n_obs"][0] = n_obs
    R = np.array(g["rows"])
    rows = np.zeros(len(R), api.ROW_DTYPE)
    rows["nx"], rows["ny"], rows["nz"], rows["b"] = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
    sfc = np.zeros(M, api.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = g["sfc_min"], g["sfc_max"]
    off = np.array([0, len(R)], dtype=np.uint64)
    ag = oracle.make_agent(n_obs=n_obs, **{k: v for k, v in g["hdr"].items()})
    lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
    lsc["nrm"] = R[:, :3].reshape(n_obs, M, 6, 3)
    lsc["d"] = R[:, 3].reshape(n_obs, M, 6)
    o = oracle.solve(cls, ag, lsc, sfc)
    assert o["status"] == 0

    def agrees(obj, x):
        return abs(o["obj"] - obj) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - x).max() <= X_TOL

    G = sol.solve_host(hdr, rows, off, sfc)
    assert G["status"][0] == 0 and agrees(G["obj"][0], G["x"][0])
    # (the flag is what makes this fixture a test of the rescue pass: should a later first pass solve the instance, look for a new one)
    assert G["info"]["flags"][0] & api.INFO_RESCUED and G["info"]["flags"][0] & api.INFO_REPAIRED
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_x = torch.zeros(sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(1, dtype=torch.float64, device=dev)
    d_st = torch.full((1,), -1, dtype=torch.int32, device=dev)
    d_info = torch.zeros(api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    args = (1, n_obs, up(hdr), up(rows), up(off), up(sfc), d_x, d_obj, d_st)
    sol.solve_device(*args, d_info=d_info)  # one pass: the honest verdict, never a wrong optimum
    torch.cuda.synchronize()
    assert d_st.item() == api.STATUS_ITER_LIMIT
    sol.solve_device(*args, d_info=d_info, retry=2)
    torch.cuda.synchronize()
    assert d_st.item() == 0 and agrees(d_obj.item(), d_x.cpu().numpy())
    assert d_info.cpu().numpy().view(api.INFO_DTYPE)["flags"][0] & api.INFO_RESCUED
    # an infeasible instance is none of the rescue pass's business: same verdict, no flag
    sfc2 = sfc.copy()  # (a corridor five metres from the start state: the acceleration limits keep the first free control points near it)
    sfc2["bmin"], sfc2["bmax"] = hdr["p0"][0] + 5.0, hdr["p0"][0] + 6.0
    G2 = sol.solve_host(hdr, rows, off, sfc2)
    assert G2["status"][0] == api.STATUS_INFEASIBLE and not (G2["info"]["flags"][0] & api.INFO_RESCUED)


@pytest.mark.pdip_only
def test_an_iteration_repeated_with_a_diagonal_shift_is_counted_once(api, oracle, torch_cuda):
    """tests/golden/shifted_pivot.json (tools/make_golden_shifted.py): a cold M = 10, 3-D, 40-neighbour instance whose interior-point solve
    loses a pivot of a late iteration's matrix -- rounding, not the matrix -- and repeats that iteration with a diagonal shift
    (LSCQP_INFO_SHIFTED).  The repeat happens at the SAME point: its stopping tests and counters (confirmations at the rounding floor, the
    jam and stall counters) were taken the first time and must not be taken again, and lscqp_info.iterations counts the iteration once --
    the number the fixture recorded (one point counted twice could have met the two-confirmation acceptance alone).  The
    optimum is the oracle's; the run-time-shaped kernel, which carries the same logic, agrees."""
    g = H.load_golden("shifted_pivot")
    M, dim, n_obs = g["M"], g["dim"], g["n_obs"]
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=g["world_min"], world_max=g["world_max"], active_set=api.ACTIVE_SET_OFF))
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=g["world_min"], world_max=g["world_max"])
    hdr = np.zeros(1, api.HEADER_DTYPE)
    for f, v in g["hdr"].items():
        hdr[f][0] = v
    hdr["n_obs"][0] = n_obs
    R = np.array(g["rows"])
    rows = np.zeros(len(R), api.ROW_DTYPE)
    rows["nx"], rows["ny"], rows["nz"], rows["b"] = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
    sfc = np.zeros(M, api.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = g["sfc_min"], g["sfc_max"]
    off = np.array([0, len(R)], dtype=np.uint64)
    ag = oracle.make_agent(n_obs=n_obs, **{k: v for k, v in g["hdr"].items()})
    lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
    lsc["nrm"] = R[:, :3].reshape(n_obs, M, 6, 3)
    lsc["d"] = R[:, 3].reshape(n_obs, M, 6)
    o = oracle.solve(cls, ag, lsc, sfc)
    assert o["status"] == 0
    G = sol.solve_host(hdr, rows, off, sfc)
    assert G["status"][0] == 0
    assert abs(o["obj"] - G["obj"][0]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][0]).max() <= X_TOL
    # (the flag is what makes this fixture a test of the repeated iteration: should a later kernel keep the pivot, look for a new instance)
    assert G["info"]["flags"][0] & api.INFO_SHIFTED
    assert G["info"]["iterations"][0] == g["iterations"], (G["info"]["iterations"][0], g["iterations"])


@pytest.mark.pdip_only
def test_pivot_breakdown_in_a_multi_wavefront_instance_ends_the_whole_workgroup(api, oracle, torch_cuda):
    """tests/golden/pivot_breakdown_w2.json: an M = 6 dense-maze instance whose factorisation breaks down after the acceptance tests
    were met at the rounding floor.  In the two-wavefront instance only wavefront 0 holds the system and sees the failed pivot;
    round 1 let it leave the loop alone, the other wavefront ran on against mismatched barriers to the iteration limit (1.6 ms
    instead of 0.25 ms for the launch, and its share of x_out written from a garbage iterate).  Now the verdict is posted to the
    whole workgroup: the launch is short, the result is the remembered accepted point and agrees with the oracle."""
    torch = torch_cuda
    g = H.load_golden("pivot_breakdown_w2")
    M, dim, n_obs = g["M"], g["dim"], g["n_obs"]
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=g["world_min"], world_max=g["world_max"]))
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=g["world_min"], world_max=g["world_max"])
    hdr = np.zeros(1, api.HEADER_DTYPE)
    for f, v in g["hdr"].items():
        hdr[f][0] = v
    hdr["n_obs"][0] = n_obs
    R = np.array(g["rows"])
    rows = np.zeros(len(R), api.ROW_DTYPE)
    rows["nx"], rows["ny"], rows["nz"], rows["b"] = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
    sfc = np.zeros(M, api.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = g["sfc_min"], g["sfc_max"]
    off = np.array([0, len(R)], dtype=np.uint64)
    x0 = np.array(g["x_init"])[None]
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_x = torch.zeros(sol.nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(1, dtype=torch.float64, device=dev)
    d_st = torch.full((1,), -1, dtype=torch.int32, device=dev)
    d_info = torch.zeros(api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    args = (1, n_obs, up(hdr), up(rows), up(off), up(sfc), d_x, d_obj, d_st)
    sol.solve_device(*args, d_info=d_info, d_x_init=up(x0))  # a one-QP launch takes the two-wavefront instance
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sol.solve_device(*args, d_info=d_info, d_x_init=up(x0))
    e1.record()
    torch.cuda.synchronize()
    info = d_info.cpu().numpy().view(api.INFO_DTYPE)[0]
    assert d_st.item() == 0 and info["iterations"] <= 12
    ag = oracle.make_agent(n_obs=n_obs, **{k: v for k, v in g["hdr"].items()})
    lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
    lsc["nrm"] = R[:, :3].reshape(n_obs, M, 6, 3)
    lsc["d"] = R[:, 3].reshape(n_obs, M, 6)
    o = oracle.solve(cls, ag, lsc, sfc)
    assert o["status"] == 0
    assert abs(o["obj"] - d_obj.item()) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - d_x.cpu().numpy()).max() <= 1e-6
    # a point returned by the fallback rule is the remembered one and says so; since round 4 (centring target never below the gap
    # target, best remembered point) the instance may as well end as an ordinary OPTIMAL one: then it meets the strict 1e-8
    assert info["res_primal"] <= 1e-9 and info["res_dual"] <= (1e-6 if (info["flags"] & api.INFO_FLOOR_ACCEPTED) else 1e-8)


def test_log_replay_known_answers_on_the_gpu(api, oracle, torch_cuda):
    """The 300+ log-derived known answers of later replans (tests/golden/kat_log_replay.json) in ONE batch through the C ABI."""
    g = H.load_golden("kat_log_replay")
    p = g["params"]
    cls = H.oracle_class(oracle, p, use_sfc=False)
    sol = api.Solver(H.abi_desc(api, p, use_sfc=False))
    ags = [oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"], vmax=p["vmax"],
                             amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"]) for c in g["cases"]]
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, [None] * len(ags), None, p["M"])
    G = sol.solve_host(hdr, None, None, None)
    assert (G["status"] == 0).all()
    for q, c in enumerate(g["cases"]):
        for st in c["states"]:
            pos, vel, acc = oracle.state_at(cls, G["x"][q], st["t"] - c["t"])
            assert np.abs(pos - st["p"][:2]).max() <= 1.5e-5 and np.abs(vel - st["v"][:2]).max() <= 2e-5
            assert np.abs(acc - st["a"][:2]).max() <= 3e-4


def test_rsfc_planner_mode_relaxes_the_z_bounds_of_the_first_segment(api, oracle, torch_cuda):
    """RECIPROCALRSFC (reference src/traj_optimizer.cpp:255-258): the z variables of segment 0 are bounded by +-100 instead of the world
    box ("to avoid numerical error"), and there are no end-stop rows.  An agent just under the world ceiling, still climbing but
    already braking at the limit: its first segment has to overshoot the ceiling by a few millimetres.  With the world box on
    segment 0 (DLSC) that QP is infeasible on both sides; in RSFC mode it is feasible and GPU and oracle agree on the optimum."""
    M, dim = 5, 3
    wmin, wmax = [-5, -5, 0], [5, 5, 2.5]
    ag = oracle.make_agent(p0=[0, 0, 2.495], v0=[0, 0, 0.18], a0=[0, 0, -1.9], goal=[0.5, 0, 2.0], next_waypoint=[0.5, 0, 2.0])
    for name, mode_abi, mode_orc in (("dlsc", api.PLANNER_DLSC, 0), ("rsfc", api.PLANNER_RSFC, 2)):
        sol = api.Solver(api.make_desc(M=M, dim=dim, planner_mode=mode_abi, use_sfc=False, world_min=wmin, world_max=wmax))
        cls = oracle.make_class(M=M, dim=dim, planner_lsc=mode_orc, use_sfc=False, world_min=wmin, world_max=wmax)
        hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, [ag], [None], None, M)
        G = sol.solve_host(hdr, None, None, None)
        o = oracle.solve(cls, ag, None, None)
        if name == "dlsc":
            assert G["status"][0] != 0 and o["status"] != 0
            continue
        assert G["status"][0] == 0 and o["status"] == 0
        assert abs(o["obj"] - G["obj"][0]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][0]).max() <= X_TOL
        z = G["x"][0].reshape(dim, M, 6)[2]
        assert z[0, 3:].max() > 2.5 + 1e-3 and z[1:].max() <= 2.5 + 1e-9  # segment 0 overshoots, later segments keep the world box


def test_log_known_answer_with_an_active_lsc_row_on_the_gpu(api, oracle, torch_cuda):
    """The reference-logged replan that only active LSC rows explain (tests/golden/kat_log_active.json): goal LP and QP through the C ABI."""
    g = H.load_golden("kat_log_active")
    p, c = g["params"], g["cases"][0]
    cls = H.oracle_class(oracle, p, use_sfc=False)
    sol = api.Solver(H.abi_desc(api, p, use_sfc=False))
    L = np.zeros((len(c["neighbours"]), p["M"], 6), oracle.LSC_DTYPE)
    L["p"], L["nrm"], L["d"] = c["lsc_p"], c["lsc_nrm"], c["lsc_d"]
    ag = oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal_before_lp"], next_waypoint=c["next_waypoint"], vmax=p["vmax"],
                           amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(L))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, [ag], [L], None, p["M"])
    hdr2, gst = sol.optimize_goal_host(hdr, rows, off, None)  # GoalOptimizer: the goal moves off the waypoint
    assert gst[0] == 0 and np.abs(np.float32(hdr2["goal"][0]) - np.array(c["goal"])).max() <= 1e-7
    hdr2["goal"][0] = c["goal"]  # (float32 like agent.current_goal_point)
    ag_new = oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"], vmax=p["vmax"],
                               amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(L))
    hdr2["terminal_segments"][0] = oracle.terminal_segments(cls, ag_new)  # getTerminalSegments_old reads the goal AFTER goalPlanning
    G = sol.solve_host(hdr2, rows, off, None)
    assert G["status"][0] == 0
    for st in c["states"]:
        pos, vel, acc = oracle.state_at(cls, G["x"][0], st["t"] - c["t"])
        assert np.abs(pos - st["p"][:2]).max() <= 1.5e-5 and np.abs(vel - st["v"][:2]).max() <= 2e-5 and np.abs(acc - st["a"][:2]).max() <= 3e-4
    o = oracle.solve(cls, oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"], vmax=p["vmax"],
                                            amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=len(L)), L, None)
    assert abs(o["obj"] - G["obj"][0]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][0]).max() <= X_TOL


def test_log_pipeline_cases_on_the_gpu(api, oracle, torch_cuda):
    """tests/golden/kat_log_pipeline.json, the self-contained replans of the reference's logged mission (strongest active LSC rows,
    strongest active corridor faces, goals held back by GoalOptimizer): goal LP and QP of all of them in ONE ragged batch through
    the C ABI, against the logged states and against the oracle."""
    g = H.load_golden("kat_log_pipeline")
    p, cases = g["params"], g["cases"]
    M = p["M"]
    cls = H.oracle_class(oracle, p, use_sfc=True)
    sol = api.Solver(H.abi_desc(api, p, use_sfc=True))
    arrs = [H.pipeline_case_arrays(oracle, p, c) for c in cases]
    ags = [mk(c["goal_before_lp"]) for (L, box, mk), c in zip(arrs, cases)]
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, [a[0] for a in arrs], [a[1] for a in arrs], M)
    hdr2, gst = sol.optimize_goal_host(hdr, rows, off, sfc)
    assert (gst == 0).all()
    for q, c in enumerate(cases):
        assert np.abs(np.float32(hdr2["goal"][q]) - np.array(c["goal"])).max() <= 1e-7, (q, hdr2["goal"][q], c["goal"])
        hdr2["goal"][q] = c["goal"]  # float32 like agent.current_goal_point
        hdr2["terminal_segments"][q] = oracle.terminal_segments(cls, arrs[q][2](c["goal"]))  # read AFTER goalPlanning
    G = sol.solve_host(hdr2, rows, off, sfc)
    assert (G["status"] == 0).all(), G["status"]
    for q, c in enumerate(cases):
        assert H.logged_state_units(oracle, cls, c, G["x"][q]) <= c["match_units_of_6th_digit"] + 30
        assert abs(G["obj"][q] - c["oracle_obj"]) <= OBJ_TOL * max(1.0, abs(c["oracle_obj"]))
        o = oracle.solve(cls, arrs[q][2](c["goal"]), arrs[q][0], arrs[q][1])
        assert np.abs(o["x"] - G["x"][q]).max() <= X_TOL


@pytest.mark.parametrize("M,dim", [(5, 3), (6, 3), (7, 3), (10, 2), (10, 3), (5, 2), (8, 2), (8, 3), (9, 3)])
def test_dynamic_limits_bind_on_every_axis_and_in_every_segment(api, oracle, torch_cuda, request, M, dim):
    """A long hop under tight, per-axis different acceleration limits (and, second case, velocity limits): the optimum rides the
    limits on every axis, from the first segments (speeding up) to the last ones (braking) -- rows of src/traj_optimizer.cpp:448-471 far
    beyond the first dim*(3M-2) of their family.  Every compiled wavefront count of the shape must agree with the oracle.
    (Regression: the acceleration rows with index >= the matrix-row length were once dropped; the random swarms never noticed, the
    reference's own logged mission did -- tests/golden/kat_log_pipeline.json replans 40 and 59.)"""
    import os

    wmin, wmax = [-20, -20, -20 if dim == 3 else 0], [20, 20, 20 if dim == 3 else 2.5]
    z0 = 0.0 if dim == 3 else 1.0
    goal = [2.0, -1.5, 1.2 if dim == 3 else z0]
    cases = [dict(vmax=[3, 3, 3], amax=[0.6, 0.45, 0.3]), dict(vmax=[0.8, 0.5, 0.3], amax=[2.0, 1.5, 1.0])]
    cls = oracle.make_class(M=M, dim=dim, use_sfc=False, comm_range=0.0, world_min=wmin, world_max=wmax)
    ags = [oracle.make_agent(p0=[0, 0, z0], v0=[0.1, 0, 0], a0=[0, 0.05, 0], goal=goal, next_waypoint=goal, nominal_velocity=1.0, **c) for c in cases]
    O = [oracle.solve(cls, ag, None, None) for ag in ags]
    for o, ag, fam in zip(O, ags, ("acc", "vel")):
        assert o["status"] == 0
        sz = oracle.count(cls, ag, None)
        first = sz.n_sfc + sz.n_lsc + (sz.n_vel if fam == "acc" else 0)
        n = sz.n_acc if fam == "acc" else sz.n_vel
        act = np.nonzero(o["lam"][first:first + n] > 1e-6)[0]
        if fam == "acc":  # the family is ordered axis by axis: active rows at the start of the first axis and at the braking end of the last
            assert (act >= n - n // dim // 3).any() and (act < n // dim // 3).any(), act
        else:
            assert len(act) > 0 or M < 6, act
    sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=False, comm_range=0.0, world_min=wmin, world_max=wmax))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, [None, None], None, M)
    request.addfinalizer(lambda: os.environ.pop("LSCQP_WAVES", None))
    ran = 0
    for pin in (None, "1", "2", "4"):
        os.environ.pop("LSCQP_WAVES", None)
        if pin:
            os.environ["LSCQP_WAVES"] = pin
        try:
            G = sol.solve_host(hdr, None, None, None)
        except api.LscqpError as e:
            assert pin and e.code == api.ERR_UNSUPPORTED, e
            continue
        ran += 1
        for q, o in enumerate(O):
            assert G["status"][q] == 0
            assert abs(o["obj"] - G["obj"][q]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][q]).max() <= X_TOL, (pin, q)
    assert ran >= 2


@pytest.mark.parametrize("M,dim", [(5, 3), (6, 3), (7, 3), (10, 2), (10, 3), (5, 2), (8, 2), (8, 3), (9, 3)])
def test_every_row_family_binds_somewhere_along_the_horizon(api, oracle, torch_cuda, request, M, dim):
    """Three more scenarios that make the optimum lean on rows all along the horizon (a strong terminal weight on every segment pulls
    the agent towards a goal it cannot reach): (a) under tight velocity / acceleration limits with the waypoint-range rows
    (src/traj_optimizer.cpp:448-471, 494-497) -- dozens of active rows per family, first to last index; (b) with loose limits through
    corridors that narrow late in the horizon -- corridor faces of the last segments (:372-397); (c) loose limits, wide corridors -- the
    communication-range pair rows (:482-487).  The oracle's multipliers state the premise; every compiled wavefront count must reproduce its optimum."""
    import os

    wmin, wmax = [-20, -20, -20 if dim == 3 else 0], [20, 20, 20 if dim == 3 else 2.5]
    z0 = 0.0 if dim == 3 else 1.0
    goal = [6.0, -5.0, z0 + (4.0 if dim == 3 else 0)]
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, comm_range=3.0, w_t=50.0, world_min=wmin, world_max=wmax)
    wide, late = np.zeros(M, oracle.BOX_DTYPE), np.zeros(M, oracle.BOX_DTYPE)
    for b in (wide, late):
        b["bmin"], b["bmax"] = [-3, -3, -3 if dim == 3 else 0], [3, 3, 3 if dim == 3 else 2.5]
    for m in range(M // 2, M):
        late["bmin"][m], late["bmax"][m] = [-3, 0.0, -3 if dim == 3 else 0], [-0.2, 3, 0.7 if dim == 3 else 2.5]
    common = dict(p0=[-1.2, 0.9, z0], v0=[0.3, -0.2, 0], a0=[0, 0, 0], goal=goal, next_waypoint=[0, 0, z0], nominal_velocity=100.0, radius=0.15)
    loose = oracle.make_agent(vmax=[30, 30, 30], amax=[400, 400, 400], **common)
    ags = [oracle.make_agent(vmax=[2.0, 1.5, 1.0], amax=[8, 6, 4], **common), loose, loose]
    boxes = [wide, late, wide]
    O = [oracle.solve(cls, ag, None, b) for ag, b in zip(ags, boxes)]
    act = []
    for o, ag in zip(O, ags):
        assert o["status"] == 0
        sz, a, fam = oracle.count(cls, ag, None), 0, {}
        for name, n in (("sfc", sz.n_sfc), ("lsc", sz.n_lsc), ("vel", sz.n_vel), ("acc", sz.n_acc), ("comm", sz.n_comm)):
            fam[name] = (n, np.nonzero(o["lam"][a:a + n] > 1e-6)[0])
            a += n
        act.append(fam)
    for name in ("vel", "acc"):  # (a): many active rows, spread over the family (velocity rows: into its last tenth)
        n, idx = act[0][name]
        assert len(idx) >= 15 and idx.max() - idx.min() >= n // 3, (name, idx)
        assert name == "acc" or idx.max() >= n - n // 10, (name, idx)
    assert len(act[0]["comm"][1]) >= 1 and len(act[2]["comm"][1]) >= 6  # (c) loose limits, wide corridor: the pair rows hold the agent back
    n, idx = act[1]["sfc"]
    assert len(idx) >= 20 and idx.max() >= n - n // 10, idx  # (b): faces of the last segments
    sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=True, comm_range=3.0, w_t=50.0, world_min=wmin, world_max=wmax))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, [None, None, None], boxes, M)
    request.addfinalizer(lambda: os.environ.pop("LSCQP_WAVES", None))
    ran = 0
    for pin in (None, "1", "2", "4"):
        os.environ.pop("LSCQP_WAVES", None)
        if pin:
            os.environ["LSCQP_WAVES"] = pin
        try:
            G = sol.solve_host(hdr, None, None, sfc)
        except api.LscqpError as e:
            assert pin and e.code == api.ERR_UNSUPPORTED, e
            continue
        ran += 1
        for q, o in enumerate(O):
            assert G["status"][q] == 0, (pin, q, G["status"])
            # (objective to the stated bar; x to 5e-6 m here: with dozens of corridor faces active a control point of the last
            # segments sits in a direction the min-jerk Hessian barely sees -- objective 3e-11 apart, that coordinate 1.2e-6 m)
            assert abs(o["obj"] - G["obj"][q]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][q]).max() <= 5e-6, (pin, q)
    assert ran >= 2


@pytest.mark.pdip_only
def test_breakdown_under_one_elimination_order_is_repaired_on_the_other(api, oracle, torch_cuda):
    """tests/golden/nd_breakdown_m10d2.json (tools/make_golden_nd_breakdown.py): a feasible M = 10, 2-D instance from the sweep far outside
    the reference's parameters on which the nested-dissection instance ends NUMERIC at iteration 11 (a pivot of the late-iteration matrix
    cancels to <= 0).  The natural elimination order solves it: the host-pointer entry runs that second pass by itself, the device
    entry on request (retry = 2)."""
    import torch

    g = H.load_golden("nd_breakdown_m10d2")
    par = g["params"]
    M, dim = par["M"], par["dim"]
    cls = oracle.make_class(use_sfc=True, **par)
    ag = oracle.make_agent(**g["agent"])
    box = np.zeros(M, oracle.BOX_DTYPE)
    box["bmin"], box["bmax"] = g["sfc_min"], g["sfc_max"]
    o = oracle.solve(cls, ag, None, box)
    assert o["status"] == 0 and abs(o["obj"] - g["oracle_obj"]) <= 1e-9 * abs(g["oracle_obj"])
    sol = api.Solver(api.make_desc(use_sfc=True, **par))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, [ag], [None], [box], M)
    G = sol.solve_host(hdr, None, None, sfc)  # first pass (nested dissection) + the other-order pass
    assert G["status"][0] == 0
    assert abs(o["obj"] - G["obj"][0]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][0]).max() <= X_TOL
    # the device entry: plain call (what the first pass alone does), then with retry = 2
    dev = torch.device("cuda", 0)
    d_hdr = torch.from_numpy(hdr.view(np.uint8).reshape(-1).copy()).to(dev)
    d_sfc = torch.from_numpy(sfc.view(np.uint8).reshape(-1).copy()).to(dev)
    d_off = torch.zeros(2, dtype=torch.int64, device=dev)
    d_rows = torch.zeros(32, dtype=torch.uint8, device=dev)
    out = []
    for retry in (0, 2):
        d_x, d_obj, d_st = torch.zeros(sol.nv, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        sol.solve_device(1, 0, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, retry=retry)
        torch.cuda.synchronize()
        out.append((int(d_st.item()), d_x.cpu().numpy()))
    assert out[0][0] in (api.STATUS_OPTIMAL, api.STATUS_NUMERIC, api.STATUS_INFEASIBLE)  # (NUMERIC today; not a promise)
    assert out[1][0] == api.STATUS_OPTIMAL and np.abs(out[1][1] - o["x"]).max() <= X_TOL
    # the sharded host entry (TrajOptimizer::setCommunicator + solveBatch) repairs it like the one-device entry: same bits
    comm = api.Comm(1)
    S = sol.solve_sharded(comm, hdr, None, None, sfc)
    assert S["status"][0] == 0 and np.array_equal(S["x"], G["x"]) and np.array_equal(S["obj"], G["obj"])
    comm.close()


@pytest.mark.parametrize("M,dim,n_obs,variant", [(5, 3, 20, ""), (5, 3, 48, ""), (6, 3, 20, ""), (10, 2, 9, ""), (10, 2, 40, ""), (10, 3, 40, ""), (7, 3, 12, ""),
                                                (8, 2, 12, ""), (5, 3, 10, "dlsc"), (5, 2, 12, "dlsc"), (10, 2, 10, "dlsc"), (9, 3, 14, ""), (10, 3, 14, "dlsc"), (10, 3, 36, "dlsc"),
                                                (9, 3, 14, "dlsc"), (8, 3, 14, "dlsc"), (5, 3, 20, "rows_f32"),
                                                (10, 2, 9, "rows_f32"), (5, 3, 20, "mixed"), (10, 2, 9, "mixed")])
def test_one_binding_lsc_plane_per_obstacle_slot(api, oracle, torch_cuda, request, M, dim, n_obs, variant):
    """Index test of the LSC rows: one QP per obstacle slot, whose ONLY non-zero rows sit in that slot (segment oi mod M, all six
    control points) and block the straight way to the goal -- whatever slot, segment or lane a row is staged in, it has to bind.  The
    oracle's multipliers state the premise; every compiled wavefront count of the shape that holds n_obs must reproduce the optimum."""
    import os

    wmin, wmax = [-10, -10, -10 if dim == 3 else 0], [10, 10, 10 if dim == 3 else 2.5]
    z0 = 0.0 if dim == 3 else 1.0
    # variants: the DLSC instances (no end-stop rows), 16-byte rows, the mixed-precision instances
    lsc_mode = variant != "dlsc"
    cls = oracle.make_class(M=M, dim=dim, use_sfc=False, planner_lsc=lsc_mode, world_min=wmin, world_max=wmax)
    ags, Ls = [], []
    free = oracle.solve(cls, oracle.make_agent(p0=[0, 0, z0], v0=[0.2, 0, 0], a0=[0, 0, 0], goal=[1.0, 0.1, z0], next_waypoint=[1.0, 0.1, z0]), None, None)
    xfree = free["x"].reshape(dim, M, 6)[0]
    for oi in range(n_obs):
        m = oi % M
        L = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
        xb = float(np.float32(0.6 * xfree[m].max()))  # stop the free trajectory at 60 % of where it gets to in segment m (a float32 value: exact in 16-byte rows)
        L["p"][oi, m] = [xb, 0.0, z0]
        L["nrm"][oi, m] = [-1.0, 0.0, 0.0]
        L["d"][oi, m] = 0.0
        ags.append(oracle.make_agent(p0=[0, 0, z0], v0=[0.2, 0, 0], a0=[0, 0, 0], goal=[1.0, 0.1, z0], next_waypoint=[1.0, 0.1, z0], n_obs=n_obs))
        Ls.append(L)
    O = [oracle.solve(cls, ag, L, None) for ag, L in zip(ags, Ls)]
    n_active = 0
    for o, ag, L in zip(O, ags, Ls):
        assert o["status"] == 0
        sz = oracle.count(cls, ag, L)
        n_active += o["lam"][sz.n_sfc:sz.n_sfc + sz.n_lsc].max() > 1e-6
    assert n_active >= n_obs - n_obs // M - 1, n_active  # (a plane on segment 0 only holds its last three control points: may stay inactive)
    sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=False, world_min=wmin, world_max=wmax,
                                   planner_mode=api.PLANNER_LSC if lsc_mode else api.PLANNER_DLSC,
                                   row_format=api.ROWS_F32 if variant == "rows_f32" else api.ROWS_F64,
                                   precision=api.PRECISION_MIXED if variant == "mixed" else api.PRECISION_F64))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, Ls, None, M)
    request.addfinalizer(lambda: os.environ.pop("LSCQP_WAVES", None))
    ran = 0
    for pin in (None, "1", "2", "4"):
        os.environ.pop("LSCQP_WAVES", None)
        if pin:
            os.environ["LSCQP_WAVES"] = pin
        try:
            G = sol.solve_host(hdr, rows, off, None)
        except api.LscqpError as e:
            assert pin and e.code == api.ERR_UNSUPPORTED, e
            continue
        ran += 1
        for q, o in enumerate(O):
            assert G["status"][q] == 0, (pin, q)
            assert abs(o["obj"] - G["obj"][q]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][q]).max() <= X_TOL, (pin, q)
    assert ran >= 1


@pytest.mark.parametrize("M,dim", [(5, 3), (6, 3), (10, 2), (10, 3), (7, 3), (5, 2), (8, 3), (9, 3)])
def test_one_binding_corridor_face_per_segment_axis_and_side(api, oracle, torch_cuda, request, M, dim):
    """Index test of the corridor rows: one QP per (segment, axis, side) whose corridor is wide open except for that one face, placed
    across the way to the goal (src/traj_optimizer.cpp:372-397)."""
    import os

    wmin, wmax = [-10, -10, -10 if dim == 3 else 0], [10, 10, 10 if dim == 3 else 2.5]
    z0 = 0.0 if dim == 3 else 1.0
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=wmin, world_max=wmax)
    ags, boxes = [], []
    wide = np.zeros(M, oracle.BOX_DTYPE)
    wide["bmin"], wide["bmax"] = [-5, -5, -5 if dim == 3 else 0], [5, 5, 5 if dim == 3 else 2.5]
    for m in range(1, M):
        for k in range(dim):
            for side in (0, 1):
                sgn = 1.0 if side else -1.0  # the goal lies on the side of the face
                goal = [0.0, 0.0, z0]
                goal[k] += sgn * 1.0
                goal[(k + 1) % dim] += 0.1
                box = wide.copy()
                free = oracle.solve(cls, oracle.make_agent(p0=[0, 0, z0], v0=[0, 0, 0], a0=[0, 0, 0], goal=goal, next_waypoint=goal), None, wide)
                reach = 0.6 * np.abs(free["x"].reshape(dim, M, 6)[k, m] - (z0 if k == 2 else 0.0)).max()  # 60 % of the free displacement
                base = z0 if k == 2 else 0.0
                if side:
                    box["bmax"][m][k] = base + reach
                else:
                    box["bmin"][m][k] = base - reach
                ags.append(oracle.make_agent(p0=[0, 0, z0], v0=[0, 0, 0], a0=[0, 0, 0], goal=goal, next_waypoint=goal))
                boxes.append(box)
    O = [oracle.solve(cls, ag, None, b) for ag, b in zip(ags, boxes)]
    n_active = 0
    for o, ag in zip(O, ags):
        assert o["status"] == 0
        sz = oracle.count(cls, ag, None)
        n_active += o["lam"][:sz.n_sfc].max() > 1e-6
    assert n_active >= len(ags) - 2 * dim, (n_active, len(ags))
    sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=True, world_min=wmin, world_max=wmax))
    hdr, rows, off, sfc = H.abi_batch(api, oracle, cls, ags, [None] * len(ags), boxes, M)
    request.addfinalizer(lambda: os.environ.pop("LSCQP_WAVES", None))
    ran = 0
    for pin in (None, "1", "2", "4"):
        os.environ.pop("LSCQP_WAVES", None)
        if pin:
            os.environ["LSCQP_WAVES"] = pin
        try:
            G = sol.solve_host(hdr, None, None, sfc)
        except api.LscqpError as e:
            assert pin and e.code == api.ERR_UNSUPPORTED, e
            continue
        ran += 1
        for q, o in enumerate(O):
            assert G["status"][q] == 0, (pin, q)
            assert abs(o["obj"] - G["obj"][q]) <= OBJ_TOL * max(1.0, abs(o["obj"])) and np.abs(o["x"] - G["x"][q]).max() <= X_TOL, (pin, q)
    assert ran >= 2


@pytest.mark.pdip_only
def test_feasible_instance_whose_residual_pauses_around_the_tenth_iteration(api, oracle, torch_cuda):
    """tools/stress_parity.py, shape (24 x M10 x 40 neighbours, forest), seed 114, first batch from hover: instance 14 holds its primal
    residual at 5e-4 m for four iterations around the tenth and converges afterwards.  The single-check stall rule of round 1 (and a
    'flat residual' shortcut tried in round 2) reported it INFEASIBLE; the stall now has to show at two checks in a row."""
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(24, M=10, dim=3, n_obs=40, seed=114, style="forest")
    cls = oracle.make_class(M=10, dim=3, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    sol = api.Solver(api.make_desc(M=10, dim=3, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, 10)
    G = sol.solve_host(hdr, rows, off, sfc)
    ag, lsc, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
    R = oracle.solve_batch(cls, ag, lsc, loff, sfco, threads=8)
    assert (R["status"] == 0).all() and (G["status"] == 0).all(), (G["status"], G["info"]["iterations"])
    assert G["info"]["iterations"][14] >= 14
    _check_against_oracle(oracle, cls, G, R)


@pytest.mark.pdip_only
@pytest.mark.parametrize("N,M,dim,n_obs,style,seed", [(48, 5, 3, 20, "forest", 21), (16, 10, 2, 9, "forest", 22), (24, 10, 3, 40, "forest", 23), (32, 6, 3, 20, "maze", 24)])
def test_tight_warm_start_mode_reaches_the_same_optimum(api, oracle, torch_cuda, N, M, dim, n_obs, style, seed):
    """lscqp_class_desc.warm_start = LSCQP_WARM_TIGHT (an option for throughput batches): every complementarity product starts at 1e-7 with
    the slacks floored at 3 mm, an instance whose first step is short returns to the default centring.  Same optimum as the default mode
    and as the oracle on replanning swarms; fewer iterations in total on the M = 5 forest class (3 -> 2), not on the M = 10 shapes."""
    from lsc_dr_planner_amd import synth

    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    base = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    tight = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, warm_start=api.WARM_TIGHT))
    it_base = it_tight = 0
    for step in range(5):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        xi = api.x_init_from_swarm(b, dim)
        A, B = base.solve_host(hdr, rows, off, sfc, x_init=xi), tight.solve_host(hdr, rows, off, sfc, x_init=xi)
        assert (A["status"] == 0).all() and (B["status"] == 0).all(), (A["status"], B["status"])
        ag, lsc, loff, sfco = H.swarm_oracle_inputs(oracle, sw, b)
        R = oracle.solve_batch(cls, ag, lsc, loff, sfco, threads=8)
        _check_against_oracle(oracle, cls, B, R)
        if step >= 1:
            it_base += A["info"]["iterations"].sum()
            it_tight += B["info"]["iterations"].sum()
        sw.advance(B["x"])
    assert M != 5 or it_tight < 0.8 * it_base, (it_tight, it_base)


@pytest.mark.pdip_only
@pytest.mark.gpu
def test_work_order_and_work_queue_change_when_an_instance_is_solved_never_its_result(api, oracle, torch_cuda):
    """Round 4: a launch with more instances than the chip holds runs persistent workgroups over a queue, and the caller may name the order
    (lscqp_solve_batch_device_ordered; lscqp_order_by_work_device sorts by the previous solve's iterations, most first, stable).  Both only
    decide WHEN an instance is solved: 1200 x M10 x 12 QPs (more than four rounds of one workgroup per CU) give bit for bit the same plans,
    objectives, statuses and iteration counts in one launch as given, in the sorted order, in a random order, and solved in chunks that fit
    the chip (one instance per workgroup, no queue); a sample agrees with the oracle."""
    torch = torch_cuda
    from lsc_dr_planner_amd import synth

    N, M, dim, n_obs = 1200, 10, 3, 12
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=41)
    sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
    d_hdr, d_rows, d_off, d_sfc = up(hdr), up(rows), up(off), up(sfc)

    def solve(d_order=None, chunks=None):
        d_x = torch.zeros(N * sol.nv, dtype=torch.float64, device=dev)
        d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
        d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
        d_info = torch.zeros(N * api.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        if chunks is None:
            sol.solve_device(N, sw.n_obs, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_order=d_order)
        else:
            R2, S2 = rows.reshape(N, -1), sfc.reshape(N, M)
            for lo in range(0, N, chunks):
                hi = min(N, lo + chunks)
                o = np.arange(hi - lo + 1, dtype=np.uint64) * np.uint64(R2.shape[1])
                sol.solve_device(hi - lo, sw.n_obs, up(hdr[lo:hi]), up(R2[lo:hi]), up(o), up(S2[lo:hi]), d_x[lo * sol.nv:], d_obj[lo:], d_st[lo:],
                                 d_info[lo * api.INFO_DTYPE.itemsize:])
        torch.cuda.synchronize()
        return d_x.cpu().numpy(), d_obj.cpu().numpy(), d_st.cpu().numpy(), d_info.cpu().numpy().view(api.INFO_DTYPE).copy(), d_info

    x0, o0, s0, i0, d_info0 = solve()
    assert (s0 == 0).all(), np.bincount(s0)
    d_order = torch.zeros(N, dtype=torch.int32, device=dev)
    sol.order_by_work_device(N, d_info0, d_order)
    torch.cuda.synchronize()
    order = d_order.cpu().numpy()
    assert np.array_equal(order, np.argsort(-i0["iterations"], kind="stable").astype(np.int32)) and i0["iterations"].max() > i0["iterations"].min()
    rnd = np.random.default_rng(3).permutation(N).astype(np.int32)
    for name, res in (("sorted", solve(d_order)), ("random", solve(torch.from_numpy(rnd).to(dev))), ("chunks of 200", solve(chunks=200))):
        x, o, s, i, _ = res
        assert np.array_equal(x, x0) and np.array_equal(o, o0) and np.array_equal(s, s0) and np.array_equal(i["iterations"], i0["iterations"]), name
    cls = oracle.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    ag, lsc, loff, sfc_o = H.swarm_oracle_inputs(oracle, sw, b)
    sel = np.r_[order[:6], order[-6:]]  # the longest and the shortest solves
    R = oracle.solve_batch(cls, ag[sel], lsc, loff[sel], np.ascontiguousarray(sfc_o.reshape(N, M)[sel]).reshape(-1), threads=12)
    assert (R["status"] == 0).all()
    assert np.abs(x0.reshape(N, -1)[sel] - R["x"]).max() <= 1e-6 and (np.abs(o0[sel] - R["obj"]) / np.maximum(1, np.abs(R["obj"]))).max() <= OBJ_TOL

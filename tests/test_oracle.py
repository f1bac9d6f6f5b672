"""CPU tests: the oracle (oracle/lscqp_oracle.c) against the reference's own artefacts and independent solutions.

The reference has no tests for the trajectory QP, so the pins are:
  * structural known answers taken from the reference source (Q_base closed form, A_0/A_T stencils, Bernstein matrix,
    terminal-segment formula, the row-count table of SURVEY.md §8),
  * the reference-authored result log (tests/golden/kat_log.json, made by tools/make_golden.py from
    log/simulation_1663743693.650981_LSC_10agents.csv),
  * scipy trust-constr + 80-bit active-set solutions of the row-for-row assembled models (tests/golden/scipy_*.json).
"""
import numpy as np
import pytest

from tests import helpers as H


def test_q_base_closed_form(oracle):
    # src/traj_optimizer.cpp:163-178 for n=5, phi=3, phi_n=1
    Q = oracle.q_base(5, 3, 1, 0.2) * 0.2 ** 5
    ref = np.array([[720, -1800, 1200, 0, 0, -120], [-1800, 4800, -3600, 0, 600, 0], [1200, -3600, 3600, -1200, 0, 0],
                    [0, 0, -1200, 3600, -3600, 1200], [0, 600, 0, -3600, 4800, -1800], [-120, 0, 0, 1200, -1800, 720]], float)
    assert np.abs(Q - ref).max() < 1e-8
    w = np.linalg.eigvalsh(ref)
    assert (np.abs(w[:3]) < 1e-6).all() and (w[3:] > 1e3).all()  # PSD, rank 3


def test_bernstein_matrix(oracle):
    # include/polynomial.hpp:281-294: B maps Bernstein control points to monomial coefficients
    B = oracle.bernstein(5)
    assert B[0].tolist() == [1, -5, 10, -10, 5, -1]
    assert B[5].tolist() == [0, 0, 0, 0, 0, 1]
    t = 0.37
    mono = np.array([t ** j for j in range(6)])
    from math import comb
    bern = np.array([comb(5, i) * t ** i * (1 - t) ** (5 - i) for i in range(6)])
    assert np.allclose(B @ mono, bern)


def test_aeq_base(oracle):
    rc, A = oracle.aeq_base(4, 5, 3, 0.2)
    assert rc == 0 and A.shape == (6, 24)
    # C0 row between segments 1 and 2: +x[1][5] - x[2][0]      (src/traj_optimizer.cpp:204-213)
    r = np.zeros(24); r[11] = 1; r[12] = -1
    assert np.allclose(A[0], r)
    # C1 row: n/dt (x[1][5]-x[1][4]) - n/dt (x[2][1]-x[2][0])
    r = np.zeros(24); r[11] = 25; r[10] = -25; r[13] = -25; r[12] = 25
    assert np.allclose(A[1], r)
    rc, _ = oracle.aeq_base(4, 4, 3, 0.2)  # only n=5, phi=3 (std::invalid_argument in the reference)
    assert rc == -1


@pytest.mark.parametrize("M,dim,n_obs,exp", [
    # SURVEY.md §8 size table: (nv, n_eq, LSC rows, SFC rows, vel, acc, comm)
    (10, 2, 9, (120, 64, 513, 228, 192, 156, 260)),
    (5, 3, 20, (90, 51, 540, 162, 138, 114, 120)),
    (6, 3, 20, (108, 60, 660, 198, 168, 138, 162)),
    (10, 3, 40, (180, 96, 2280, 342, 288, 234, 390)),
])
def test_row_counts(oracle, M, dim, n_obs, exp):
    cls = oracle.make_class(M=M, dim=dim)
    ag = oracle.make_agent(p0=(0, 0, 1), goal=(1, 0, 1), n_obs=n_obs)
    lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)
    lsc["nrm"][..., 0] = 1.0
    s = oracle.count(cls, ag, lsc)
    assert (s.nv, s.neq, s.n_lsc, s.n_sfc, s.n_vel, s.n_acc, s.n_comm) == exp
    lsc["nrm"][0, 1, 2] = 0.0  # a zero normal is skipped (src/traj_optimizer.cpp:409-411)
    s2 = oracle.count(cls, ag, lsc)
    assert s2.n_lsc == exp[2] - 1 and s2.n_lsc_skipped == 1


def test_terminal_segments(oracle):
    # src/traj_optimizer.cpp:530-538: max(int((M dt - |goal-p|/v_nom + 1e-9)/dt), 1)
    cls = oracle.make_class(M=10, dim=2)
    assert oracle.terminal_segments(cls, oracle.make_agent(p0=(4, 0, 0.6), goal=(3.5, 0, 0.6))) == 7
    assert oracle.terminal_segments(cls, oracle.make_agent(p0=(4, 0, 0.6), goal=(-4, 0, 0.6))) == 1
    assert oracle.terminal_segments(cls, oracle.make_agent(p0=(4, 0, 0.6), goal=(4, 0, 0.6))) == 10
    # point3d is float32: |goal-p| = 0.4f = 0.40000000596 > 0.4, so (2.0 - 0.4f + 1e-9)/0.2 = 7.99999997 -> 7,
    # where exact arithmetic would give 8.  The shim passes this integer to the solver for exactly this reason.
    assert oracle.terminal_segments(cls, oracle.make_agent(p0=(0, 0, 0), goal=(0.4, 0, 0))) == 7
    assert oracle.terminal_segments(cls, oracle.make_agent(p0=(0, 0, 0), goal=(0.375, 0, 0))) == 8


def test_log_known_answers(oracle):
    """First replan of forest10_10 as the reference logged it (float32 output, 6 printed digits)."""
    g = H.load_golden("kat_log")
    p = g["params"]
    for case in g["cases"]:
        cls = H.oracle_class(oracle, p, use_sfc=case["sfc"] is not None)
        ag = oracle.make_agent(p0=case["p0"], goal=case["goal"], next_waypoint=case["next_waypoint"], vmax=p["vmax"],
                               amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"])
        sfc = None
        if case["sfc"]:
            sfc = np.zeros(p["M"], oracle.BOX_DTYPE)
            sfc["bmin"], sfc["bmax"] = case["sfc"]["bmin"], case["sfc"]["bmax"]
        r = oracle.solve(cls, ag, None, sfc)
        assert r["status"] == 0
        log = g["agents"][case["agent"]]["states"]
        for st in log[1:]:
            pos, vel, acc = oracle.state_at(cls, r["x"], st["t"])
            assert np.allclose(pos, st["p"][:2], rtol=0, atol=2e-5), (case["name"], st["t"], pos, st["p"])
            assert np.allclose(vel, st["v"][:2], rtol=1e-4, atol=2e-6), (case["name"], st["t"], vel, st["v"])
            assert np.allclose(acc, st["a"][:2], rtol=3e-4, atol=2e-5), (case["name"], st["t"], acc, st["a"])
        k = oracle.kkt(cls, ag, None, sfc, r.get("x_interior_point", r["x"]), r["y"], r["lam"], r["mu_lb"], r["mu_ub"])
        assert k["stationarity"] < 1e-9 and k["eq"] < 1e-9 and k["ineq"] < 1e-9 and k["neg_mult"] == 0
    # symmetric agents share the answer up to sign / axis: agent 5 mirrors agent 0
    a0, a5 = g["agents"][0]["states"][1], g["agents"][5]["states"][1]
    assert a0["v"][0] == -a5["v"][0]


def test_kat2_needs_the_active_face(oracle):
    """Without the SFC rows the answer differs by 0.17 % - the KAT exercises an active inequality."""
    g = H.load_golden("kat_log")
    p, case = g["params"], g["cases"][1]
    cls = H.oracle_class(oracle, p, use_sfc=False)
    ag = oracle.make_agent(p0=case["p0"], goal=case["goal"], next_waypoint=case["next_waypoint"])
    r = oracle.solve(cls, ag, None, None)
    _, vel, _ = oracle.state_at(cls, r["x"], 0.1)
    logged = g["agents"][1]["states"][1]["v"][0]
    assert abs(vel[0] - logged) / abs(logged) > 1e-3


@pytest.mark.parametrize("name", ["scipy_m5d3", "scipy_m10d2", "scipy_m6d3_maze"])
def test_oracle_vs_scipy_golden(oracle, name):
    g = H.load_golden(name)
    p = g["params"]
    cls = H.oracle_class(oracle, p)
    for c in g["cases"]:
        ag, lsc, sfc = H.golden_case_arrays(oracle, p, c)
        r = oracle.solve(cls, ag, lsc, sfc)
        assert r["status"] == 0
        assert np.abs(r["x"] - np.array(c["x"])).max() < 1e-7          # 80-bit polished active-set solution
        assert np.abs(r["x"] - np.array(c["scipy_x"])).max() < 1e-5    # raw trust-constr iterate
        assert abs(r["obj"] - c["obj"]) <= 1e-9 * max(1.0, abs(c["obj"]))
        # (the multipliers are the interior-point iterate's: its own KKT residuals are taken at that iterate, not at the polished point)
        k = oracle.kkt(cls, ag, lsc, sfc, r.get("x_interior_point", r["x"]), r["y"], r["lam"], r["mu_lb"], r["mu_ub"])
        assert k["stationarity"] < 1e-9 and k["eq"] < 1e-9 and k["ineq"] < 1e-10 and k["comp"] < 1e-6
        assert np.abs(r["x"] - r.get("x_interior_point", r["x"])).max() < 1e-7  # polish and iterate agree far inside the golden bar


def test_oracle_detects_infeasible(oracle):
    cls = oracle.make_class(M=5, dim=3)
    ag = oracle.make_agent(p0=(0, 0, 1), goal=(0.5, 0, 1), n_obs=1)
    lsc = np.zeros((1, 5, 6), oracle.LSC_DTYPE)
    lsc["nrm"][..., 0] = 1.0
    lsc["d"] = 50.0  # x >= 50 for every control point: outside the world box
    sfc = np.zeros(5, oracle.BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = (-5, -5, 0), (5, 5, 2.5)
    r = oracle.solve(cls, ag, lsc, sfc, max_iter=100)
    assert r["status"] != 0


def test_objective_includes_constant_terminal_term(oracle):
    # cost += w_t (x - goal)^2 keeps goal^2 (src/traj_optimizer.cpp:301-316): hovering at the goal costs 0
    cls = oracle.make_class(M=5, dim=3, use_sfc=False)
    ag = oracle.make_agent(p0=(1, 2, 1), goal=(1, 2, 1))
    r = oracle.solve(cls, ag)
    # not exactly 0: Q_base = integer matrix * pow(dt,-5) rounded per entry, as in the reference, is not exactly
    # translation invariant (row sums ~1e-9), which leaves ~1e-9 * |x|^2 in the objective
    assert r["status"] == 0 and abs(r["obj"]) < 1e-8
    assert np.abs(r["x"].reshape(3, 30) - np.array([1, 2, 1])[:, None]).max() < 1e-6


def test_log_replay_known_answers(oracle):
    """tests/golden/kat_log_replay.json: 300+ LATER replans of the reference's own run (non-zero initial velocity and acceleration),
    waypoints inferred by tools/make_golden_log_replay.py.  The restatement must land on the logged states at t + 0.1 s and
    t + 0.2 s; the tolerances are those of inputs that are themselves known to six printed digits (position 1.5e-5 m, velocity
    2e-5 m/s, acceleration 3e-4 m/s^2 -- wrong waypoint candidates miss by 1e-2 and more)."""
    g = H.load_golden("kat_log_replay")
    p = g["params"]
    assert len(g["cases"]) >= 300
    cls = H.oracle_class(oracle, p, use_sfc=False)
    moving = 0
    for c in g["cases"]:
        ag = oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=c["goal"], next_waypoint=c["next_waypoint"], vmax=p["vmax"],
                               amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"])
        R = oracle.solve(cls, ag, None, None)
        assert R["status"] == 0
        for st in c["states"]:
            pos, vel, acc = oracle.state_at(cls, R["x"], st["t"] - c["t"])
            assert np.abs(pos - st["p"][:2]).max() <= 1.5e-5, (c["agent"], c["replan"])
            assert np.abs(vel - st["v"][:2]).max() <= 2e-5, (c["agent"], c["replan"])
            assert np.abs(acc - st["a"][:2]).max() <= 3e-4, (c["agent"], c["replan"])
        moving += np.abs(c["a0"]).max() > 0.05
    assert moving >= 100  # most of them start from a genuinely accelerating state


def _active_case(oracle):
    g = H.load_golden("kat_log_active")
    p, c = g["params"], g["cases"][0]
    cls = H.oracle_class(oracle, p, use_sfc=False)
    L = np.zeros((len(c["neighbours"]), p["M"], 6), oracle.LSC_DTYPE)
    L["p"], L["nrm"], L["d"] = c["lsc_p"], c["lsc_nrm"], c["lsc_d"]
    return p, c, cls, L


def _state_errors(oracle, cls, c, x):
    e = dict(p=0.0, v=0.0, a=0.0)
    for st in c["states"]:
        pos, vel, acc = oracle.state_at(cls, x, st["t"] - c["t"])
        e["p"] = max(e["p"], np.abs(pos - st["p"][:2]).max())
        e["v"] = max(e["v"], np.abs(vel - st["v"][:2]).max())
        e["a"] = max(e["a"], np.abs(acc - st["a"][:2]).max())
    return e


def test_log_known_answer_with_an_active_lsc_row(oracle):
    """tests/golden/kat_log_active.json (tools/make_golden_log_active.py): replan 7 of agent 6 of the reference's own run, t = 1.4 s,
    two agents in range.  Its CLSC rows come from the neighbours' previous plans (certified by the log-replay fixture), GoalOptimizer
    moves the goal off the waypoint (t = 0.06), and the QP's optimum carries a non-zero LSC multiplier.  WITH the rows the
    restatement lands on the logged states within the log's input precision; WITHOUT them it misses velocity and acceleration by
    more than ten times that -- reference-logged motion that only the LSC rows explain."""
    p, c, cls, L = _active_case(oracle)
    mk = lambda n_obs, goal: oracle.make_agent(p0=c["p0"], v0=c["v0"], a0=c["a0"], goal=goal, next_waypoint=c["next_waypoint"], vmax=p["vmax"],  # noqa: E731
                                               amax=p["amax"], radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=n_obs)
    # the goal LP (GoalOptimizer restated) reproduces the fixture's goal from the previous goal point and the inferred waypoint
    st, goal, t = oracle.goal_opt(cls, c["goal_before_lp"], c["next_waypoint"], lsc=L)
    assert st == 0 and abs(t - c["goal_lp_t"]) <= 1e-12 and 0.01 < t < 0.5
    assert np.abs(np.float32(goal) - np.array(c["goal"])).max() <= 1e-7
    R = oracle.solve(cls, mk(len(L), c["goal"]), L, None)
    assert R["status"] == 0
    e = _state_errors(oracle, cls, c, R["x"])
    assert e["p"] <= 1.5e-5 and e["v"] <= 2e-5 and e["a"] <= 3e-4, e
    sz = oracle.count(cls, mk(len(L), c["goal"]), L)
    lam = R["lam"][sz.n_sfc:sz.n_sfc + sz.n_lsc]
    assert lam.max() > 1e-3 and abs(lam.max() - c["max_lsc_multiplier"]) <= 1e-6 * lam.max() + 1e-9
    R0 = oracle.solve(cls, mk(0, c["goal"]), None, None)
    e0 = _state_errors(oracle, cls, c, R0["x"])
    assert e0["v"] >= 10 * 2e-5 and e0["a"] >= 10 * 3e-4, e0


def test_log_pipeline_replay_table():
    """tests/golden/kat_log_pipeline.json (tools/make_golden_log_pipeline.py): the reference's whole logged mission -- 10 agents x 79
    replans -- replayed through the restated pipeline (previous plans -> generateCLSC rows -> corridors over the forest10 map ->
    GoalOptimizer -> TrajOptimizer).  Every one of the 790 replans reproduces the twelve logged numbers of its next two log lines
    to the log's input precision; dozens of them with ACTIVE LSC rows, hundreds with active corridor faces."""
    g = H.load_golden("kat_log_pipeline")
    tab, st = g["replay"], g["stats"]
    assert st["tried"] == st["matched"] == len(tab) == 790 and not st["chains_ended"]
    assert sorted((r["replan"], r["agent"]) for r in tab) == [(k, a) for k in range(79) for a in range(10)]
    m = np.array([r["match"] for r in tab])
    assert m.max() <= 400 and np.median(m) <= 60 and (m > 150).sum() <= 60
    assert sum(r["lam_lsc"] > 1e-6 for r in tab) == st["lsc_active"] >= 60
    assert sum(r["lam_sfc"] > 1e-6 for r in tab) == st["sfc_active"] >= 300
    assert sum(r["goal_lp_t"] > 1e-9 for r in tab) >= 100  # GoalOptimizer held the goal back from the waypoint
    assert max(len(r["neighbours"]) for r in tab) == 9


def test_log_pipeline_cases(oracle):
    """The self-contained cases of the same fixture: goal LP and QP re-run from the stored rows / boxes land on the logged states
    exactly as the replay did, and the stored multipliers say which constraint families the optimum leans on."""
    g = H.load_golden("kat_log_pipeline")
    p = g["params"]
    cls = H.oracle_class(oracle, p, use_sfc=True)
    n_lsc = n_sfc = n_goal = 0
    assert 20 <= len(g["cases"]) <= 64
    for c in g["cases"]:
        L, box, mk = H.pipeline_case_arrays(oracle, p, c)
        st, goal, t = oracle.goal_opt(cls, c["goal_before_lp"], c["next_waypoint"], lsc=L, sfc_last=box[p["M"] - 1])
        assert st == 0 and abs(t - c["goal_lp_t"]) <= 1e-9 and np.abs(np.float32(goal) - np.array(c["goal"])).max() <= 1e-7
        R = oracle.solve(cls, mk(c["goal"]), L, box)
        assert R["status"] == 0 and abs(R["obj"] - c["oracle_obj"]) <= 1e-9 * max(1.0, abs(c["oracle_obj"]))
        assert H.logged_state_units(oracle, cls, c, R["x"]) <= c["match_units_of_6th_digit"] + 2 <= 402
        sz = oracle.count(cls, mk(c["goal"]), L)
        lam = R["lam"]
        if sz.n_lsc:
            assert abs(lam[sz.n_sfc:sz.n_sfc + sz.n_lsc].max() - c["max_lsc_multiplier"]) <= 1e-5 * max(1.0, c["max_lsc_multiplier"])
        n_lsc += c["max_lsc_multiplier"] > 1e-3
        n_sfc += c["max_sfc_multiplier"] > 1e-3
        n_goal += c["goal_lp_t"] > 1e-6
    assert n_lsc >= 10 and n_sfc >= 10 and n_goal >= 4
    # the LSC rows matter: without them the strongest case misses the log by far more than the replay's acceptance bound
    c = max(g["cases"], key=lambda c: c["max_lsc_multiplier"])
    L, box, mk = H.pipeline_case_arrays(oracle, p, c)
    ag0 = mk(c["goal"])
    ag0["n_obs"] = 0
    R0 = oracle.solve(cls, ag0, None, box)
    assert R0["status"] != 0 or H.logged_state_units(oracle, cls, c, R0["x"]) >= 10 * 400

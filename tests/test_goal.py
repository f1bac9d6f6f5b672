"""Goal LP (reference src/goal_optimizer.cpp): oracle restatement against an independent LP
solver's solutions and the reference log's known answer; the HIP closed form against the oracle."""
import numpy as np
import pytest

from tests import helpers as H


def _case_inputs(O, c):
    cls = O.make_class(M=c["M"], dim=c["dim"], use_sfc=bool(c["use_sfc"]))
    n_obs = len(c["lsc_d"])
    lsc = np.zeros((n_obs, c["M"], 6), O.LSC_DTYPE)
    if n_obs:
        lsc["p"][:, c["M"] - 1, 5] = np.array(c["lsc_p"])
        lsc["nrm"][:, c["M"] - 1, 5] = np.array(c["lsc_nrm"])
        lsc["d"][:, c["M"] - 1, 5] = np.array(c["lsc_d"])
    box = np.zeros(1, O.BOX_DTYPE)
    box["bmin"], box["bmax"] = c["box_min"], c["box_max"]
    return cls, lsc, box


def test_goal_lp_oracle_against_highs_golden(oracle):
    g = H.load_golden("goal_lp")
    n_feas = 0
    for c in g["cases"]:
        cls, lsc, box = _case_inputs(oracle, c)
        a, cc = oracle.goal_rows(cls, c["goal"], c["next_waypoint"], lsc if len(lsc) else None, box[0] if c["use_sfc"] else None)
        assert np.allclose(a, c["rows_a"], rtol=0, atol=1e-15) and np.allclose(cc, c["rows_c"], rtol=0, atol=1e-15)
        st, goal, t = oracle.goal_opt(cls, c["goal"], c["next_waypoint"], lsc if len(lsc) else None, box[0] if c["use_sfc"] else None)
        assert (st == 0) == (c["status"] == 0), (st, c["status"])
        if st == 0:
            n_feas += 1
            assert abs(t - c["t"]) <= 1e-9  # HiGHS run with 1e-10 feasibility tolerances
            gw = np.array(c["goal"]) - np.array(c["next_waypoint"])
            assert np.abs(goal - (gw * c["t"] + np.array(c["next_waypoint"]))).max() <= 1e-8
    assert n_feas >= 100


def test_goal_lp_reference_log_known_answer(oracle):
    """forest10_10, agent 1: waypoint x = 2.5, current goal x = 3.0, the SFC's -x face at 2.55 ->
    GoalOptimizer returns x = 2.55, which is what makes the logged trajectory reproduce (tests/golden/kat_log.json)."""
    cls = oracle.make_class(M=10, dim=2, use_sfc=True)
    box = np.zeros(1, oracle.BOX_DTYPE)
    box["bmin"], box["bmax"] = [2.55, -10, -10], [10, 10, 10]
    st, goal, t = oracle.goal_opt(cls, [3.0, 2.5, 1.0], [2.5, 2.5, 1.0], None, box[0])
    assert st == 0 and abs(goal[0] - 2.55) <= 1e-12 and abs(t - 0.1) <= 1e-12
    kat = H.load_golden("kat_log")
    case = [c for c in kat["cases"] if c["sfc"] is not None][0]
    assert abs(case["goal"][0] - 2.55) <= 1e-6  # the value the QP fixture was generated with


def test_goal_equal_to_waypoint_returns_waypoint(oracle):
    cls = oracle.make_class(M=5, dim=3, use_sfc=False)
    st, goal, t = oracle.goal_opt(cls, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0 + 5e-6])
    assert st == 0 and np.array_equal(goal, [1.0, 2.0, 3.0 + 5e-6])


@pytest.mark.gpu
def test_gpu_goal_lp_matches_oracle_on_golden(api, oracle):
    import torch

    assert torch.cuda.is_available()
    g = H.load_golden("goal_lp")
    by_class = {}
    for c in g["cases"]:
        by_class.setdefault((c["M"], c["dim"], c["use_sfc"]), []).append(c)
    checked = 0
    for (M, dim, use_sfc), cases in by_class.items():
        sol = api.Solver(api.make_desc(M=M, dim=dim, use_sfc=bool(use_sfc)))
        n = len(cases)
        hdr = np.zeros(n, api.HEADER_DTYPE)
        rows, off = [], [0]
        sfc = np.zeros((n, M), api.BOX_DTYPE)
        want = []
        for q, c in enumerate(cases):
            cls, lsc, box = _case_inputs(oracle, c)
            hdr["goal"][q], hdr["next_waypoint"][q] = c["goal"], c["next_waypoint"]
            hdr["n_obs"][q] = len(lsc)
            if len(lsc):
                rows.append(api.pack_rows(lsc).reshape(-1))
            off.append(off[-1] + len(lsc) * M * 6)
            sfc["bmin"][q], sfc["bmax"][q] = box["bmin"][0], box["bmax"][0]
            want.append(oracle.goal_opt(cls, c["goal"], c["next_waypoint"], lsc if len(lsc) else None, box[0] if use_sfc else None))
        rows = np.concatenate(rows) if rows else np.zeros(1, api.ROW_DTYPE)
        out, status = sol.optimize_goal_host(hdr, rows, np.array(off, dtype=np.uint64), sfc if use_sfc else None)
        for q, (st, goal, t) in enumerate(want):
            assert (status[q] == 0) == (st == 0), (q, status[q], st)
            if st == 0:
                # closed form on packed rows (b = d + n.p) vs candidate enumeration on the reference's records: fp64 noise
                assert np.abs(out["goal"][q] - goal).max() <= 1e-9, (q, out["goal"][q], goal)
            else:
                assert np.array_equal(out["goal"][q], hdr["goal"][q])  # untouched
            checked += 1
    assert checked == len(g["cases"])

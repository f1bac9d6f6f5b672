"""Structural known answers for the dim = 3 model: every vector the reference holds for this path is 2-D (its logs are dim 2, M 10), so
the 3-D assembler was pinned only transitively.  Here the reference-logged replans are EMBEDDED in three dimensions: the two logged
axes are mapped onto two of (x, y, z) by a permutation, the third axis gets a problem whose solution is "stay" (same start, goal and
waypoint, no velocity, normals without a component on it, wide bounds).  A dim-3 model that treats any axis differently from the
reference's per-axis loops (src/traj_optimizer.cpp:238-511 loop `k < dim` everywhere) -- wrong offset_dim stride, a z row that is
not assembled like x and y, LSC normals' third component mishandled -- cannot reproduce the logged motion in the permuted axes.
Cases: the first replan with an active corridor face (kat_log), the replan that only active LSC rows explain (kat_log_active), and a
sample of the 337 later replans (kat_log_replay).  CPU oracle here; the HIP kernel (M = 10 in 3-D: nz = 84, the two-/four-wavefront
nested-dissection instances) in the -m gpu test."""
import itertools

import numpy as np
import pytest

from tests import helpers as H

PERMS = [(0, 2, 1), (2, 0, 1), (1, 2, 0), (2, 1, 0)]  # old axis k of (x, y, third) -> new axis perm[k]; every one moves a logged axis into z
THIRD = 0.6


def _embed(v3, perm, third=None):
    out = np.zeros(3)
    for k in range(3):
        out[perm[k]] = v3[k] if (k < 2 or third is None) else third
    return out


def _cases(oracle):
    """(name, params, p0, v0, a0, goal, waypoint, LSC[n,M,6] or None, sfc dict or None, logged states [(dt, p, v, a)])"""
    out = []
    g = H.load_golden("kat_log")
    for c in g["cases"]:
        log = [(st["t"], st["p"], st["v"], st["a"]) for st in g["agents"][c["agent"]]["states"][1:]]
        out.append(("first_replan_agent%d" % c["agent"], g["params"], c["p0"], [0, 0, 0], [0, 0, 0], c["goal"], c["next_waypoint"], None, c["sfc"], log))
    g = H.load_golden("kat_log_active")
    c = g["cases"][0]
    L = np.zeros((len(c["neighbours"]), g["params"]["M"], 6), oracle.LSC_DTYPE)
    L["p"], L["nrm"], L["d"] = c["lsc_p"], c["lsc_nrm"], c["lsc_d"]
    out.append(("active_lsc_row", g["params"], c["p0"], c["v0"], c["a0"], c["goal"], c["next_waypoint"], L, None,
                [(st["t"] - c["t"], st["p"], st["v"], st["a"]) for st in c["states"]]))
    g = H.load_golden("kat_log_replay")
    for c in g["cases"][::40]:
        out.append(("replay_a%d_r%d" % (c["agent"], c["replan"]), g["params"], c["p0"], c["v0"], c["a0"], c["goal"], c["next_waypoint"], None, None,
                    [(st["t"] - c["t"], st["p"], st["v"], st["a"]) for st in c["states"]]))
    return out


def _embedded(oracle, case, perm):
    name, p, p0, v0, a0, goal, wp, L, sfc, log = case
    M = p["M"]
    wmin, wmax = _embed(p["world_min"], perm), _embed(p["world_max"], perm)
    cls = oracle.make_class(M=M, dim=3, dt=p["dt"], w_c=p["w_c"], w_t=p["w_t"], comm_range=p["comm_range"], planner_lsc=True, use_sfc=sfc is not None,
                            world_min=wmin, world_max=wmax)
    e = lambda v, third=None: _embed(np.asarray(v, dtype=np.float64), perm, third)  # noqa: E731
    ag = oracle.make_agent(p0=e(p0, THIRD), v0=e(v0, 0.0), a0=e(a0, 0.0), goal=e(goal, THIRD), next_waypoint=e(wp, THIRD), vmax=p["vmax"], amax=p["amax"],
                           radius=p["radius"], nominal_velocity=p["nominal_velocity"], n_obs=0 if L is None else len(L))
    L3 = None
    if L is not None:
        L3 = np.zeros(L.shape, oracle.LSC_DTYPE)
        for idx in np.ndindex(L.shape):
            nrm = np.array(L[idx]["nrm"])
            nrm[2] = 0.0  # (the 2-D model ignores the third component, src/traj_optimizer.cpp:419-421 `if (dim == 3)`)
            L3[idx]["p"], L3[idx]["nrm"], L3[idx]["d"] = e(L[idx]["p"], THIRD), _embed(nrm, perm), L[idx]["d"]
    box = None
    if sfc is not None:
        box = np.zeros(M, oracle.BOX_DTYPE)
        box["bmin"], box["bmax"] = e(sfc["bmin"]), e(sfc["bmax"])
    return cls, ag, L3, box


def _check_logged_motion(oracle, cls, x, log, perm, third_tol=1e-9):
    for dt_, lp, lv, la in log:
        pos, vel, acc = oracle.state_at(cls, x, dt_)
        for k in range(2):
            assert abs(pos[perm[k]] - lp[k]) <= 2e-5 and abs(vel[perm[k]] - lv[k]) <= max(2e-5, 1e-4 * abs(lv[k])) and abs(acc[perm[k]] - la[k]) <= max(3e-4, 3e-4 * abs(la[k])), (dt_, k)
        assert abs(pos[perm[2]] - THIRD) <= third_tol and abs(vel[perm[2]]) <= third_tol * 10 and abs(acc[perm[2]]) <= third_tol * 100


def test_reference_log_embedded_in_three_dimensions_oracle(oracle):
    cases = _cases(oracle)
    assert len(cases) >= 10
    for case, perm in itertools.product(cases, PERMS):
        name, p = case[0], case[1]
        cls3, ag3, L3, box3 = _embedded(oracle, case, perm)
        R3 = oracle.solve(cls3, ag3, L3, box3)
        assert R3["status"] == 0, (name, perm)
        _check_logged_motion(oracle, cls3, R3["x"], case[9], perm)
        # ... and it IS the dim-2 optimum, axis by axis, objective included
        cls2 = H.oracle_class(oracle, p, use_sfc=case[8] is not None)
        ag2 = oracle.make_agent(p0=case[2], v0=case[3], a0=case[4], goal=case[5], next_waypoint=case[6], vmax=p["vmax"], amax=p["amax"], radius=p["radius"],
                                nominal_velocity=p["nominal_velocity"], n_obs=0 if case[7] is None else len(case[7]))
        box2 = None
        if case[8] is not None:
            box2 = np.zeros(p["M"], oracle.BOX_DTYPE)
            box2["bmin"], box2["bmax"] = case[8]["bmin"], case[8]["bmax"]
        R2 = oracle.solve(cls2, ag2, case[7], box2)
        P = 6 * p["M"]
        x2, x3 = R2["x"].reshape(2, P), R3["x"].reshape(3, P)
        for k in range(2):
            assert np.abs(x3[perm[k]] - x2[k]).max() <= 1e-8, (name, perm, k)
        assert np.abs(x3[perm[2]] - THIRD).max() <= 1e-9
        # the third axis adds only its constant terminal term w_t * ts * third^2 ... no: (x - goal)^2 = 0 there; the objectives agree
        assert abs(R3["obj"] - R2["obj"]) <= 1e-9 * max(1.0, abs(R2["obj"])), (name, perm, R3["obj"], R2["obj"])


@pytest.mark.gpu
def test_reference_log_embedded_in_three_dimensions_gpu(api, oracle, torch_cuda, monkeypatch):
    """The same embedded replans through the C ABI: the M = 10, dim 3 class (nz = 84) on every compiled wavefront count and on the
    run-time-shaped kernel."""
    cases = _cases(oracle)
    for knob in ({}, {"LSCQP_WAVES": "2"}, {"LSCQP_WAVES": "4"}, {"LSCQP_FORCE_GENERIC": "1"}):
        for k_ in ("LSCQP_WAVES", "LSCQP_FORCE_GENERIC"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in knob.items():
            monkeypatch.setenv(k_, v_)
        for perm in PERMS[:2]:
            for use_sfc in (False, True):
                sel = [c for c in cases if (c[8] is not None) == use_sfc]
                if not sel:
                    continue
                emb = [_embedded(oracle, c, perm) for c in sel]
                cls3 = emb[0][0]
                p = sel[0][1]
                sol = api.Solver(api.make_desc(M=p["M"], dim=3, dt=p["dt"], w_c=p["w_c"], w_t=p["w_t"], comm_range=p["comm_range"], use_sfc=use_sfc,
                                               world_min=_embed(p["world_min"], perm), world_max=_embed(p["world_max"], perm)))
                hdr, rows, off, sfc = H.abi_batch(api, oracle, cls3, [e[1] for e in emb], [e[2] for e in emb], [e[3] for e in emb] if use_sfc else None, p["M"])
                G = sol.solve_host(hdr, rows, off, sfc)
                assert (G["status"] == 0).all(), (knob, perm, G["status"])
                for q, c in enumerate(sel):
                    _check_logged_motion(oracle, emb[q][0], G["x"][q], c[9], perm, third_tol=1e-8)

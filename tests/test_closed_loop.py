"""All rows of the path chained in closed loop on the device, in the reference's default configuration (generateCLSC +
constructSFCFromConvexHull + goal LP + trajectory QP + isSolValid / doStep + safety metrics), on the reference's forest10
world with its 10 agents (tools/closed_loop.py; the waypoints come from a host-side stand-in for the out-of-scope grid
planner, without MAPF conflict resolution)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["compiled", "runtime_shaped"])
def test_forest10_closed_loop_is_safe_and_feasible(kernel, monkeypatch):
    """kernel = runtime_shaped: the same mission with every QP on csrc/lscqp_generic.hip (LSCQP_FORCE_GENERIC=1) -- 600 QPs of a closed
    loop whose rows, corridors and goals the device produced itself."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    if kernel == "runtime_shaped":
        monkeypatch.setenv("LSCQP_FORCE_GENERIC", "1")
    else:
        monkeypatch.delenv("LSCQP_FORCE_GENERIC", raising=False)
    log = closed_loop.run(os.path.join(ROOT, "tests", "golden", "forest10_world.json"), steps=60)
    # every QP of 60 replans x 10 agents solves and passes isSolValid: the generated constraints are mutually consistent
    assert log["qp_failed"] == 0 and log["invalid"] == 0 and log["sfc_kept"] == 0, log
    # the LSC guarantee: agents never come closer than the sum of their radii (reference summary: safety_ratio_agent >= 1)
    # (up to the float32 truncation of the control points, which the reference applies as well: 5e-7 m on a 5 m coordinate
    # is 3e-6 of the 0.3 m the ratio is measured in)
    assert log["min_safety_ratio"] >= 1.0 - 5e-6, log
    assert log["max_vel_excess"] <= 1e-5 and log["max_acc_excess"] <= 1e-5, log
    assert log["mean_progress_m"] > 1.5, log  # and they do fly towards their goals
    assert log["max_iters"] <= 30, log


@pytest.mark.gpu
def test_closed_loop_replan_matches_the_oracle(oracle):
    """One replan from the middle of the run -- CLSC rows, corridors and goal all produced on the device -- re-solved by the
    CPU oracle from the very same buffers: the parity bar of the QP (objective 1e-8, x 1e-6 m) holds on the reference's
    default configuration too (M = 10, 2-D, 9 neighbours, agents interacting)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    log = closed_loop.run(os.path.join(ROOT, "tests", "golden", "forest10_world.json"), steps=26, keep_step=25)
    K = log["kept"]
    hdr, rows, sfc = K["hdr"], K["rows"], K["sfc"]
    N, M, n_obs = len(hdr), sfc.shape[1], K["n_obs"]
    cls = oracle.make_class(M=M, dim=2, use_sfc=True, world_min=K["world_min"], world_max=K["world_max"])
    R = rows.reshape(N, n_obs, M, 6)
    active = 0
    for q in range(N):
        ag = oracle.make_agent(p0=hdr["p0"][q], v0=hdr["v0"][q], a0=hdr["a0"][q], goal=hdr["goal"][q], next_waypoint=hdr["next_waypoint"][q],
                               vmax=hdr["vmax"][q], amax=hdr["amax"][q], radius=hdr["radius"][q],
                               nominal_velocity=hdr["nominal_velocity"][q], n_obs=n_obs)
        lsc = np.zeros((n_obs, M, 6), oracle.LSC_DTYPE)  # packed rows n.c >= b  ==  LSC with p_obs = 0, d = b
        lsc["nrm"][..., 0], lsc["nrm"][..., 1], lsc["nrm"][..., 2], lsc["d"] = R["nx"][q], R["ny"][q], R["nz"][q], R["b"][q]
        box = np.zeros(M, oracle.BOX_DTYPE)
        box["bmin"], box["bmax"] = sfc["bmin"][q], sfc["bmax"][q]
        o = oracle.solve(cls, ag, lsc, box)
        assert o["status"] == 0 and K["status"][q] == 0
        assert abs(o["obj"] - K["obj"][q]) <= 1e-8 * max(1.0, abs(o["obj"])), (q, o["obj"], K["obj"][q])
        assert np.abs(o["x"] - K["x"][q]).max() <= 1e-6, q
        active += int((np.linalg.norm(lsc["nrm"], axis=-1) > 1e-5).any())
    assert active >= N // 2  # the agents do see each other at that point of the run


@pytest.mark.gpu
def test_synthetic_forest_closed_loop_with_64_agents():
    """BASELINE configs[1]'s agent count in closed loop: 64 agents swapping sides through a synthetic forest (the reference's
    world format), M = 10 segments, up to ~40 agents within the 3 m communication range.  The row slots start at 20 and follow
    the in-range counts (the reference hands every in-range agent to the planner): NO agent's neighbour list is cut in any
    replan.  Every QP of 80 replans solves (a jammed warm start is re-launched cold), nothing collides."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    log = closed_loop.run(closed_loop.random_forest_world(64), steps=80, n_obs=20)
    assert log["qp_failed"] == 0 and log["invalid"] == 0, log
    assert log["min_safety_ratio"] >= 1.0 - 5e-6 and log["max_vel_excess"] <= 1e-5 and log["max_acc_excess"] <= 1e-5, log
    assert log["max_in_range"] > 20 and log["mean_progress_m"] > 3.0, log
    assert log["truncated_agent_steps"] == 0 and log["row_slots"] >= log["max_in_range"], log


@pytest.mark.gpu
def test_device_pipeline_replays_the_reference_log(oracle):
    """The reference's own logged mission (forest10_10, 10 agents x 79 replans; tests/golden/sim_log_states.json) flown again with EVERY
    row of the path on the device: each replan takes the agents' logged states and the waypoints the replay fixture inferred
    (tests/golden/kat_log_pipeline.json `replay`; the grid planner / MAPF layer that produced them is out of scope), everything
    else -- shifted previous plans, agents in range, CLSC rows, corridors over the voxel map, goal LP, trajectory QP -- is the
    device chain's own state from replan to replan.  Every replan must land on the next two logged lines (positions, velocities,
    accelerations) to the precision of the log's six digits, and on the goal point of the CPU replay."""
    import json

    import numpy as np

    from tests import helpers as H

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    g = H.load_golden("kat_log_pipeline")
    S = H.load_golden("sim_log_states")
    W = json.load(open(os.path.join(ROOT, "tests", "golden", "forest10_world.json")))
    p = g["params"]
    K, N = 79, 10
    pos, vel, acc, t = np.array(S["pos"]), np.array(S["vel"]), np.array(S["acc"]), np.array(S["t"])
    way, goal, match, n_nbr = np.zeros((K, N, 3)), np.zeros((K, N, 3)), np.zeros((K, N)), np.zeros((K, N), int)
    way[..., 2] = goal[..., 2] = W["z_2d"]
    for r in g["replay"]:
        way[r["replan"], r["agent"], :2], goal[r["replan"], r["agent"], :2], match[r["replan"], r["agent"]] = r["waypoint"], r["goal"], r["match"]
        n_nbr[r["replan"], r["agent"]] = len(r["neighbours"])
    state = np.concatenate([pos[0:2 * K:2], vel[0:2 * K:2], acc[0:2 * K:2]], axis=2)
    state[..., 2] = W["z_2d"]
    log = closed_loop.run(W, steps=K, script=dict(waypoint=way, state=state))
    assert log["qp_failed"] == 0 and log["invalid"] == 0 and log["goal_infeasible"] == 0 and log.get("truncated_agent_steps", 0) == 0, {
        k: v for k, v in log.items() if k not in ("x", "goal", "n_in_range")}
    cls = H.oracle_class(oracle, p, use_sfc=True)
    worst, n_rep = 0.0, 0
    for k in range(K):
        assert np.array_equal(log["n_in_range"][k], n_nbr[k])  # broadcastMsgs' range filter saw the same agents
        assert np.abs(np.float32(log["goal"][k][:, :2]) - goal[k, :, :2]).max() <= 2e-6, (k, log["goal"][k], goal[k])
        for a in range(N):
            err = 0.0
            for j in (2 * k + 1, 2 * k + 2):
                got = oracle.state_at(cls, log["x"][k][a], t[j] - t[2 * k])
                for gv, lv in zip(got, (pos[j, a], vel[j, a], acc[j, a])):
                    err = max(err, max(H.log_units(gv[i], lv[i]) for i in range(2)))
            assert err <= match[k, a] + 40, (k, a, err, match[k, a])
            worst, n_rep = max(worst, err), n_rep + 1
    assert n_rep == 790 and worst <= 440

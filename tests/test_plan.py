"""lscqp_plan (include/lscqp.h, "the caller of the path"): the whole replan of a batch of agents as one chain of device work --
the device analogue of TrajPlanner::planImpl (reference src/traj_planner.cpp:117-139) -- eager and through a captured hipGraph,
checked against the reference's own logged mission."""
import json
import os
import time

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mission():
    g = H.load_golden("kat_log_pipeline")
    S = H.load_golden("sim_log_states")
    W = json.load(open(os.path.join(ROOT, "tests", "golden", "forest10_world.json")))
    K, N = 79, 10
    pos, vel, acc, t = np.array(S["pos"]), np.array(S["vel"]), np.array(S["acc"]), np.array(S["t"])
    way, goal, match, n_nbr = np.zeros((K, N, 3)), np.zeros((K, N, 3)), np.zeros((K, N)), np.zeros((K, N), int)
    way[..., 2] = goal[..., 2] = W["z_2d"]
    for r in g["replay"]:
        k, a = r["replan"], r["agent"]
        way[k, a, :2], goal[k, a, :2], match[k, a], n_nbr[k, a] = r["waypoint"], r["goal"], r["match"], len(r["neighbours"])
    state = np.concatenate([pos[0:2 * K:2], vel[0:2 * K:2], acc[0:2 * K:2]], axis=2)
    state[..., 2] = W["z_2d"]
    return g, W, dict(K=K, N=N, pos=pos, vel=vel, acc=acc, t=t, way=way, goal=goal, match=match, n_nbr=n_nbr, state=state)


def _make_plan(api, W, N, n_obs=9, closed_loop=False):
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, n_obs, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True,
                    closed_loop=closed_loop, z_2d=W["z_2d"])
    return sol, wmap, plan


def _fly(api, torch, plan, W, m, graph):
    plan.reset(np.array(W["starts"], dtype=np.float64))
    xs, goals, counts = [], [], []
    for k in range(m["K"]):
        plan.put(api.PLAN_STATE, m["state"][k])
        plan.put(api.PLAN_WAYPOINT, m["way"][k])
        plan.step(graph=graph)
        torch.cuda.synchronize()
        st = plan.get(api.PLAN_STATUS)
        assert (st == 0).all() and (plan.get(api.PLAN_GOAL_STATUS) == 0).all(), (k, st)
        xs.append(plan.get(api.PLAN_PLAN).reshape(m["N"], -1))
        goals.append(plan.get(api.PLAN_GOAL).reshape(m["N"], 3))
        counts.append(plan.get(api.PLAN_IN_RANGE))
    return np.array(xs), np.array(goals), np.array(counts)


@pytest.mark.gpu
def test_plan_chain_replays_the_reference_log_eagerly_and_as_a_graph(api, oracle, torch_cuda):
    """The reference's logged mission (forest10_10: 10 agents x 79 replans) flown again by lscqp_plan_step: per replan the host only
    writes the logged states and the waypoints the replay fixture inferred; shifted plans, range filter, CLSC rows, corridors, goal LP
    (held as float32), terminal segments, QP, failsafe and plan update all run in the plan's chain.  Every replan lands on the next two
    logged lines to the log's precision and on the CPU replay's goal point (float32, at most one ulp apart).  The captured hipGraph gives the same
    bits as the eager chain."""
    import torch

    g, W, m = _mission()
    sol, wmap, plan = _make_plan(api, W, m["N"])
    x_e, goal_e, cnt_e = _fly(api, torch, plan, W, m, graph=False)
    assert plan.graph_nodes() == 0
    x_g, goal_g, cnt_g = _fly(api, torch, plan, W, m, graph=True)
    assert plan.graph_nodes() >= 8
    assert np.array_equal(x_e, x_g) and np.array_equal(goal_e, goal_g) and np.array_equal(cnt_e, cnt_g)
    assert np.array_equal(cnt_e, m["n_nbr"])  # broadcastMsgs' range filter saw the same agents as the replay
    # the goal points of the CPU replay, to the last float32 bit but for a handful that sit one ulp away (the LP's step t differs in
    # the last bits of its fp64 value between the two implementations, and the goal is rounded to float32 afterwards)
    ga, gb = np.float32(goal_e[..., :2]), np.float32(m["goal"][..., :2])
    assert (np.abs(ga - gb) <= np.spacing(np.maximum(np.abs(gb), np.float32(1.0)))).all() and (ga != gb).sum() <= 8
    cls = H.oracle_class(oracle, g["params"], use_sfc=True)
    worst = 0.0
    for k in range(m["K"]):
        for a in range(m["N"]):
            err = 0.0
            for j in (2 * k + 1, 2 * k + 2):
                got = oracle.state_at(cls, x_e[k, a], m["t"][j] - m["t"][2 * k])
                for gv, lv in zip(got, (m["pos"][j, a], m["vel"][j, a], m["acc"][j, a])):
                    err = max(err, max(H.log_units(gv[i], lv[i]) for i in range(2)))
            assert err <= m["match"][k, a] + 40, (k, a, err, m["match"][k, a])
            worst = max(worst, err)
    assert worst <= 440
    plan.close()


@pytest.mark.gpu
def test_plan_closed_loop_graph_equals_the_eager_chain_and_keeps_the_mission_safe(api, torch_cuda):
    """Closed loop on the device (the plan steps its own agents: doStep's state becomes the next replan's state), waypoints fixed at
    the mission goals' first grid step: 60 replans eager, 60 through the graph -- same bits, no failed QP.  The two
    durations are written to gpurun_out/plan_chain_timing.json (recorded, never asserted: this suite carries correctness)."""
    import torch

    g, W, m = _mission()
    sol, wmap, plan = _make_plan(api, W, m["N"], closed_loop=True)
    starts = np.array(W["starts"], dtype=np.float64)
    out = {}
    for mode in ("eager", "graph"):
        plan.reset(starts)
        plan.put(api.PLAN_WAYPOINT, m["way"][0])
        plan.step(graph=False)  # first replan (initializeSFC): eager in both runs
        plan.put(api.PLAN_WAYPOINT, m["way"][5])  # half a metre on, then held: the agents fly there and hover
        plan.step(graph=(mode == "graph"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            plan.step(graph=(mode == "graph"))
        torch.cuda.synchronize()
        out[mode] = (time.perf_counter() - t0) / 60, plan.get(api.PLAN_PLAN).copy(), plan.get(api.PLAN_STATE).copy()
        assert (plan.get(api.PLAN_STATUS) == 0).all()
    assert np.array_equal(out["eager"][1], out["graph"][1]) and np.array_equal(out["eager"][2], out["graph"][2])
    print("replan chain, 10 agents x M10: eager %.1f us, graph %.1f us per replan" % (out["eager"][0] * 1e6, out["graph"][0] * 1e6))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(eager_us=out["eager"][0] * 1e6, graph_us=out["graph"][0] * 1e6, nodes=plan.graph_nodes()),
              open(os.path.join(ROOT, "gpurun_out", "plan_chain_timing.json"), "w"))
    plan.close()


@pytest.mark.gpu
def test_plan_sharded_over_two_owners_equals_the_single_plan(api, torch_cuda):
    """Section 8e layout of the chain: two plans own agents 0-4 and 5-9 of the same mission (as two ranks would); after every replan the
    owners' slices of the plan and goal buffers are exchanged (what lscqp_allgather does between GPUs; here through the host) -- the first
    20 replans of the logged mission give bit for bit the plans of the single 10-agent plan."""
    import torch

    g, W, m = _mission()
    N, K, h = m["N"], 20, 5
    sol, wmap, whole = _make_plan(api, W, N)
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    parts = [api.Plan(sol, wmap, h, 9, ag, n_total=N, first_agent=f, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"])
             for f in (0, h)]
    starts = np.array(W["starts"], dtype=np.float64)
    for p in [whole] + parts:
        p.reset(starts)
    for k in range(K):
        whole.put(api.PLAN_STATE, m["state"][k])
        whole.put(api.PLAN_WAYPOINT, m["way"][k])
        whole.step()
        for r, p in enumerate(parts):
            p.put(api.PLAN_STATE, m["state"][k])
            p.put(api.PLAN_WAYPOINT, m["way"][k][r * h:(r + 1) * h])
            p.step()
        torch.cuda.synchronize()
        x = [p.get(api.PLAN_PLAN).reshape(N, -1) for p in parts]
        gl = [p.get(api.PLAN_GOAL).reshape(N, 3) for p in parts]
        x_all, g_all = np.concatenate([x[0][:h], x[1][h:]]), np.concatenate([gl[0][:h], gl[1][h:]])
        for p in parts:  # the all-gather: every owner receives the other's slice
            p.put(api.PLAN_PLAN, x_all)
            p.put(api.PLAN_GOAL, g_all)
        assert np.array_equal(x_all, whole.get(api.PLAN_PLAN).reshape(N, -1)), k
        assert np.array_equal(g_all, whole.get(api.PLAN_GOAL).reshape(N, 3)), k
        assert np.array_equal(np.concatenate([p.get(api.PLAN_IN_RANGE) for p in parts]), whole.get(api.PLAN_IN_RANGE))
    for p in [whole] + parts:
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_plan_group_step_over_the_communicator_equals_the_plain_step(api, torch_cuda, graph):
    """lscqp_plan_group_step (section 8e for the chain): the plans of a mission, one per device of a communicator, stepped on the
    communicator's streams and followed by the in-place RCCL exchange of the owners' plan / state / goal slices.  The GPU box has one
    device, so the group is one plan owning every agent: the exchange runs through RCCL (a one-rank ncclAllGather in place) and must
    leave the closed loop bit for bit where plain lscqp_plan_step calls take it.  A group that does not cover the mission, or whose
    plans are not one per device, is refused."""
    g, W, m = _mission()
    N, K = m["N"], 15
    sol, wmap, ref = _make_plan(api, W, N, closed_loop=True)
    sol2, wmap2, grp = _make_plan(api, W, N, closed_loop=True)
    comm = api.Comm(1)
    assert comm.size == 1
    starts = np.array(W["starts"], dtype=np.float64)
    ref.reset(starts)
    grp.reset(starts)
    for k in range(K):
        ref.put(api.PLAN_WAYPOINT, m["way"][k])
        grp.put(api.PLAN_WAYPOINT, m["way"][k])
        ref.step(graph=graph)
        comm.plan_group_step([grp], graph=graph)
        torch_cuda.cuda.synchronize()
        comm.synchronize()
        for which in (api.PLAN_PLAN, api.PLAN_STATE, api.PLAN_GOAL, api.PLAN_STATUS):
            assert np.array_equal(ref.get(which), grp.get(which)), (k, which)
    assert (grp.get(api.PLAN_STATUS) == 0).all()
    if graph:
        assert grp.graph_nodes() == ref.graph_nodes() > 0
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    half = api.Plan(sol, wmap, 5, 9, ag, n_total=N, first_agent=0, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"])
    with pytest.raises(api.LscqpError, match="cover the mission"):
        comm.plan_group_step([half])
    late = api.Plan(sol, wmap, 5, 9, ag, n_total=N, first_agent=5, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"])
    with pytest.raises(api.LscqpError, match="consecutive blocks"):
        comm.plan_group_step([late])
    with pytest.raises(ValueError):
        comm.plan_group_step([half, late])
    for p in (ref, grp, half, late):
        p.close()
    comm.close()


@pytest.mark.gpu
def test_plan_tensor_views_carry_the_exchange_between_two_owners(api, torch_cuda):
    """Plan.tensor: zero-copy torch views of the plan's device buffers -- what sharding.exchange_plan_buffers hands to torch.distributed
    in a one-process-per-GPU job.  Two owners (agents 0-4 and 5-9) on one device, closed loop; after every replan each owner's block of
    the plan / state / goal views is copied into the other owner's views ON THE DEVICE (what the broadcasts do between ranks): 12
    replans equal the single 10-agent plan bit for bit."""
    torch = torch_cuda
    from lsc_dr_planner_amd import sharding

    g, W, m = _mission()
    N, K = m["N"], 12
    sol, wmap, whole = _make_plan(api, W, N, closed_loop=True)
    parts = []
    for r in range(2):
        lo, hi = sharding.shard_range(N, 2, r)
        parts.append(api.Plan(sol, wmap, hi - lo, 9, _agents(api, W, N), n_total=N, first_agent=lo, constraint_mode=api.GEN_CLSC,
                              sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=W["z_2d"]))
    starts = np.array(W["starts"], dtype=np.float64)
    for p in [whole] + parts:
        p.reset(starts)
    views = [[p.tensor(w) for w in (api.PLAN_PLAN, api.PLAN_STATE, api.PLAN_GOAL)] for p in parts]
    assert np.array_equal(views[0][0].cpu().numpy(), parts[0].get(api.PLAN_PLAN))
    for k in range(K):
        whole.put(api.PLAN_WAYPOINT, m["way"][k])
        whole.step()
        for r, p in enumerate(parts):
            lo, hi = sharding.shard_range(N, 2, r)
            p.put(api.PLAN_WAYPOINT, m["way"][k][lo:hi])
            p.step()
        torch.cuda.synchronize()
        for b in range(3):
            per = views[0][b].numel() // N
            for r in range(2):
                lo, hi = sharding.shard_range(N, 2, r)
                views[1 - r][b][lo * per:hi * per].copy_(views[r][b][lo * per:hi * per])
        torch.cuda.synchronize()
        for w in (api.PLAN_PLAN, api.PLAN_STATE, api.PLAN_GOAL):
            assert np.array_equal(parts[0].get(w), whole.get(w)) and np.array_equal(parts[1].get(w), whole.get(w)), (k, w)
    sharding.exchange_plan_buffers(views[0], N)  # (no process group: nothing to exchange, nothing touched)
    for p in [whole] + parts:
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_plan_of_a_single_agent_without_neighbour_slots(api, torch_cuda, graph):
    """The empty end of the chain: one agent, n_obs = 0 (no row buffer, no range filter hits, the constraint generation is skipped, the
    goal LP and the QP see corridor rows only).  Agent 0 of the forest10 mission alone, closed loop, its logged waypoints: every replan
    solves, the plan starts at the current state, stays inside its corridors, and the agent gets somewhere."""
    g, W, m = _mission()
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    plan = api.Plan(sol, wmap, 1, 0, _agents(api, W, 1), constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=W["z_2d"])
    start = np.array(W["starts"][0:1], dtype=np.float64)
    plan.reset(start)
    for k in range(40):
        plan.put(api.PLAN_WAYPOINT, m["way"][k][0:1])
        before = plan.get(api.PLAN_STATE).reshape(1, 9)
        plan.step(graph=graph)
        torch_cuda.cuda.synchronize()
        assert plan.get(api.PLAN_STATUS)[0] == 0 and plan.get(api.PLAN_GOAL_STATUS)[0] == 0, k
        assert plan.get(api.PLAN_VALID)[0] == 1 and plan.get(api.PLAN_IN_RANGE)[0] == 0
        x = plan.get(api.PLAN_PLAN).reshape(2, 10, 6)
        assert np.abs(x[:, 0, 0] - before[0, :2]).max() < 1e-6  # (float32 truncation of the plan)
        box = plan.get(api.PLAN_SFC)
        for mseg in range(10):
            assert (x[:, mseg, :] >= box["bmin"][mseg][:2, None] - 1e-6).all() and (x[:, mseg, :] <= box["bmax"][mseg][:2, None] + 1e-6).all()
    moved = np.linalg.norm(plan.get(api.PLAN_STATE).reshape(1, 9)[0, :2] - start[0, :2])
    assert moved > 2.0, moved
    plan.close()


@pytest.mark.gpu
def test_plan_chain_with_512_agents(api, torch_cuda):
    """The chain at the agent count of BASELINE configs[2]: 512 agents on a ring of 49 m radius (0.6 m apart: about ten of them within
    the 3 m communication range of each) around a synthetic forest of 600 pillars, M = 10, 2-D, 20 neighbour slots, safety figures on.
    Ten replans through the captured graph, every agent heading for the opposite side: every QP solves and is valid, no neighbour list
    is cut, nobody comes closer than the radii allow, the graph replays what the eager chain does (plans bit for bit)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    N = 512
    W = closed_loop.random_forest_world(N, side=110.0, n_boxes=600, seed=3)
    starts = np.array(W["starts"], dtype=np.float64)
    assert np.unique(starts, axis=0).shape[0] == N  # (the 0.5 m snap of the ring leaves them apart)
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    plans = [api.Plan(sol, wmap, N, 20, _agents(api, W, N), constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True,
                      z_2d=W["z_2d"], safety_samples=2, record_time_step=0.1) for _ in range(2)]
    goals = np.array(W["goals"], dtype=np.float64)
    way = starts + 0.5 * (goals - starts) / np.linalg.norm(goals - starts, axis=1, keepdims=True)  # half a metre inwards: what a router hands out
    way[:, 2] = W["z_2d"]
    for p in plans:
        p.reset(starts)
    worst = 1e9
    for k in range(10):
        for graph, p in zip((False, True), plans):
            p.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
            p.step(graph=graph)
        torch_cuda.cuda.synchronize()
        e, gph = plans
        assert np.array_equal(e.get(api.PLAN_PLAN), gph.get(api.PLAN_PLAN)), k
        st = gph.get(api.PLAN_STATUS)
        assert (st == 0).all(), (k, np.bincount(st))
        assert (gph.get(api.PLAN_VALID) == 1).all() and (gph.get(api.PLAN_GOAL_STATUS) == 0).all()
        cnt = gph.get(api.PLAN_IN_RANGE)
        assert cnt.max() <= 20 and cnt.min() >= 4, (cnt.min(), cnt.max())
        worst = min(worst, float(gph.get(api.PLAN_SAFETY)["safety_ratio"].min()))
        state = gph.get(api.PLAN_STATE).reshape(N, 9)
        step_in = np.abs(state[:, :2] - way[:, :2]).max(axis=1) < 0.3
        way[step_in, :2] += 0.5 * (goals[step_in, :2] - way[step_in, :2]) / np.linalg.norm(goals[step_in, :2] - way[step_in, :2], axis=1, keepdims=True)
    assert worst >= 1.0 - 5e-6, worst
    moved = np.linalg.norm(plans[1].get(api.PLAN_STATE).reshape(N, 9)[:, :2] - starts[:, :2], axis=1)
    assert moved.mean() > 0.5, moved.mean()
    assert plans[1].graph_nodes() >= 8
    for p in plans:
        p.close()


def _world_3d(n_agents=12):
    """An 8 x 8 x 4 m room with six boxes (two of them floating), agents on a tilted ring swapping sides: start / goal pairs antipodal."""
    ang = np.linspace(0, 2 * np.pi, n_agents, endpoint=False)
    starts = np.c_[3.0 * np.cos(ang), 3.0 * np.sin(ang), 2.0 + 0.8 * np.sin(2 * ang)]
    starts = np.round(starts * 4) / 4
    goals = np.c_[-starts[:, 0], -starts[:, 1], 4.0 - starts[:, 2]]
    boxes = [[0.0, 0.0, 1.0, 0.6, 0.6, 2.0], [1.5, -1.0, 2.75, 0.5, 0.5, 1.5], [-1.5, 1.25, 0.75, 0.5, 0.5, 1.5], [0.0, 2.0, 2.0, 0.8, 0.4, 0.6],
             [0.5, -2.25, 2.5, 0.4, 0.8, 0.5], [-2.0, -0.5, 2.0, 0.4, 0.4, 4.0]]
    return dict(boxes=boxes, world_min=[-4.0, -4.0, 0.0], world_max=[4.0, 4.0, 4.0], resolution=0.1, max_dist=1.0, radius=0.15,
                starts=starts, goals=goals)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["LSC", "BVC", "LSC-nomap"])
def test_plan_chain_in_three_dimensions_with_static_goals(api, oracle, torch_cuda, mode):
    """The chain at BASELINE configs[1]'s class -- M = 5, three dimensions, downwash 2 -- in the modes the forest10 replay does not
    touch: generateLSC (mode/planner dlsc) resp. generateBVC with prediction and initial trajectory from the current position
    (mode/planner bvc), corridors from constructSFCFromPoint, static goal points (no goal LP), every agent every other's neighbour.
    Twelve agents swap sides of a room with six boxes, 45 replans in closed loop through the captured graph: every QP solves and is
    valid, nobody comes closer than the downwash ellipsoids allow (safety ratio >= 1 up to the float32 truncation of the plans) and
    no limit is exceeded, the agents make way; and three replans are re-solved by the CPU oracle from the chain's own header, row and
    corridor buffers at the QP's parity bar."""
    W = _world_3d()
    N, M = len(W["starts"]), 5
    corridors = mode != "LSC-nomap"  # (world/use_octomap = false: no map, no corridor rows -- the class is built with use_sfc = 0)
    sol = api.Solver(api.make_desc(M=M, dim=3, dt=0.2, comm_range=0.0, use_sfc=corridors, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"]) if corridors else None
    kw = dict(constraint_mode=api.GEN_LSC) if mode != "BVC" else dict(constraint_mode=api.GEN_BVC, prediction_mode=api.TRAJ_FROM_POSITION,
                                                                       initial_traj_mode=api.TRAJ_FROM_POSITION)
    plan = api.Plan(sol, wmap, N, N - 1, _agents(api, W, N), sfc_mode=api.SFC_FROM_POINT, optimize_goal=False, closed_loop=True,
                    safety_samples=2, record_time_step=0.1, **kw)
    plan.reset(W["starts"], W["goals"])
    cls = oracle.make_class(M=M, dim=3, use_sfc=corridors, comm_range=0.0, world_min=W["world_min"], world_max=W["world_max"])
    worst_ratio, failed = 1e9, 0
    for k in range(45):
        plan.step(graph=True)
        torch_cuda.cuda.synchronize()
        st = plan.get(api.PLAN_STATUS)
        failed += int((st != 0).sum())
        assert (plan.get(api.PLAN_VALID)[st == 0] == 1).all(), k
        assert (plan.get(api.PLAN_IN_RANGE) == N - 1).all()
        S = plan.get(api.PLAN_SAFETY)
        worst_ratio = min(worst_ratio, float(S["safety_ratio"].min()))
        # (the figures are taken on the float32 plans: a 2e-7 m rounding of a control point is 20 / dt^2 = 500 times that in acceleration)
        assert S["vel_excess_ratio"].max() <= 1e-5 and S["acc_excess_ratio"].max() <= 2e-4, k
        if k in (3, 15, 30):
            hdr, rows, sfc = plan.get(api.PLAN_HEADER), plan.get(api.PLAN_ROWS).reshape(N, N - 1, M, 6), plan.get(api.PLAN_SFC).reshape(N, M)
            x, obj, info = plan.get(api.PLAN_PLAN).reshape(N, -1), plan.get(api.PLAN_OBJECTIVE), plan.get(api.PLAN_INFO)
            for q in range(N):
                if st[q] != 0:
                    continue
                ag = oracle.make_agent(p0=hdr["p0"][q], v0=hdr["v0"][q], a0=hdr["a0"][q], goal=hdr["goal"][q], next_waypoint=hdr["next_waypoint"][q],
                                       vmax=hdr["vmax"][q], amax=hdr["amax"][q], radius=hdr["radius"][q], nominal_velocity=hdr["nominal_velocity"][q],
                                       n_obs=N - 1)
                lsc = np.zeros((N - 1, M, 6), oracle.LSC_DTYPE)
                lsc["nrm"][..., 0], lsc["nrm"][..., 1], lsc["nrm"][..., 2], lsc["d"] = rows["nx"][q], rows["ny"][q], rows["nz"][q], rows["b"][q]
                box = np.zeros(M, oracle.BOX_DTYPE)
                box["bmin"], box["bmax"] = sfc["bmin"][q], sfc["bmax"][q]
                o = oracle.solve(cls, ag, lsc, box)
                assert o["status"] == 0, (k, q)
                assert abs(o["obj"] - obj[q]) <= 1e-8 * max(1.0, abs(o["obj"])), (k, q, o["obj"], obj[q])
                # (a REMEMBERED point -- flagged: the iteration broke down and the best point it had seen is returned; the objective bar
                # holds and its stationarity is within 1e-8, or 1e-7 with FLOOR_ACCEPTED -- may sit some 1e-5 m along a flat direction:
                # BVC cells leave the z axis of a blocked agent almost free)
                floor = bool(info["flags"][q] & (api.INFO_FLOOR_ACCEPTED | api.INFO_REMEMBERED))
                dxq = np.abs(o["x"] - x[q]).max()
                if dxq > 2e-6 and not floor:
                    # an ordinary OPTIMAL result further than 2e-6 m from the oracle's point must itself be a KKT point of the reference's
                    # row-for-row model at the 1e-8 bar (then the two points are two ends of a flat valley: same objective to 1e-8,
                    # asserted above -- not a solver error)
                    stat, eqv, iqv = H.kkt_from_primal(oracle, cls, ag, lsc, box, x[q])
                    assert stat <= 1e-8 and eqv <= 1e-8 and iqv <= 1e-8, (k, q, dxq, stat, eqv, iqv)
                    floor = True
                assert dxq <= (5e-5 if floor else 2e-6), (k, q, floor)
    assert failed == 0, failed
    assert worst_ratio >= 1.0 - 5e-6, worst_ratio
    state = plan.get(api.PLAN_STATE).reshape(N, 9)
    progress = np.linalg.norm(W["goals"] - W["starts"], axis=1) - np.linalg.norm(W["goals"] - state[:, :3], axis=1)
    assert progress.mean() > 1.5, progress
    assert plan.graph_nodes() >= 8
    plan.close()


def _agents(api, W, N):
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    return ag


def _const_vel_traj(pos, vel, M, dt):
    """Trajectory::planConstVelTraj (reference src/trajectory.cpp:79-91) in numpy: point3d arithmetic, the double clock advancing by
    dt / n after every control point (n + 1 times per segment)."""
    out = np.zeros((len(pos), M, 6, 3))
    time = 0.0
    for m in range(M):
        for i in range(6):
            out[:, m, i, :] = (np.float32(pos) + np.float32(vel) * np.float32(time)).astype(np.float64)
            time += dt / 5
    return out


@pytest.mark.gpu
def test_plan_predicts_a_disturbed_agent_to_stay_where_it_is(api, torch_cuda):
    """checkObstacleDisturbance (reference src/traj_planner.cpp:312-319) inside the chain: an agent whose predicted trajectory starts
    further than reset_threshold from where it is now is predicted to stay there -- for the OTHERS; its own initial trajectory is not
    touched (AgentManager's is_disturbed stays false).  Six logged replans, then agent 3 is moved by 0.32 m: the others' rows equal
    bit for bit those of a plan without the check in which agent 3's previous plan was replaced by hand with "hover at the new
    position", agent 3's own rows equal those of a plan without the check, and the check did change something."""
    g, W, m = _mission()
    N, K0, who = m["N"], 6, 3
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    kw = dict(constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"])
    A = api.Plan(sol, wmap, N, 9, _agents(api, W, N), reset_threshold=0.1, **kw)
    B = api.Plan(sol, wmap, N, 9, _agents(api, W, N), reset_threshold=0.0, **kw)
    D = api.Plan(sol, wmap, N, 9, _agents(api, W, N), reset_threshold=0.0, **kw)
    for p in (A, B, D):
        p.reset(np.array(W["starts"], dtype=np.float64))
        for k in range(K0):
            p.put(api.PLAN_STATE, m["state"][k])
            p.put(api.PLAN_WAYPOINT, m["way"][k])
            p.step()
    torch_cuda.cuda.synchronize()
    assert np.array_equal(A.get(api.PLAN_ROWS), D.get(api.PLAN_ROWS))  # (no disturbance so far: the check is silent)
    state = m["state"][K0].copy()
    state[who, 0] = np.float32(state[who, 0] + 0.25)
    state[who, 1] = np.float32(state[who, 1] - 0.2)
    x = B.get(api.PLAN_PLAN).reshape(N, 2, 60)
    x[who, 0, :], x[who, 1, :] = state[who, 0], state[who, 1]
    B.put(api.PLAN_PLAN, x.reshape(N, -1))
    for p in (A, B, D):
        p.put(api.PLAN_STATE, state)
        p.put(api.PLAN_WAYPOINT, m["way"][K0])
        p.step()
    torch_cuda.cuda.synchronize()
    rows = {n: p.get(api.PLAN_ROWS).reshape(N, -1) for n, p in (("A", A), ("B", B), ("D", D))}
    others = [a for a in range(N) if a != who]
    assert np.array_equal(rows["A"][others], rows["B"][others])
    assert np.array_equal(rows["A"][who], rows["D"][who])
    assert not np.array_equal(rows["A"][others], rows["D"][others]), "agent 3 is nobody's neighbour: the case tests nothing"
    assert (A.get(api.PLAN_IN_RANGE) == D.get(api.PLAN_IN_RANGE)).all()
    for p in (A, B, D):
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pred,init", [("VELOCITY", "VELOCITY"), ("POSITION", "POSITION"), ("VELOCITY", "PREVIOUS_SOLUTION"), ("PREVIOUS_SOLUTION", "POSITION")])
def test_plan_prediction_and_initial_trajectory_modes(api, torch_cuda, pred, init):
    """obstaclePrediction / initialTrajPlanning in their three modes (reference src/traj_planner.cpp:228-253, 360-423): the rows the
    chain generates on its second replan equal, bit for bit, lscqp_generate_constraints_device run agent by agent on trajectories put
    together by hand -- the agent's own entry from its mode, the others' from theirs; constant-velocity trajectories follow
    Trajectory::planConstVelTraj (numpy restatement above), previous-solution ones come from lscqp_shift_traj_device."""
    torch = torch_cuda
    g, W, m = _mission()
    N, M, n_obs, dev = m["N"], 10, 9, torch.device("cuda", 0)
    sol = api.Solver(api.make_desc(M=M, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    mode = dict(PREVIOUS_SOLUTION=api.TRAJ_FROM_PREVIOUS_SOLUTION, POSITION=api.TRAJ_FROM_POSITION, VELOCITY=api.TRAJ_FROM_VELOCITY)
    plan = api.Plan(sol, wmap, N, n_obs, _agents(api, W, N), constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"],
                    prediction_mode=mode[pred], initial_traj_mode=mode[init], reset_threshold=0.0)
    plan.reset(np.array(W["starts"], dtype=np.float64))
    for k in (0, 1, 2):
        plan.put(api.PLAN_STATE, m["state"][k])
        plan.put(api.PLAN_WAYPOINT, m["way"][k])
        plan.step()
    torch.cuda.synchronize()
    k = 6  # (any state may follow with closed_loop = 0: one with the agents on the move)
    state = m["state"][k]
    x_prev, goal_all = plan.get(api.PLAN_PLAN), plan.get(api.PLAN_GOAL)
    plan.put(api.PLAN_STATE, state)
    plan.put(api.PLAN_WAYPOINT, m["way"][k])
    plan.step()
    torch.cuda.synchronize()
    got = plan.get(api.PLAN_ROWS).reshape(N, -1)
    assert (plan.get(api.PLAN_STATUS) >= 0).all()
    # the same rows, put together by hand
    d_prev = torch.from_numpy(x_prev).to(dev)
    d_shift = torch.zeros(N * M * 18, dtype=torch.float64, device=dev)
    sol.shift_traj_device(N, d_prev, d_shift, z_2d=W["z_2d"])
    shifted = d_shift.cpu().numpy().reshape(N, M, 6, 3)
    source = dict(PREVIOUS_SOLUTION=shifted, POSITION=_const_vel_traj(state[:, 0:3], np.zeros((N, 3)), M, 0.2),
                  VELOCITY=_const_vel_traj(state[:, 0:3], state[:, 3:6], M, 0.2))
    d_pos = torch.from_numpy(np.ascontiguousarray(state[:, 0:3])).to(dev)
    d_nbr = torch.zeros(N * n_obs, dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(N, dtype=torch.int32, device=dev)
    sol.select_neighbours_device(N, 0, N, n_obs, 3.0, d_pos, d_nbr, d_cnt)
    d_radius = torch.full((N,), float(W["radius"]), dtype=torch.float64, device=dev)
    d_down = torch.full((N,), 2.0, dtype=torch.float64, device=dev)
    d_goal = torch.from_numpy(goal_all).to(dev)
    for a in range(N):
        traj = source[pred].copy()
        traj[a] = source[init][a]
        d_traj = torch.from_numpy(traj.reshape(-1)).to(dev)
        d_rows = torch.zeros(n_obs * M * 6 * 4, dtype=torch.float64, device=dev)
        sol.generate_constraints_device(api.GEN_CLSC, 1, n_obs, a, d_traj, d_nbr[a * n_obs:(a + 1) * n_obs].contiguous(), d_radius, d_down, d_goal, d_rows)
        torch.cuda.synchronize()
        want = d_rows.cpu().numpy().view(api.ROW_DTYPE)
        assert np.array_equal(got[a], want), (pred, init, a)
    plan.close()


@pytest.mark.gpu
def test_plan_with_a_simulation_step_shorter_than_a_segment(api, oracle, torch_cuda):
    """multisim_time_step < dt (reference src/traj_planner.cpp:413-421): the chain then re-plans from the state at time_step along the
    plan and starts from prev_traj with segment 0 := subSegment(time_step / dt, 1).  Closed loop, 12 replans of the forest10 mission at
    time_step = dt / 2: every QP solves, each replan starts exactly where the previous plan is at time_step (position, velocity,
    acceleration: the QP's equality rows), and the new plan's start of segment 1 stays within millimetres of the previous plan's."""
    import torch

    g, W, m = _mission()
    N = m["N"]
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, 9, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, time_step=0.1, z_2d=W["z_2d"])
    plan.reset(np.array(W["starts"], dtype=np.float64))
    cls = H.oracle_class(oracle, g["params"], use_sfc=True)
    prev_x = None
    for k in range(12):
        plan.put(api.PLAN_WAYPOINT, m["way"][min(k // 2, m["K"] - 1)])
        plan.step(graph=(k >= 3))
        torch.cuda.synchronize()
        assert (plan.get(api.PLAN_STATUS) == 0).all(), (k, plan.get(api.PLAN_STATUS))
        x = plan.get(api.PLAN_PLAN).reshape(N, -1)
        hdr = plan.get(api.PLAN_HEADER)
        nxt = plan.get(api.PLAN_NEXT_STATE).reshape(N, 9)
        for a in range(N):
            pos, vel, acc = oracle.state_at(cls, x[a], 0.1)
            assert np.abs(pos - nxt[a, 0:2]).max() <= 1e-6 and np.abs(vel - nxt[a, 3:5]).max() <= 1e-5 and np.abs(acc - nxt[a, 6:8]).max() <= 1e-3
            if prev_x is not None:  # this replan started where the previous plan was at 0.1 s
                p0, v0, a0 = oracle.state_at(cls, prev_x[a], 0.1)
                assert np.abs(p0 - hdr["p0"][a][:2]).max() <= 1e-6 and np.abs(v0 - hdr["v0"][a][:2]).max() <= 1e-5
                # ... and from the previous plan cut at 0.1 s: the junction to segment 1 is the previous plan's, up to the re-optimisation
                j_prev, _, _ = oracle.state_at(cls, prev_x[a], 0.2)
                j_new, _, _ = oracle.state_at(cls, x[a], 0.1)
                assert np.abs(j_prev - j_new).max() <= 0.05
        prev_x = x
    assert plan.graph_nodes() >= 8
    plan.close()


def test_plan_entry_points_validate_their_arguments(api):
    """No GPU needed: the create call checks its arguments before it touches the device."""
    import ctypes as C

    L = api.lib()
    d = api.PlanDesc()
    h = C.c_void_p()
    assert L.lscqp_plan_create(None, None, C.byref(d), None, C.byref(h)) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_plan_step(None, None) == api.ERR_INVALID_ARGUMENT and L.lscqp_plan_step_graph(None, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_plan_graph_nodes(None) == 0
    L.lscqp_plan_destroy(None)


@pytest.mark.gpu
def test_plan_safety_figures_at_the_end_of_the_step(api, torch_cuda):
    """safety_samples > 0: the chain ends with MultiSyncSimulator::update's safety figures of the new plans (lscqp_safety_metrics_device):
    over the first 30 replans of the logged mission no agent comes closer than the sum of the radii, no limit is exceeded, and the
    closest pair the reference's summary reports (safety ratio 1.02089 for the whole run) is not undercut."""
    import torch

    g, W, m = _mission()
    N = m["N"]
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, 9, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"], safety_samples=2, record_time_step=0.1)
    plan.reset(np.array(W["starts"], dtype=np.float64))
    worst = np.inf
    for k in range(30):
        plan.put(api.PLAN_STATE, m["state"][k])
        plan.put(api.PLAN_WAYPOINT, m["way"][k])
        plan.step(graph=(k >= 2))
        torch.cuda.synchronize()
        saf = plan.get(api.PLAN_SAFETY)
        assert (plan.get(api.PLAN_STATUS) == 0).all()
        assert saf["safety_ratio"].min() >= 1.0 - 5e-6 and saf["vel_excess_ratio"].max() <= 1e-5 and saf["acc_excess_ratio"].max() <= 1e-5, (k, saf)
        worst = min(worst, saf["safety_ratio"].min())
    assert 1.0 <= worst + 5e-6 and worst < 5.0
    assert plan.graph_nodes() >= 9
    plan.close()
    with pytest.raises(api.LscqpError):  # the figures need every agent's new plan on the device
        api.Plan(sol, wmap, 5, 9, ag, n_total=N, constraint_mode=api.GEN_CLSC, z_2d=W["z_2d"], safety_samples=2)


@pytest.mark.gpu
def test_plan_with_an_empty_block_steps_as_a_no_op(api, torch_cuda):
    """A mission of 9 agents over 4 devices is cut into 3 + 3 + 3 + 0 (lscqp_shard_range): the last device's plan owns no agent.  It
    can be created, reset and stepped (eager and graph: nothing is enqueued), and keeps the all-agent buffers the exchange fills."""
    g, W, m = _mission()
    N = 9
    assert [api.shard_range(N, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 0)]
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = _agents(api, W, N)
    empty = api.Plan(sol, wmap, 0, 8, ag, n_total=N, first_agent=9, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, z_2d=W["z_2d"])
    starts = np.array(W["starts"], dtype=np.float64)[:N]
    empty.reset(starts)
    before = empty.get(api.PLAN_PLAN).copy()
    for graph in (False, True, True):
        empty.step(graph=graph)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(before, empty.get(api.PLAN_PLAN)) and empty.graph_nodes() == 0
    assert empty.get(api.PLAN_STATE).reshape(N, 9)[:, :2].tolist() == np.float32(starts[:, :2]).astype(np.float64).tolist()
    empty.close()


@pytest.mark.gpu
def test_update_after_the_graph_capture_reaches_the_replayed_chain(api, torch_cuda):
    """TrajOptimizer::updateParam (reference src/traj_optimizer.cpp:158-160) -> lscqp_update AFTER a plan has captured its graph: the
    class constants travel to the kernels by value, so the captured graph is a snapshot -- the plan notices the handle's generation and
    captures again.  Two plans on ONE handle, one eager and one replaying its graph, stay bit-identical across an update that changes
    the terminal weight; and the update does change the plans (so the comparison is not vacuous).  Same for the tight-warm-start
    clone of the handle that a plan may own."""
    g, W, m = _mission()
    N = m["N"]
    for tight in (False, True):
        sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
        wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
        ag = _agents(api, W, N)
        kw = dict(constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True, closed_loop=True, z_2d=W["z_2d"], tight_warm_start=tight)
        pe, pg = api.Plan(sol, wmap, N, 9, ag, **kw), api.Plan(sol, wmap, N, 9, ag, **kw)
        starts = np.array(W["starts"], dtype=np.float64)
        for p in (pe, pg):
            p.reset(starts)
            p.put(api.PLAN_WAYPOINT, m["way"][0])
            p.step()
            p.put(api.PLAN_WAYPOINT, m["way"][5])
        for _ in range(4):
            pe.step(graph=False)
            pg.step(graph=True)
        torch_cuda.cuda.synchronize()
        assert pg.graph_nodes() > 0 and np.array_equal(pe.get(api.PLAN_PLAN), pg.get(api.PLAN_PLAN))
        x_before = pe.get(api.PLAN_PLAN).copy()
        sol.update(api.make_desc(M=10, dim=2, dt=0.2, w_t=6.0, world_min=W["world_min"], world_max=W["world_max"]))
        for _ in range(3):
            pe.step(graph=False)
            pg.step(graph=True)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(pe.get(api.PLAN_PLAN), pg.get(api.PLAN_PLAN)) and np.array_equal(pe.get(api.PLAN_OBJECTIVE), pg.get(api.PLAN_OBJECTIVE))
        assert (pe.get(api.PLAN_STATUS) == 0).all() and not np.array_equal(x_before, pe.get(api.PLAN_PLAN))
        pe.close()
        pg.close()


@pytest.mark.gpu
def test_update_that_changes_the_shape_under_a_live_plan_is_refused(api, torch_cuda):
    """lscqp_update may change anything a TrajOptimizer::updateParam call can (src/traj_optimizer.cpp:158-160) -- but a plan's buffers and
    launch shapes were sized at lscqp_plan_create.  An update to another segment count (or dimension) must make the next step fail with
    INVALID_ARGUMENT instead of running prepare / commit with the old sizes against the new class; updating back makes the plan usable
    again, with the same bits as a plan that never saw the detour."""
    g, W, m = _mission()
    N = m["N"]
    mk = lambda M: api.make_desc(M=M, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"])  # noqa: E731
    sol, ref = api.Solver(mk(10)), api.Solver(mk(10))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = _agents(api, W, N)
    kw = dict(constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True, closed_loop=True, z_2d=W["z_2d"])
    p, q = api.Plan(sol, wmap, N, 9, ag, **kw), api.Plan(ref, wmap, N, 9, ag, **kw)
    starts = np.array(W["starts"], dtype=np.float64)
    for pl in (p, q):
        pl.reset(starts)
        pl.put(api.PLAN_WAYPOINT, m["way"][0])
        pl.step()
        pl.put(api.PLAN_WAYPOINT, m["way"][5])
        pl.step(graph=True)
    torch_cuda.cuda.synchronize()
    sol.update(mk(5))
    for graph in (False, True):
        with pytest.raises(api.LscqpError) as e:
            p.step(graph=graph)
        assert e.value.code == api.ERR_INVALID_ARGUMENT and "destroy the plan" in str(e.value)
    sol.update(mk(10))
    for _ in range(3):
        p.step(graph=True)
        q.step(graph=True)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(p.get(api.PLAN_PLAN), q.get(api.PLAN_PLAN)) and (p.get(api.PLAN_STATUS) == 0).all()
    p.close()
    q.close()


@pytest.mark.pdip_only
@pytest.mark.gpu
def test_large_plan_carries_its_work_order_and_graph_equals_eager(api, torch_cuda):
    """A plan of 1200 agents -- more than the chip works on at once (lscqp_launch_capacity: 1024 QPs of this class; the corridor kernel's
    throughput build: 1024 agents) -- sorts its QP launch by the previous replan's iteration counts and its corridor launch by the previous
    replan's recorded costs (two more nodes of the chain).  The order decides WHEN an agent's work runs, never its result: two plans of
    the same mission, one stepped eagerly and one through its captured graph, stay bit-identical over six replans; every QP solves."""
    N, M = 1200, 5
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(12), indexing="ij"), -1).reshape(-1, 3)[:N].astype(np.float64)
    starts = g * 1.25 + np.array([-5.5, -5.5, 1.0])
    goals = starts + np.array([0.5, 0.25, 0.0])
    wmin, wmax = [-8.0, -8.0, 0.0], [8.0, 8.0, 16.5]
    sol = api.Solver(api.make_desc(M=M, dim=3, dt=0.2, comm_range=0.0, use_sfc=True, world_min=wmin, world_max=wmax))
    # (a few pillars between the lattice's columns: the corridor launch is sorted by the previous replan's recorded costs as well)
    boxes = [[-4.9 + 2.5 * i, -4.9 + 2.5 * j, 8.0, 0.3, 0.3, 16.0] for i in range(4) for j in range(4)]
    wmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
    kw = dict(constraint_mode=api.GEN_LSC, sfc_mode=api.SFC_FROM_POINT, optimize_goal=False, closed_loop=True)
    pe, pg = api.Plan(sol, wmap, N, 12, ag, **kw), api.Plan(sol, wmap, N, 12, ag, **kw)
    for p in (pe, pg):
        p.reset(starts, goals)
        p.put(api.PLAN_WAYPOINT, goals)
    for k in range(6):
        pe.step(graph=False)
        pg.step(graph=True)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(pe.get(api.PLAN_PLAN), pg.get(api.PLAN_PLAN)) and np.array_equal(pe.get(api.PLAN_OBJECTIVE), pg.get(api.PLAN_OBJECTIVE)), k
        assert (pe.get(api.PLAN_STATUS) == 0).all(), (k, np.bincount(pe.get(api.PLAN_STATUS)))
    it = pe.get(api.PLAN_INFO)["iterations"]
    assert N > sol.launch_capacity(N, 12) > 0
    assert it.max() >= 2 and pg.graph_nodes() >= 10  # (the eight nodes of a small plan's chain + the two sorts)
    pe.close()
    pg.close()
    wmap.close()


@pytest.mark.gpu
def test_persistent_qp_launch_inside_a_captured_graph(api, torch_cuda):
    """300 agents of the M = 10 class: more QPs than the chip works on at once (256: one workgroup per CU), so the plan's QP launch runs
    persistent workgroups over a work queue whose counter is cleared by a memset node of the captured graph, in the order of the previous
    replan's iteration counts.  Graph and eager stay bit-identical over five replans, and a third plan stepped through its own graph at the
    same time (its launches own other counters) does not disturb them."""
    N, M = 300, 10
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:N].astype(np.float64)
    starts = g * 1.5 + np.array([-6.75, -6.75, 1.0])
    goals = starts + np.array([0.75, 0.5, 0.25])
    wmin, wmax = [-9.0, -9.0, 0.0], [9.0, 9.0, 6.0]
    sol = api.Solver(api.make_desc(M=M, dim=3, dt=0.2, comm_range=0.0, use_sfc=False, world_min=wmin, world_max=wmax))
    assert 0 < sol.launch_capacity(N, 10) < N
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
    kw = dict(constraint_mode=api.GEN_LSC, optimize_goal=False, closed_loop=True)
    pe, pg, pc = (api.Plan(sol, None, N, 10, ag, **kw) for _ in range(3))
    for p in (pe, pg, pc):
        p.reset(starts, goals)
        p.put(api.PLAN_WAYPOINT, goals)
    side = torch_cuda.cuda.Stream()
    for k in range(5):
        pe.step(graph=False)
        pg.step(graph=True)
        pc.step(stream=side, graph=True)  # a second graph of the same handle, replayed concurrently on another stream
        torch_cuda.cuda.synchronize()
        assert np.array_equal(pe.get(api.PLAN_PLAN), pg.get(api.PLAN_PLAN)) and np.array_equal(pe.get(api.PLAN_PLAN), pc.get(api.PLAN_PLAN)), k
        assert (pe.get(api.PLAN_STATUS) == 0).all(), (k, np.bincount(pe.get(api.PLAN_STATUS)))
    for p in (pe, pg, pc):
        p.close()

"""The C++ shim (lsc_dr_planner_amd/shim): the reference's TrajOptimizer / CollisionConstraints / Trajectory class
surface over the C ABI.  Host-logic checks run on CPU; solves run on the GPU (-m gpu)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim_exe(api):
    from lsc_dr_planner_amd.shim import build as SB

    return SB.build()


def run(exe, scenario, cwd=None):
    out = subprocess.run([exe, scenario], capture_output=True, text=True, timeout=120, cwd=cwd)
    assert out.returncode == 0, out.stderr
    # (RCCL prints a version banner on stdout when a communicator is initialised: only the JSON lines are the driver's)
    return [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")]


def test_host_logic(shim_exe):
    import torch

    r = run(shim_exe, "host")[0]
    # constructor errors mirror the reference (std::invalid_argument, src/traj_optimizer.cpp:200, :249)
    assert r["threw_n"] and r["threw_dim"] and r["copies_ok"]
    # container: setLSC(oi, m, point, normal, d) fills all n+1 control points (src/collision_constraints.cpp:532-539)
    assert r["obs"] == 3 and r["lsc_d"] == 0.25 and r["lsc_pz"] == 3
    # Box::convertToLSCs: face 2i+1 = (-e_i, d = -box_max(i))  (src/collision_constraints.cpp:37-59)
    assert r["faces"] == 6 and r["face3_d"] == -2 and r["face3_n"] == -1
    assert r["B00"] == 1 and r["B01"] == -5  # include/polynomial.hpp:281-294
    # planConstVelTraj spaces control points dt/n apart across segment borders (src/trajectory.cpp:79-91), so segment 1
    # starts at 6*dt/5: at t = 0.3 the point is p + v*0.34, float32
    assert abs(r["lin"]["p"][0] - (1 + 0.5 * 0.34)) < 1e-6 and abs(r["lin"]["v"][0] - 0.5) < 1e-5
    if not torch.cuda.is_available():
        assert "no CPU fallback" in r["solve_without_gpu"]  # loud failure, never a silent CPU path


@pytest.mark.gpu
def test_reference_log_through_the_shim(shim_exe):
    """forest10_10 first replan through TrajOptimizer::solve with float32 truncation and Trajectory::getStateAt,
    i.e. the same pipeline that wrote the reference log: matches it to the printed digits."""
    g = H.load_golden("kat_log")
    res = {r["scenario"]: r for r in run(shim_exe, "kat")}
    for case in g["cases"]:
        r = res["kat1" if case["agent"] == 0 else "kat2"]
        log = g["agents"][case["agent"]]["states"]
        for key, st in (("t0.1", log[1]), ("t0.2", log[2])):
            for comp in ("p", "v", "a"):
                for got, logged in zip(r[key][comp], st[comp]):
                    # the log prints 6 significant digits (std::ofstream default): agree to one unit of the 6th digit
                    ulp6 = 10.0 ** (np.floor(np.log10(abs(logged))) - 5) if logged != 0 else 1e-9
                    assert abs(got - logged) <= 1.01 * ulp6, (key, comp, got, logged)
        assert abs(r["z"] - 0.6) < 1e-7  # dim == 2: z := world_z_2d (src/traj_optimizer.cpp:78-80)
    assert abs(res["kat1"]["cost"] - 0.2024006128) < 1e-8  # -85.5476 + 7 * 3.5^2


@pytest.mark.gpu
def test_shim_matches_oracle(shim_exe, oracle):
    r = run(shim_exe, "pair")[0]
    assert r["batch_same"]
    cls = oracle.make_class(M=5, dim=3)
    ag = oracle.make_agent(p0=(0, 0, 1), v0=np.float32([0.3, 0.0, 0.05]).astype(float), goal=np.float32([0.5, 0.1, 1]).astype(float),
                           n_obs=2)
    lsc = np.zeros((2, 5, 6), oracle.LSC_DTYPE)
    for oi in range(2):
        lsc["p"][oi] = r["obs"][oi]
        lsc["nrm"][oi] = r["nrm"][oi][:3]
        lsc["d"][oi] = r["nrm"][oi][3]
    sfc = np.zeros(5, oracle.BOX_DTYPE)
    sfc["bmin"] = np.float32([-0.55, -0.65, 0.45]).astype(float)
    sfc["bmax"] = np.float32([0.95, 0.75, 1.65]).astype(float)
    o = oracle.solve(cls, ag, lsc, sfc)
    assert o["status"] == 0
    x = np.array(r["x"])
    assert np.abs(x - o["x"]).max() < 1e-7
    assert abs(r["cost"] - o["obj"]) <= 1e-8 * max(1, abs(o["obj"]))
    # float32 truncation of the returned trajectory (src/traj_optimizer.cpp:74-76)
    xs = x.reshape(3, 5, 6)
    assert np.array_equal(np.float32(r["cp_f32"]), np.float32(xs[:, 2, 3]))


@pytest.mark.gpu
def test_qpfailed_is_thrown(shim_exe, tmp_path):
    """Failure contract of TrajOptimizer::solve (reference src/traj_optimizer.cpp:103-152): QPFAILED is thrown, the model has been
    exported as an LP file under package_path/log, and the conflict is named -- here the LSC row `x >= 50` that no point of the
    world satisfies."""
    os.makedirs(tmp_path / "log")
    r = run(shim_exe, "infeasible", cwd=str(tmp_path))[0]
    assert r["thrown"] == "QPFAILED"
    assert "No solution at mav" in r["conflict"] and "LSC row, oi: 0" in r["conflict"] and r["lp_bytes"] > 5000
    lp = open(tmp_path / "log" / "QPmodel_trajOpt.lp").read()
    assert "Subject To" in lp and "x_4_5" in lp and lp.rstrip().endswith("End")


@pytest.mark.gpu
def test_goal_optimizer_through_the_shim(shim_exe):
    """GoalOptimizer (reference include/goal_optimizer.hpp surface) over lscqp_optimize_goal: the forest10_10 agent-1 case of
    the reference log (goal x = 2.55 from the SFC face), goal == waypoint, and QPFAILED when the
    rows cut the whole segment off (reference src/goal_optimizer.cpp:57-69)."""
    r = run(shim_exe, "goal")[0]
    assert np.allclose(r["goal"], [np.float32(2.55), 2.5, np.float32(0.6)], rtol=0, atol=1e-6)
    assert np.allclose(r["same"], [2.5, 2.5, np.float32(0.6)], rtol=0, atol=1e-7)
    assert r["thrown"] == "QPFAILED"


def test_result_csv_writer_reproduces_reference_log_lines(shim_exe, tmp_path):
    """SimulationResultCsv (shim/include/result_csv.hpp) against the reference's own result log: the states of its first
    three logged rows, fed back through the writer, give the log's header and rows character for character
    (MultiSyncSimulator::saveSimulationResultAsCSV, reference src/multi_sync_simulator.cpp:586-656)."""
    g = H.load_golden("sim_log_states")
    p = tmp_path / "states.txt"
    with open(p, "w") as f:
        f.write("10 3\n")
        for r in range(3):
            for q in range(10):
                vals = [g["t"][r]] + g["pos"][r][q] + g["vel"][r][q] + g["acc"][r][q] + [g["planning_time"][r][q]]
                f.write(" ".join(repr(float(v)) for v in vals) + "\n")
    out = subprocess.run([shim_exe, "csv", str(p)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[:4] == g["raw_lines"]
    # writeStep: header, then samples at future_time = 0 and 0.1 (< time_step 0.2) of constant-velocity plans, t advancing
    f4 = [float(v) for v in lines[4].split(",")]
    want4 = [0, 1, 1, 2, 0.5, 0.5, 0, -0.25, 0, 0, 0, 0.001, 1, 1, 2, 2, 0.5, 0.5, 0, -0.25, 0, 0, 0, 0.002]
    assert np.abs(np.array(f4) - want4).max() < 1e-4  # float32 second differences leave 1e-5 m/s^2 of rounding noise
    f5 = lines[5].split(",")
    assert f5[1] == "1.1" and abs(float(f5[2]) - 1.05) < 1e-6 and abs(float(f5[4]) - 0.475) < 1e-6 and len(lines) == 9  # (+ the header and two rows of the obstacle-column scenario)


def test_result_csv_with_obstacle_columns(shim_exe, tmp_path):
    """mission.on != 0 (reference src/multi_sync_simulator.cpp:603-610, 638-652): every agent's block ends in ",", then "obs_id,t,px,py,pz,size"
    per obstacle; the header names them once.  Two agents, two obstacles, two sample times through writeStep."""
    p = tmp_path / "states.txt"
    p.write_text("1 0\n")
    out = subprocess.run([shim_exe, "csv", str(p)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    hdr, r0, r1 = lines[-3:]
    assert hdr == ",".join(["id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time"] * 2 + ["obs_id,t,px,py,pz,size"] * 2)
    f0, f1 = r0.split(","), r1.split(",")
    assert len(f0) == len(f1) == 2 * 12 + 2 * 6 == len(hdr.split(","))
    assert f0[24:] == ["0", "1", "3", "-1", "1", "0.15", "1", "1", "4", "-1", "1", "0.3"]
    assert f1[24:] == ["0", "1.1", "3.05", "-1", "1", "0.15", "1", "1.1", "4.05", "-1", "1", "0.3"]
    assert f0[:2] == ["0", "1"] and f1[12:14] == ["1", "1.1"]


def test_summary_csv_writer_reproduces_reference_summary_lines(shim_exe, tmp_path):
    """SimulationSummaryCsv against the reference's own log/summary_LSC_10agents.csv: its fields fed back through the writer give the
    description line and both mission rows character for character (column names included: traj_optimization_time, safety_ratio_agent,
    safety_ratio_obs, the excess ratios; reference src/multi_sync_simulator.cpp:658-709), and the append rule writes the description
    once."""
    g = H.load_golden("summary_log_lines")
    for k, want in enumerate(g["raw_lines"][1:]):
        src = tmp_path / ("fields%d.txt" % k)
        src.write_text("\n".join(want.split(",")) + "\n")
        scratch = tmp_path / ("summary%d.csv" % k)
        out = subprocess.run([shim_exe, "summary", str(src), str(scratch)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        assert out.stdout.splitlines() == [g["raw_lines"][0], want]
        assert scratch.read_text().splitlines() == [g["raw_lines"][0], want, want]


@pytest.mark.gpu
def test_corridors_through_the_shim_match_oracle(shim_exe, oracle, tmp_path):
    """CollisionConstraints::initializeSFC / constructSFCFromConvexHull / constructSFCFromPoint over the reference's world
    CSV format (DistanceMap in place of DynamicEDTOctomap): the boxes equal the CPU restatement bit for bit."""
    g = H.load_golden("forest10_world")
    world = tmp_path / "forest10.csv"
    np.savetxt(world, np.array(g["boxes"]), delimiter=",", fmt="%.17g")
    out = subprocess.run([shim_exe, "sfc", str(world)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    res = {r["scenario"]: r for r in (json.loads(l) for l in out.stdout.strip().splitlines())}
    mp = oracle.Map(g["boxes"], g["world_min"], g["world_max"], g["resolution"], g["max_dist"])
    f32 = lambda *v: np.float32(v).astype(np.float64)  # noqa: E731
    sfc = np.zeros((1, 10), oracle.BOX_DTYPE)

    def pts(a, b, c):
        return np.stack([a, b, c])[None]

    def same(name):
        got = np.float32(res[name]["boxes"])  # printed with 9 significant digits: round-trips float32 exactly
        assert np.array_equal(got[:, :3], np.float32(sfc[0]["bmin"])) and np.array_equal(got[:, 3:], np.float32(sfc[0]["bmax"])), name

    p0 = f32(3.0, 2.5, 0.6)
    assert mp.construct_sfc(oracle.SFC_INIT, pts(p0, p0, p0), 0.15, sfc)[0] == 1
    same("sfc_init")
    assert abs(res["sfc_init"]["boxes"][0][0] - 2.55) <= 1e-6  # the face the reference's log pins
    mp.construct_sfc(oracle.SFC_FROM_HULL, pts(f32(2.8, 2.5, 0.6), f32(2.55, 2.5, 0.6), f32(2.5, 2.5, 0.6)), 0.15, sfc)
    same("sfc_hull")
    mp.construct_sfc(oracle.SFC_FROM_POINT, pts(f32(2.7, 2.4, 0.6), f32(-3.0, -2.5, 0.6), f32(-3.0, -2.5, 0.6)), 0.15, sfc)
    same("sfc_point")
    assert res["sfc_invalid"]["threw"] is True  # std::invalid_argument("Invalid initial SFC"), :377-379


@pytest.mark.gpu
def test_solve_batch_over_a_communicator(shim_exe):
    """TrajOptimizer::solveBatch with a communicator over every visible device (lscqp_solve_batch_sharded through the C++ class
    surface; one device on the test box): same results as without, bit for bit; RCCL initialised by the C++ host itself."""
    r = run(shim_exe, "sharded")[0]
    assert r["n"] == 300 and r["ok"] == 300 and r["bit_identical"]
    assert r["devices"] >= 1 and r["devices_used"] == min(r["devices"], 300 // 64) == r["devices_for_300"]
    assert "rccl" in r["backend"]


@pytest.mark.gpu
def test_dynamic_obstacle_rows_have_a_free_slack(shim_exe):
    """Reference src/traj_optimizer.cpp:272-283, 423-425: the rows of an obstacle in the dynamic-obstacle set carry a slack in
    (-inf, 0] with no cost, i.e. they never bind: the optimum is the one of the QP without that obstacle, and differs from the
    hard-row optimum when the obstacle's rows are active."""
    r = run(shim_exe, "dynamic_obstacle")[0]
    assert abs(r["slack"] - r["without"]) <= 1e-9 * max(1.0, abs(r["without"]))
    assert r["hard"] > r["without"] + 1e-6


@pytest.mark.gpu
def test_plan_chain_from_a_cpp_host_equals_the_python_binding(shim_exe, api, tmp_path):
    """INTEGRATION.md section 9 compiled and run: a C++ host drives lscqp_plan_* (world CSV -> map -> plan -> 20 closed-loop replans of
    the forest10 mission through the captured hipGraph, safety figures included).  The final plans give the same checksum as the
    same mission flown through the Python binding, the first agent's final state is identical."""
    import torch

    g = H.load_golden("forest10_world")
    world = tmp_path / "forest10.csv"
    np.savetxt(world, np.array(g["boxes"]), delimiter=",", fmt="%.17g")
    mission = tmp_path / "mission.txt"
    np.savetxt(mission, np.c_[np.array(g["starts"]), np.array(g["goals"])], fmt="%.17g")
    out = subprocess.run([shim_exe, "plan", str(world), str(mission), "20"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    r = [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")][0]
    assert "error" not in r, r
    assert r["agents"] == 10 and r["failed"] == 0 and r["graph_nodes"] >= 9 and r["worst_safety_ratio"] >= 1.0 - 5e-6
    N = 10
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=g["world_min"], world_max=g["world_max"]))
    wmap = api.WorldMap(g["boxes"], g["world_min"], g["world_max"], g["resolution"], g["max_dist"])
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, 9, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=g["starts"][0][2],
                    safety_samples=2, record_time_step=0.1)
    starts, goals = np.array(g["starts"], dtype=np.float64), np.array(g["goals"], dtype=np.float64)
    way = starts.copy()
    d = goals[:, :2] - starts[:, :2]
    way[:, :2] = np.float32(starts[:, :2] + np.where(d > 1e-6, 0.5, np.where(d < -1e-6, -0.5, 0.0)))
    plan.reset(starts)
    for k in range(20):
        plan.put(api.PLAN_WAYPOINT, starts if k == 0 else way)
        plan.step(graph=True)
        torch.cuda.synchronize()
    x = plan.get(api.PLAN_PLAN)
    assert abs(float(np.sum(x)) - r["plan_sum"]) <= 1e-9  # (the two sums run in different orders)
    st = plan.get(api.PLAN_STATE).reshape(N, 9)
    assert np.array_equal(np.float32(st[0, :3]), np.float32(r["state0"]))
    plan.close()

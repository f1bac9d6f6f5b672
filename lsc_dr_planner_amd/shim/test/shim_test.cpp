// Test driver of the C++ shim: exercises the reference's class surface (TrajOptimizer / CollisionConstraints /
// Trajectory) end to end and prints one JSON object per scenario for tests/test_shim.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <goal_optimizer.hpp>
#include <traj_optimizer.hpp>
#include <result_csv.hpp>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

using namespace DynamicPlanning;

static Agent make_agent(point3d p, point3d goal, point3d wp) {
    Agent a;
    a.id = 0;
    a.cid = 1;
    a.current_state.position = p;
    a.current_state.velocity = point3d(0, 0, 0);
    a.current_state.acceleration = point3d(0, 0, 0);
    a.start_point = p;
    a.desired_goal_point = goal;
    a.current_goal_point = goal;
    a.next_waypoint = wp;
    a.max_vel = {1.0, 1.0, 1.0};
    a.max_acc = {2.0, 2.0, 2.0};
    a.radius = 0.15;
    a.downwash = 2.0;
    a.nominal_velocity = 1.0;
    a.collision_alert = false;
    return a;
}

static void print_state(const char* key, const State& s, bool comma) {
    printf("\"%s\": {\"p\": [%.9g, %.9g, %.9g], \"v\": [%.9g, %.9g, %.9g], \"a\": [%.9g, %.9g, %.9g]}%s", key, s.position.x(),
           s.position.y(), s.position.z(), s.velocity.x(), s.velocity.y(), s.velocity.z(), s.acceleration.x(),
           s.acceleration.y(), s.acceleration.z(), comma ? ", " : "");
}

// forest10_10 first replan, launch/simulation.launch parameters (dim 2, M 10): KAT1 (agent 0) and KAT2 (agent 1)
static int scenario_kat(bool gpu) {
    Param param;
    param.world_dimension = 2;
    param.M = 10;
    param.world_z_2d = 0.6;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    for (int kat = 1; kat <= 2; kat++) {
        param.world_use_octomap = (kat == 2);
        TrajOptimizer opt(param, mission, B);
        CollisionConstraints cons(param, mission);
        cons.initializeLSC(0);
        Agent a = kat == 1 ? make_agent(point3d(4, 0, 0.6f), point3d(3.5f, 0, 0.6f), point3d(3.5f, 0, 0.6f))
                           : make_agent(point3d(3, 2.5f, 0.6f), point3d(2.55f, 2.5f, 0.6f), point3d(2.5f, 2.5f, 0.6f));
        if (kat == 2)
            for (int m = 0; m < param.M; m++) cons.setSFC(m, Box(point3d(2.55f, -5, 0), point3d(5, 5, 2.5f)));
        traj_t init(param.M, param.n, param.dt);
        init.planConstVelTraj(a.current_state.position, point3d(0, 0, 0));
        if (!gpu) {
            printf("{\"scenario\": \"kat%d\", \"constructed\": true}\n", kat);
            continue;
        }
        TrajOptResult r = opt.solve(a, cons, init, true);
        printf("{\"scenario\": \"kat%d\", \"cost\": %.12g, \"iters\": %d, ", kat, r.total_qp_cost, opt.lastIterations());
        print_state("t0.1", r.desired_traj.getStateAt(0.1), true);
        print_state("t0.2", r.desired_traj.getStateAt(0.2), true);
        printf("\"z\": %.9g, \"cp_last\": [%.9g, %.9g]}\n", r.desired_traj[3][2].z(), r.desired_traj.lastPoint().x(), r.desired_traj.lastPoint().y());
    }
    return 0;
}

// one neighbour ahead, dim 3, M 5: prints inputs and the raw fp64 solution so the test can feed the oracle
static int scenario_pair() {
    Param param;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    TrajOptimizer opt(param, mission, B);
    CollisionConstraints cons(param, mission);
    Agent a = make_agent(point3d(0, 0, 1), point3d(0.5f, 0.1f, 1), point3d(0.5f, 0.1f, 1));
    a.current_state.velocity = point3d(0.3f, 0.0f, 0.05f);
    cons.initializeLSC(2);
    point3d obs0(0.45f, 0.05f, 1.0f), obs1(-0.2f, 0.5f, 1.2f);
    for (int m = 0; m < param.M; m++) {
        points_t cps(param.n + 1, obs0);
        point3d nrm = (a.current_state.position - obs0).normalized();
        std::vector<double> ds(param.n + 1);
        for (int i = 0; i <= param.n; i++) ds[i] = 0.5 * (0.3 + (a.current_state.position - obs0).dot(nrm));
        cons.setLSC(0, m, cps, nrm, ds);
        cons.setLSC(1, m, obs1, (a.current_state.position - obs1).normalized(), 0.31);
        cons.setSFC(m, Box(point3d(-0.55f, -0.65f, 0.45f), point3d(0.95f, 0.75f, 1.65f)));
    }
    traj_t init(param.M, param.n, param.dt);
    init.planConstVelTraj(a.current_state.position, a.current_state.velocity);
    TrajOptResult r = opt.solve(a, cons, init, true);
    std::vector<double> raw = opt.lastRawSolution();
    // the same item twice through the additive batch call
    std::vector<TrajOptimizer::BatchItem> items(2, TrajOptimizer::BatchItem{&a, &cons, &init});
    std::vector<TrajOptResult> res;
    std::vector<bool> ok;
    opt.solveBatch(items, res, ok);
    bool same = ok[0] && ok[1] && res[0].total_qp_cost == r.total_qp_cost && res[1].total_qp_cost == r.total_qp_cost;
    printf("{\"scenario\": \"pair\", \"cost\": %.15g, \"batch_same\": %s, \"obs\": [[%.9g, %.9g, %.9g], [%.9g, %.9g, %.9g]], \"nrm\": [",
           r.total_qp_cost, same ? "true" : "false", obs0.x(), obs0.y(), obs0.z(), obs1.x(), obs1.y(), obs1.z());
    for (int oi = 0; oi < 2; oi++) {
        LSC l = cons.getLSC(oi, 1, 4);
        printf("[%.9g, %.9g, %.9g, %.17g]%s", l.normal_vector.x(), l.normal_vector.y(), l.normal_vector.z(), l.d, oi ? "" : ", ");
    }
    printf("], \"x\": [");
    for (size_t i = 0; i < raw.size(); i++) printf("%.17g%s", raw[i], i + 1 < raw.size() ? ", " : "");
    printf("], \"cp_f32\": [%.9g, %.9g, %.9g]}\n", r.desired_traj[2][3].x(), r.desired_traj[2][3].y(), r.desired_traj[2][3].z());
    return 0;
}

static int scenario_infeasible() {
    Param param;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    TrajOptimizer opt(param, mission, B);
    CollisionConstraints cons(param, mission);
    Agent a = make_agent(point3d(0, 0, 1), point3d(0.5f, 0, 1), point3d(0.5f, 0, 1));
    cons.initializeLSC(1);
    for (int m = 0; m < param.M; m++) {
        cons.setLSC(0, m, point3d(0, 0, 1), point3d(1, 0, 0), 50.0);  // x >= 50: outside the world
        cons.setSFC(m, Box(point3d(-1, -1, 0.5f), point3d(1, 1, 1.5f)));
    }
    traj_t init(param.M, param.n, param.dt);
    init.planConstVelTraj(a.current_state.position, point3d(0, 0, 0));
    try {
        opt.solve(a, cons, init, true);
        printf("{\"scenario\": \"infeasible\", \"thrown\": \"nothing\"}\n");
    } catch (PlanningReport r) {
        // the caller's fail-safe (src/traj_planner.cpp:767-797): result.desired_traj = initial_traj.  Before throwing the optimizer has
        // exported the model (log/QPmodel_trajOpt.lp under param.package_path, src/traj_optimizer.cpp:45,103) and named the conflict.
        FILE* lp = fopen((param.package_path + "/log/QPmodel_trajOpt.lp").c_str(), "r");
        long lp_bytes = 0;
        if (lp) {
            fseek(lp, 0, SEEK_END);
            lp_bytes = ftell(lp);
            fclose(lp);
        }
        printf("{\"scenario\": \"infeasible\", \"thrown\": \"%s\", \"conflict\": \"%s\", \"lp_bytes\": %ld}\n",
               r == PlanningReport::QPFAILED ? "QPFAILED" : "other", opt.lastConflict().c_str(), lp_bytes);
    }
    return 0;
}

static int scenario_host() {
    // constructor validation mirrors the reference's exceptions; no device needed
    Param param;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    bool threw_n = false, threw_dim = false;
    try {
        Param p = param;
        p.n = 4;
        TrajOptimizer o(p, mission, B);
    } catch (const std::invalid_argument& e) {
        threw_n = std::string(e.what()).find("only n=5, phi=3") != std::string::npos;
    }
    try {
        Param p = param;
        p.world_dimension = 4;
        TrajOptimizer o(p, mission, B);
    } catch (const std::invalid_argument&) {
        threw_dim = true;
    }
    TrajOptimizer opt(param, mission, B);
    Param p2 = param;
    p2.planner_mode = PlannerMode::DLSC;
    opt.updateParam(p2);
    // copyable like the reference's class: a copy (and an assigned-to object) owns its own handle and outlives the original
    bool copies_ok = false;
    {
        TrajOptimizer* first = new TrajOptimizer(param, mission, B);
        TrajOptimizer copy(*first);
        TrajOptimizer assigned(p2, mission, B);
        assigned = *first;
        delete first;
        copy.updateParam(p2);
        assigned.updateParam(param);
        copies_ok = true;
    }
    // container semantics
    CollisionConstraints cons(param, mission);
    cons.initializeLSC(3);
    cons.setLSC(2, 4, point3d(1, 2, 3), point3d(0, 0, 1), 0.25);
    LSC l = cons.getLSC(2, 4, 5);
    const size_t n_obs_seen = cons.getObsSize();
    Box box(point3d(-1, -2, -3), point3d(1, 2, 3));
    LSCs f = box.convertToLSCs(3);
    // trajectory evaluation: a straight constant-velocity line
    traj_t tr(param.M, param.n, param.dt);
    tr.planConstVelTraj(point3d(1, 1, 1), point3d(0.5f, 0, -0.25f));
    State s = tr.getStateAt(0.3);
    std::string no_device;
    try {
        Agent a = make_agent(point3d(0, 0, 1), point3d(0.5f, 0, 1), point3d(0.5f, 0, 1));
        cons.initializeLSC(0);
        for (int m = 0; m < param.M; m++) cons.setSFC(m, box);
        opt.solve(a, cons, tr, true);
        no_device = "solved";
    } catch (const std::runtime_error& e) {
        no_device = e.what();
    } catch (PlanningReport) {
        no_device = "QPFAILED";
    }
    printf("{\"scenario\": \"host\", \"threw_n\": %s, \"threw_dim\": %s, \"obs\": %zu, \"lsc_d\": %.9g, \"lsc_pz\": %.9g, "
           "\"faces\": %zu, \"face3_d\": %.9g, \"face3_n\": %.9g, \"B00\": %.9g, \"B01\": %.9g, ",
           threw_n ? "true" : "false", threw_dim ? "true" : "false", n_obs_seen, l.d, l.obs_control_point.z(), f.size(),
           f[3].d, f[3].normal_vector.y(), B(0, 0), B(0, 1));
    print_state("lin", s, true);
    printf("\"copies_ok\": %s, \"solve_without_gpu\": \"%s\"}\n", copies_ok ? "true" : "false", no_device.c_str());
    return 0;
}

// GoalOptimizer of forest10_10's agent 1: waypoint x = 2.5, current goal x = 3.0, the SFC's -x face
// at 2.55 -> goal x = 2.55; then an LSC row that cuts the whole segment off -> QPFAILED like the reference
static int scenario_goal() {
    Param param;
    param.world_dimension = 2;
    param.M = 10;
    param.world_z_2d = 0.6;
    param.world_use_octomap = true;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    GoalOptimizer gopt(param, mission);
    CollisionConstraints cons(param, mission);
    cons.initializeLSC(1);
    for (int m = 0; m < param.M; m++) cons.setSFC(m, Box(point3d(2.55f, -5, 0), point3d(5, 5, 2.5f)));
    Agent a = make_agent(point3d(3, 2.5f, 0.6f), point3d(3.0f, 2.5f, 0.6f), point3d(2.5f, 2.5f, 0.6f));
    // obstacle far behind: its row n = (+1,0,0), p = (0,2.5), d = 1 holds on the whole segment -> the SFC face decides
    std::vector<double> d(param.n + 1, 1.0);
    points_t obs(param.n + 1, point3d(0, 2.5f, 0.6f));
    for (int m = 0; m < param.M; m++) cons.setLSC(0, m, obs, point3d(1, 0, 0), d);
    point3d g = gopt.solve(a, cons, point3d(3.0f, 2.5f, 0.6f), point3d(2.5f, 2.5f, 0.6f));
    point3d same = gopt.solve(a, cons, point3d(2.5f, 2.5f, 0.6f), point3d(2.5f, 2.5f, 0.6f));
    // now the obstacle row demands x >= 3.5: infeasible on [2.5, 3.0]
    std::vector<double> d2(param.n + 1, 3.5);
    for (int m = 0; m < param.M; m++) cons.setLSC(0, m, obs, point3d(1, 0, 0), d2);
    const char* thrown = "nothing";
    try {
        gopt.solve(a, cons, point3d(3.0f, 2.5f, 0.6f), point3d(2.5f, 2.5f, 0.6f));
    } catch (PlanningReport r) {
        thrown = r == PlanningReport::QPFAILED ? "QPFAILED" : "other";
    }
    printf("{\"scenario\": \"goal\", \"goal\": [%.9g, %.9g, %.9g], \"same\": [%.9g, %.9g, %.9g], \"thrown\": \"%s\"}\n", g.x(), g.y(),
           g.z(), same.x(), same.y(), same.z(), thrown);
    return 0;
}

// result log writer: states from a text file (qn, rows, then rows x qn x (t, p, v, a, planning_time)) -> CSV on stdout,
// followed by two rows written through writeStep from constant-velocity trajectories
static int scenario_csv(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return 2;
    int qn = 0, rows = 0;
    if (fscanf(f, "%d %d", &qn, &rows) != 2) return 2;
    SimulationResultCsv w(std::cout, (size_t)qn);
    w.writeHeader();
    for (int r = 0; r < rows; r++) {
        std::vector<State> st((size_t)qn);
        std::vector<double> pt((size_t)qn);
        double t = 0;
        for (int q = 0; q < qn; q++) {
            double v[11];
            for (double& x : v)
                if (fscanf(f, "%lf", &x) != 1) return 2;
            t = v[0];
            st[q].position = point3d((float)v[1], (float)v[2], (float)v[3]);
            st[q].velocity = point3d((float)v[4], (float)v[5], (float)v[6]);
            st[q].acceleration = point3d((float)v[7], (float)v[8], (float)v[9]);
            pt[q] = v[10];
        }
        w.writeRow(t, st, pt);
    }
    fclose(f);
    std::vector<traj_t> trajs;
    for (int q = 0; q < 2; q++) {
        traj_t tr(5, 5, 0.2);
        tr.planConstVelTraj(point3d(1.0f + q, 2.0f, 0.5f), point3d(0.5f, 0.0f, -0.25f));
        trajs.push_back(tr);
    }
    SimulationResultCsv w2(std::cout, 2);
    w2.writeStep(1.0, 0.2, 0.1, trajs, {0.001, 0.002});
    // a mission with two obstacles (mission.on = 2): the agents' columns end in "," and the obstacle columns follow (:603-610, :638-652)
    SimulationResultCsv w3(std::cout, 2, 2);
    w3.writeHeader();
    w3.writeStep(1.0, 0.2, 0.1, trajs, {0.001, 0.002}, [](size_t oi, double ft) {
        ObstacleSample o;
        o.px = 3.0 + (double)oi + 0.5 * ft, o.py = -1.0, o.pz = 1.0, o.size = 0.15 * (double)(oi + 1);
        return o;
    });
    return 0;
}

// the mission summary: fields from a text file (one token per field, the reference's order) -> description + row on stdout, then the
// append rule on a scratch file (description only into a new or empty file)
static int scenario_summary(const char* path, const char* scratch) {
    std::ifstream f(path);
    if (!f) return 2;
    SimulationSummary s;
    f >> s.start_time >> s.total_flight_time >> s.total_flight_distance >> s.safety_ratio_agent >> s.safety_ratio_obs >> s.vel_excess_ratio >>
        s.acc_excess_ratio >> s.mapf_time_average >> s.mapf_time_min >> s.mapf_time_max >> s.planning_time_average >> s.planning_time_min >>
        s.planning_time_max >> s.initial_traj_planning_time >> s.obstacle_prediction_time >> s.goal_planning_time >> s.lsc_generation_time >>
        s.sfc_generation_time >> s.traj_optimization_time >> s.mission_file_name >> s.world_file_name >> s.planner_mode >> s.goal_mode >> s.mapf_mode >>
        s.communication_range >> s.world_dimension >> s.M >> s.dt;
    if (!f) return 2;
    SimulationSummaryCsv::writeDescription(std::cout);
    SimulationSummaryCsv::writeRow(std::cout, s);
    if (!SimulationSummaryCsv::append(scratch, s) || !SimulationSummaryCsv::append(scratch, s)) return 3;
    return 0;
}

// corridors through the reference's CollisionConstraints surface: world CSV -> DistanceMap -> initializeSFC ->
// constructSFCFromConvexHull -> constructSFCFromPoint, boxes printed for tests/test_shim.py
static int scenario_sfc(const char* world_csv) {
    Param param;
    param.M = 10;
    param.world_dimension = 2;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    CollisionConstraints cc(param, mission);
    auto dm = std::make_shared<DistanceMap>(std::string(world_csv), mission.world_min, mission.world_max, 0.1);
    dm->prepare(0.15);  // (the free-space table: the boxes below must be the ones the plain map gives)
    cc.setDistmap(dm);
    auto show = [&](const char* name) {
        printf("{\"scenario\": \"%s\", \"boxes\": [", name);
        for (int m = 0; m < param.M; m++) {
            const Box b = cc.getSFC(m);
            printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", m ? ", " : "", b.box_min.x(), b.box_min.y(), b.box_min.z(), b.box_max.x(),
                   b.box_max.y(), b.box_max.z());
        }
        printf("]}\n");
    };
    cc.initializeSFC(point3d(3.0f, 2.5f, 0.6f), 0.15);
    show("sfc_init");
    cc.constructSFCFromConvexHull({point3d(2.8f, 2.5f, 0.6f), point3d(2.55f, 2.5f, 0.6f)}, point3d(2.5f, 2.5f, 0.6f), 0.15);
    show("sfc_hull");
    cc.constructSFCFromPoint(point3d(2.7f, 2.4f, 0.6f), point3d(-3.0f, -2.5f, 0.6f), 0.15);
    show("sfc_point");
    bool threw = false;
    try {
        cc.initializeSFC(point3d(2.2f, 2.1f, 0.6f), 0.15);  // inside the obstacle at (2.18, 2.10)
    } catch (const std::invalid_argument&) {
        threw = true;
    }
    printf("{\"scenario\": \"sfc_invalid\", \"threw\": %s}\n", threw ? "true" : "false");
    return 0;
}

// Multi-GPU behind the reference's class surface: TrajOptimizer::solveBatch with a communicator over every visible device
// (G = 1 on a single-GPU box: same code path, one block), against the same batch without a communicator.
static int scenario_sharded() {
    Param param;
    Mission mission;
    mission.world_min = point3d(-40, -40, 0);
    mission.world_max = point3d(40, 40, 6);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    TrajOptimizer opt(param, mission, B);
    const int N = 300;
    std::vector<Agent> agents;
    std::vector<std::unique_ptr<CollisionConstraints>> cons;
    std::vector<traj_t> inits;
    for (int q = 0; q < N; q++) {
        const float x = -30.0f + 2.0f * (q % 30), y = -9.0f + 2.0f * (q / 30), z = 1.0f + 0.1f * (q % 7);
        const point3d p(x, y, z), wp(x + 0.5f * ((q % 3) - 1), y + 0.5f * ((q % 5) / 2 - 1), z);
        Agent a = make_agent(p, wp, wp);
        a.current_state.velocity = point3d(0.1f * (q % 4), -0.05f * (q % 3), 0.0f);
        agents.push_back(a);
        cons.emplace_back(new CollisionConstraints(param, mission));
        cons.back()->initializeLSC(1);
        const point3d obs(x + 0.6f, y + 0.1f * (q % 5), z);
        for (int m = 0; m < param.M; m++) {
            cons.back()->setLSC(0, m, obs, (p - obs).normalized(), 0.31);
            cons.back()->setSFC(m, Box(point3d(x - 1.5f, y - 1.5f, 0.4f), point3d(x + 1.5f, y + 1.5f, 3.0f)));
        }
        traj_t init(param.M, param.n, param.dt);
        init.planConstVelTraj(a.current_state.position, a.current_state.velocity);
        inits.push_back(init);
    }
    std::vector<TrajOptimizer::BatchItem> items(N);
    for (int q = 0; q < N; q++) items[q] = TrajOptimizer::BatchItem{&agents[q], cons[q].get(), &inits[q]};
    std::vector<TrajOptResult> r0, r1;
    std::vector<bool> ok0, ok1;
    opt.solveBatch(items, r0, ok0);
    const std::vector<double> raw0 = opt.lastRawSolution();
    lscqp_comm comm = nullptr;
    if (lscqp_comm_create(0, nullptr, &comm) != LSCQP_OK) {
        printf("{\"scenario\": \"sharded\", \"error\": \"%s\"}\n", lscqp_last_error());
        return 1;
    }
    lscqp_comm_set_min_agents_per_device(comm, 64);
    TrajOptimizer::setCommunicator(comm);
    opt.solveBatch(items, r1, ok1);
    const std::vector<double> raw1 = opt.lastRawSolution();
    TrajOptimizer::setCommunicator(nullptr);
    int n_ok = 0;
    bool same = raw0.size() == raw1.size();
    for (int q = 0; q < N; q++) n_ok += (ok0[q] && ok1[q]) ? 1 : 0;
    for (size_t i = 0; same && i < raw0.size(); i++) same = raw0[i] == raw1[i];
    printf("{\"scenario\": \"sharded\", \"devices\": %d, \"devices_used\": %d, \"devices_for_300\": %d, \"backend\": \"%s\", \"n\": %d, "
           "\"ok\": %d, \"bit_identical\": %s}\n",
           lscqp_comm_size(comm), opt.lastDevicesUsed(), lscqp_comm_devices_for(comm, N), lscqp_comm_backend(comm), N, n_ok,
           same ? "true" : "false");
    lscqp_comm_destroy(comm);
    return 0;
}

// An obstacle in the dynamic-obstacle set has a free slack on its rows in the reference (src/traj_optimizer.cpp:272-283,
// 423-425, no cost term): the QP is the one without those rows.
static int scenario_dynamic_obstacle() {
    Param param;
    Mission mission;
    mission.world_min = point3d(-5, -5, 0);
    mission.world_max = point3d(5, 5, 2.5);
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    TrajOptimizer opt(param, mission, B);
    Agent a = make_agent(point3d(0, 0, 1), point3d(0.5f, 0.1f, 1), point3d(0.5f, 0.1f, 1));
    const point3d obs0(0.45f, 0.05f, 1.0f), obs1(-0.2f, 0.5f, 1.2f);
    auto fill = [&](CollisionConstraints& c, bool with0) {
        c.initializeLSC(with0 ? 2 : 1);
        for (int m = 0; m < param.M; m++) {
            int oi = 0;
            if (with0) c.setLSC(oi++, m, obs0, (a.current_state.position - obs0).normalized(), 0.35);
            c.setLSC(oi, m, obs1, (a.current_state.position - obs1).normalized(), 0.31);
            c.setSFC(m, Box(point3d(-0.55f, -0.65f, 0.45f), point3d(0.95f, 0.75f, 1.65f)));
        }
    };
    CollisionConstraints hard(param, mission), slack(param, mission), without(param, mission);
    fill(hard, true);
    fill(slack, true);
    slack.markDynamicObstacle(0);
    fill(without, false);
    traj_t init(param.M, param.n, param.dt);
    init.planConstVelTraj(a.current_state.position, a.current_state.velocity);
    const double c_hard = opt.solve(a, hard, init, true).total_qp_cost;
    const double c_slack = opt.solve(a, slack, init, true).total_qp_cost;
    const double c_without = opt.solve(a, without, init, true).total_qp_cost;
    printf("{\"scenario\": \"dynamic_obstacle\", \"hard\": %.15g, \"slack\": %.15g, \"without\": %.15g}\n", c_hard, c_slack, c_without);
    return 0;
}

// Segment::subSegment (src/trajectory.cpp:15-49): the piece [0.5, 1] of a segment
static int scenario_subsegment() {
    Param param;
    Eigen::MatrixXd B, B_inv;
    buildBernsteinBasis(param.n, B, B_inv);
    Segment<point3d> s;
    s.segment_time = 0.2;
    const float pts[6][3] = {{0.0f, 0.0f, 1.0f}, {0.1f, 0.02f, 1.0f}, {0.25f, 0.1f, 1.05f}, {0.45f, 0.3f, 1.1f}, {0.6f, 0.55f, 1.2f}, {0.7f, 0.9f, 1.25f}};
    for (auto& p : pts) s.control_points.push_back(point3d(p[0], p[1], p[2]));
    Segment<point3d> h = s.subSegment(0.5, 1.0, B, B_inv);
    printf("{\"scenario\": \"subsegment\", \"segment_time\": %.17g, \"cp\": [", h.segment_time);
    for (int i = 0; i < 6; i++) printf("[%.9g, %.9g, %.9g]%s", h[i].x(), h[i].y(), h[i].z(), i < 5 ? ", " : "");
    printf("]}\n");
    return 0;
}

// The whole replan as one chain of device work from a C++ host: world CSV + mission file (one line per
// agent: start x y z, goal x y z) -> map, plan, `replans` closed-loop replans through the captured hipGraph with the waypoint one grid
// step towards the goal.  Prints the statuses, the worst safety ratio and a fingerprint of the final plans.
static int scenario_plan(const char* world_csv, const char* mission_txt, int replans) {
    std::vector<double> starts, goals;
    {
        FILE* f = fopen(mission_txt, "r");
        if (!f) return 2;
        double v[6];
        while (fscanf(f, "%lf %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6) {
            starts.insert(starts.end(), v, v + 3);
            goals.insert(goals.end(), v + 3, v + 6);
        }
        fclose(f);
    }
    const int n = (int)(starts.size() / 3);
    lscqp_class_desc cd;
    memset(&cd, 0, sizeof cd);
    cd.M = 10, cd.n = 5, cd.phi = 3, cd.phi_n = 1, cd.dim = 2;
    cd.planner_mode = LSCQP_PLANNER_LSC, cd.use_sfc = 1;
    cd.dt = 0.2, cd.control_input_weight = 0.01, cd.terminal_weight = 1.0, cd.communication_range = 3.0;
    const double wmin[3] = {-5, -5, 0}, wmax[3] = {5, 5, 2.5};
    for (int k = 0; k < 3; k++) cd.world_min[k] = wmin[k], cd.world_max[k] = wmax[k];
    lscqp_handle h = nullptr;
    lscqp_map map = nullptr;
    lscqp_plan plan = nullptr;
    auto fail = [&](const char* what) {
        printf("{\"scenario\": \"plan\", \"error\": \"%s: %s\"}\n", what, lscqp_last_error());
        return 1;
    };
    if (lscqp_create(&cd, &h) != LSCQP_OK) return fail("lscqp_create");
    if (lscqp_map_create_from_csv(world_csv, wmin, wmax, 0.1, 1.0, &map) != LSCQP_OK) return fail("lscqp_map_create_from_csv");
    lscqp_plan_desc pd;
    memset(&pd, 0, sizeof pd);
    pd.n_agents = pd.n_total = n;
    pd.n_obs = n - 1 < lscqp_max_obstacles(h) ? n - 1 : lscqp_max_obstacles(h);
    pd.constraint_mode = LSCQP_GEN_CLSC, pd.sfc_mode = LSCQP_SFC_FROM_HULL, pd.optimize_goal = 1, pd.closed_loop = 1;
    pd.safety_samples = 2, pd.record_time_step = 0.1;
    pd.time_step = 0.2, pd.z_2d = starts[2];
    std::vector<lscqp_agent_param> ap(n);
    for (auto& a : ap) {
        a.radius = 0.15, a.downwash = 2.0, a.nominal_velocity = 1.0;
        for (int k = 0; k < 3; k++) a.max_vel[k] = 1.0, a.max_acc[k] = 2.0;
    }
    if (lscqp_plan_create(h, map, &pd, ap.data(), &plan) != LSCQP_OK) return fail("lscqp_plan_create");
    if (lscqp_plan_reset(plan, starts.data(), nullptr) != LSCQP_OK) return fail("lscqp_plan_reset");
    std::vector<double> way(starts);
    for (int a = 0; a < n; a++)
        for (int k = 0; k < 2; k++) {
            const double d = goals[3 * a + k] - starts[3 * a + k];
            way[3 * a + k] = (double)(float)(starts[3 * a + k] + (d > 1e-6 ? 0.5 : (d < -1e-6 ? -0.5 : 0.0)));
        }
    std::vector<int32_t> status(n);
    std::vector<lscqp_safety> saf(n);
    int failed = 0;
    double worst = 1e300;
    for (int r = 0; r < replans; r++) {
        if (lscqp_plan_upload(plan, LSCQP_PLAN_BUF_WAYPOINT, r == 0 ? starts.data() : way.data(), 0, 24u * n) != LSCQP_OK) return fail("upload");
        if (lscqp_plan_step_graph(plan, nullptr) != LSCQP_OK) return fail("lscqp_plan_step_graph");
        if (lscqp_plan_download(plan, LSCQP_PLAN_BUF_STATUS, status.data(), 0, 4u * n) != LSCQP_OK) return fail("download");
        if (lscqp_plan_download(plan, LSCQP_PLAN_BUF_SAFETY, saf.data(), 0, sizeof(lscqp_safety) * n) != LSCQP_OK) return fail("download");
        for (int a = 0; a < n; a++) {
            failed += status[a] != LSCQP_STATUS_OPTIMAL;
            worst = saf[a].safety_ratio < worst ? saf[a].safety_ratio : worst;
        }
    }
    const int nv = lscqp_num_variables(h);
    std::vector<double> x((size_t)n * nv), st((size_t)n * 9);
    lscqp_plan_download(plan, LSCQP_PLAN_BUF_PLAN, x.data(), 0, 8u * x.size());
    lscqp_plan_download(plan, LSCQP_PLAN_BUF_STATE, st.data(), 0, 8u * st.size());
    double sum = 0;
    for (double v : x) sum += v;
    printf("{\"scenario\": \"plan\", \"agents\": %d, \"replans\": %d, \"failed\": %d, \"worst_safety_ratio\": %.9g, \"graph_nodes\": %lld, "
           "\"plan_sum\": %.17g, \"state0\": [%.9g, %.9g, %.9g]}\n",
           n, replans, failed, worst, (long long)lscqp_plan_graph_nodes(plan), sum, st[0], st[1], st[2]);
    lscqp_plan_destroy(plan);
    lscqp_map_destroy(map);
    lscqp_destroy(h);
    return 0;
}

int main(int argc, char** argv) {
    std::string s = argc > 1 ? argv[1] : "host";
    if (s == "plan" && argc > 3) return scenario_plan(argv[2], argv[3], argc > 4 ? atoi(argv[4]) : 20);
    if (s == "host") return scenario_host();
    if (s == "kat") return scenario_kat(true);
    if (s == "pair") return scenario_pair();
    if (s == "infeasible") return scenario_infeasible();
    if (s == "goal") return scenario_goal();
    if (s == "sharded") return scenario_sharded();
    if (s == "subsegment") return scenario_subsegment();
    if (s == "dynamic_obstacle") return scenario_dynamic_obstacle();
    if (s == "csv" && argc > 2) return scenario_csv(argv[2]);
    if (s == "summary" && argc > 3) return scenario_summary(argv[2], argv[3]);
    if (s == "sfc" && argc > 2) return scenario_sfc(argv[2]);
    fprintf(stderr, "usage: shim_test host|kat|pair|infeasible|goal\n");
    return 2;
}

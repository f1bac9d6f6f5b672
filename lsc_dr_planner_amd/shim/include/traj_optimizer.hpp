// Drop-in replacement for the reference's include/traj_optimizer.hpp.
//
// PUBLIC SURFACE IS THE REFERENCE'S (include/traj_optimizer.hpp:18-31):
//     struct TrajOptResult { traj_t desired_traj; double total_qp_cost; };
//     class  TrajOptimizer { TrajOptimizer(const Param&, const Mission&, const Eigen::MatrixXd& B);
//                            TrajOptResult solve(const Agent&, const CollisionConstraints&, const traj_t&, bool);
//                            void updateParam(const Param&); };
// What changed: <ilcplex/ilocplex.h> and the private CPLEX members are gone; the private section holds a handle of
// the HIP solver behind the C ABI (include/lscqp.h).  Additive: solveBatch() for callers that can hand over all
// agents of a replan step at once, and lastRawSolution() exposing the fp64 control points before the float32
// truncation the reference applies (src/traj_optimizer.cpp:71-83).
#pragma once
#include <collision_constraints.hpp>
#include <eigen_standin.hpp>
#include <mission.hpp>
#include <param.hpp>
#include <polynomial.hpp>
#include <sp_const.hpp>

#include <lscqp.h>

namespace DynamicPlanning {
struct TrajOptResult {
    traj_t desired_traj;
    double total_qp_cost = 0;
};

class TrajOptimizer {
public:
    TrajOptimizer(const Param& param, const Mission& mission, const Eigen::MatrixXd& B);
    ~TrajOptimizer();
    // copyable like the reference's class (implicit copy there): a copy owns its own solver handle of the same class
    TrajOptimizer(const TrajOptimizer& other);
    TrajOptimizer& operator=(const TrajOptimizer& other);

    TrajOptResult solve(const Agent& agent, const CollisionConstraints& constraints, const traj_t& initial_traj,
                        bool use_primal_algorithm);

    void updateParam(const Param& param);

    // ---- additive API ----
    struct BatchItem {
        const Agent* agent;
        const CollisionConstraints* constraints;
        const traj_t* initial_traj = nullptr;  // optional: primal start of the interior-point iteration
    };
    // Solves every item in one launch.  results[i] is filled for every item; ok[i] == false where the reference
    // would have thrown PlanningReport::QPFAILED (the caller substitutes initial_traj, src/traj_planner.cpp:767-797).
    void solveBatch(const std::vector<BatchItem>& items, std::vector<TrajOptResult>& results, std::vector<bool>& ok);
    // Multi-GPU (one host process, G devices; include/lscqp.h "multi-GPU"): with a communicator set, solveBatch cuts the
    // items into contiguous blocks over as many devices as the batch justifies (lscqp_comm_devices_for) and solves them
    // concurrently; results come back in item order.  The communicator is shared by all TrajOptimizer objects of the process
    // (the reference has one per agent) and owned by the caller.  nullptr (default): single device.
    static void setCommunicator(lscqp_comm comm) { communicator() = comm; }
    int lastDevicesUsed() const { return last_devices_used; }
    const std::vector<double>& lastRawSolution() const { return raw_x; }
    int lastIterations() const { return last_iterations; }
    // the message of the last failed solve: which row of the model the solver could not satisfy (family, oi, m, i, violation) --
    // what the reference's conflict refiner prints (src/traj_optimizer.cpp:105-135); empty after a successful solve
    const std::string& lastConflict() const { return last_conflict; }

private:
    Param param;
    Mission mission;
    Eigen::MatrixXd B;
    int M, n, phi, dim;
    double dt;
    lscqp_handle handle = nullptr;
    std::vector<double> raw_x;
    int last_iterations = 0;
    std::string last_conflict;
    int last_devices_used = 1;
    static lscqp_comm& communicator() {
        static lscqp_comm c = nullptr;
        return c;
    }

    void configure();
    [[nodiscard]] int getTerminalSegments_old(const Agent& agent) const;
    void pack(const Agent& agent, const CollisionConstraints& constraints, lscqp_header& hdr, std::vector<lscqp_row>& rows,
              std::vector<lscqp_box>& boxes) const;
    TrajOptResult unpack(const double* x, double obj) const;
};
}  // namespace DynamicPlanning

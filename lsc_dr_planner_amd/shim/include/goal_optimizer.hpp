// Drop-in replacement for the reference's include/goal_optimizer.hpp.
//
// PUBLIC SURFACE IS THE REFERENCE'S (include/goal_optimizer.hpp:16-24):
//     class GoalOptimizer { GoalOptimizer(const Param&, const Mission&);
//                           point3d solve(const Agent&, const CollisionConstraints&, const point3d& current_goal_point,
//                                         const point3d& next_waypoint); };
// What changed: <ilcplex/ilocplex.h> and the private populatebyrow(IloModel, ...) are gone; the one-variable LP is solved
// in closed form on the device behind lscqp_optimize_goal (include/lscqp.h).  With this and traj_optimizer.hpp the
// planner no longer links CPLEX at all.
#pragma once
#include <collision_constraints.hpp>
#include <mission.hpp>
#include <param.hpp>
#include <sp_const.hpp>

#include <lscqp.h>

namespace DynamicPlanning {
class GoalOptimizer {
public:
    GoalOptimizer(const Param& param, const Mission& mission);
    ~GoalOptimizer();
    GoalOptimizer(const GoalOptimizer&) = delete;
    GoalOptimizer& operator=(const GoalOptimizer&) = delete;

    point3d solve(const Agent& agent, const CollisionConstraints& constraints, const point3d& current_goal_point,
                  const point3d& next_waypoint);

private:
    Param param;
    Mission mission;
    lscqp_handle handle = nullptr;
};
}  // namespace DynamicPlanning

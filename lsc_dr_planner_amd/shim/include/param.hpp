// Stand-in for the reference's include/param.hpp: the fields the trajectory QP reads (SURVEY.md §5 "Config"),
// with the defaults of src/param.cpp / launch/simulation.launch.  No ROS.
#pragma once
#include <sp_const.hpp>
#include <string>

namespace DynamicPlanning {
class Param {
public:
    bool log_solver = false;
    std::string package_path = ".";
    // World
    int world_dimension = 3;
    bool world_use_octomap = true;
    double world_z_2d = 1.0;
    // Planner mode
    PlannerMode planner_mode = PlannerMode::LSC;
    SlackMode slack_mode = SlackMode::NONE;
    GoalMode goal_mode = GoalMode::GRIDBASEDPLANNER;
    // Trajectory representation
    double dt = 0.2;
    int M = 5;
    int n = 5;
    int phi = 3;
    int phi_n = 1;
    // Trajectory optimization
    double control_input_weight = 0.01;
    double terminal_weight = 1.0;
    double slack_collision_weight = 1.0;
    // Goal
    double goal_threshold = 0.1;
    // Communication
    double communication_range = 3.0;
    // Multisim
    double multisim_time_step = 0.2;
};
}  // namespace DynamicPlanning

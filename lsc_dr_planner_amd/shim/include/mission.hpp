// Stand-in for the reference's include/mission.hpp: the QP reads only world_min / world_max
// (src/traj_optimizer.cpp:252-253).
#pragma once
#include <sp_const.hpp>

namespace DynamicPlanning {
class Mission {
public:
    size_t qn = 0;
    std::vector<Agent> agents;
    point3d world_min, world_max;
};
}  // namespace DynamicPlanning

#pragma once
#include <octomap/OcTree.h>
class DynamicEDTOctomap {
public:
    DynamicEDTOctomap(float, octomap::OcTree*, octomap::point3d, octomap::point3d, bool) {}
    void update(bool = true) {}
    float getDistance(const octomap::point3d&) const { return 0; }
    void getDistanceAndClosestObstacle(const octomap::point3d&, float&, octomap::point3d&) const {}
};

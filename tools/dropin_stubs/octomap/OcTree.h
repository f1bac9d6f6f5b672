#pragma once
#include <octomap/octomap_types.h>
#include <string>
namespace octomap {
struct OcTreeNode {};
class OcTree {
public:
    explicit OcTree(double) {}
    explicit OcTree(const std::string&) {}
    double getResolution() const { return 0.1; }
    void getMetricMin(double&, double&, double&) const {}
    void getMetricMax(double&, double&, double&) const {}
    OcTreeNode* search(const point3d&) const { return nullptr; }
    bool isNodeOccupied(const OcTreeNode*) const { return false; }
    void updateNode(const point3d&, bool) {}
    bool readBinary(const std::string&) { return true; }
};
}

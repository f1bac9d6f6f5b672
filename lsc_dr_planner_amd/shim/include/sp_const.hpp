// ROS-free, octomap-free stand-in for the reference's include/sp_const.hpp (only what the trajectory-QP path uses).
// Same names, same namespace, same semantics; see the file:line notes.  In the reference tree the real header is used.
#pragma once

#define SP_EPSILON 1e-9        // include/sp_const.hpp:3
#define SP_EPSILON_FLOAT 1e-5  // include/sp_const.hpp:4
#define SP_INFINITY 1e+9       // include/sp_const.hpp:5

#include <cmath>
#include <stdexcept>
#include <vector>

namespace octomath {
// The subset of octomath::Vector3 (octomap) the QP path touches: three FLOAT32 components.
class Vector3 {
public:
    Vector3() : data{0.f, 0.f, 0.f} {}
    Vector3(float x, float y, float z) : data{x, y, z} {}
    float& x() { return data[0]; }
    float& y() { return data[1]; }
    float& z() { return data[2]; }
    const float& x() const { return data[0]; }
    const float& y() const { return data[1]; }
    const float& z() const { return data[2]; }
    float& operator()(unsigned i) { return data[i]; }
    const float& operator()(unsigned i) const { return data[i]; }
    Vector3 operator+(const Vector3& o) const { return Vector3(data[0] + o.data[0], data[1] + o.data[1], data[2] + o.data[2]); }
    Vector3 operator-(const Vector3& o) const { return Vector3(data[0] - o.data[0], data[1] - o.data[1], data[2] - o.data[2]); }
    Vector3 operator-() const { return Vector3(-data[0], -data[1], -data[2]); }
    Vector3 operator*(float s) const { return Vector3(data[0] * s, data[1] * s, data[2] * s); }
    double dot(const Vector3& o) const { return data[0] * o.data[0] + data[1] * o.data[1] + data[2] * o.data[2]; }
    double norm_sq() const { return data[0] * data[0] + data[1] * data[1] + data[2] * data[2]; }  // float arithmetic
    double norm() const { return std::sqrt(norm_sq()); }
    double distance(const Vector3& o) const { return (*this - o).norm(); }
    Vector3 normalized() const {
        double n = norm();
        return n > 0 ? (*this) * (float)(1.0 / n) : *this;
    }
    bool operator==(const Vector3& o) const { return data[0] == o.data[0] && data[1] == o.data[1] && data[2] == o.data[2]; }

protected:
    float data[3];
};
}  // namespace octomath
namespace octomap {
typedef octomath::Vector3 point3d;
}

namespace DynamicPlanning {
typedef octomap::point3d point3d;   // include/sp_const.hpp:15
typedef octomap::point3d vector3d;  // :16
typedef std::vector<point3d> points_t;

enum class PlannerMode { DLSC, LSC, BVC, ORCA, RECIPROCALRSFC, CIRCLETEST };                     // :19-26
enum class PredictionMode { POSITION, VELOCITY, ORCA, PREVIOUSSOLUTION };                        // :28-33
enum class InitialTrajMode { POSITION, VELOCITY, ORCA, PREVIOUSSOLUTION, SKIP };                 // :35-41
enum class SlackMode { NONE, CONTINUITY, COLLISIONCONSTRAINT };                                  // :43-47
enum class GoalMode { STATIC, ORCA, RIGHTHAND, PRIORBASED, DYNAMICPRIORITY, ENTROPY, GRIDBASEDPLANNER };
enum class MAPFMode { PIBT, ECBS };

enum PlanningReport {  // :72-79 — TrajOptimizer::solve throws QPFAILED (src/traj_optimizer.cpp:143,152)
    Initialized,
    INITTRAJGENERATIONFAILED,
    CONSTRAINTGENERATIONFAILED,
    QPFAILED,
    WAITFORROSMSG,
    SUCCESS,
};

struct State {  // :140-144
    point3d position;
    point3d velocity;
    point3d acceleration;
};

struct Agent {  // :146-160
    int id;
    int cid;
    State current_state;
    point3d start_point;
    point3d desired_goal_point;
    point3d current_goal_point;
    point3d next_waypoint;
    std::vector<double> max_vel;
    std::vector<double> max_acc;
    double radius;
    double downwash;
    double nominal_velocity;
    bool collision_alert;
};
}  // namespace DynamicPlanning

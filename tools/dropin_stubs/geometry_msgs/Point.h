#pragma once
#include <string>
namespace std_msgs { struct Header { unsigned seq = 0; double stamp = 0; std::string frame_id; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct Twist { Vector3 linear, angular; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}

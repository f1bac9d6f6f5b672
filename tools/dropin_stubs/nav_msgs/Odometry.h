#pragma once
#include <geometry_msgs/Point.h>
namespace nav_msgs { struct Odometry { std_msgs::Header header; struct { geometry_msgs::Pose pose; } pose; struct { geometry_msgs::Twist twist; } twist; }; }

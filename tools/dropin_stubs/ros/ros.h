#pragma once
// names only (see README.md)
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <set>
#include <array>
#include <memory>
#include <algorithm>
#include <cmath>
namespace ros {
struct Time { double sec = 0; static Time now() { return Time(); } double toSec() const { return sec; } };
inline double operator-(const Time& a, const Time& b) { return a.sec - b.sec; }
struct Duration { double s; explicit Duration(double v = 0) : s(v) {} double toSec() const { return s; } };
struct NodeHandle {
    template <class T> bool param(const std::string&, T&, const T&) const { return true; }
    template <class T> bool getParam(const std::string&, T&) const { return true; }
};
}  // namespace ros
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
#define ROS_INFO_ONCE(...) ((void)0)
#define ROS_WARN_ONCE(...) ((void)0)

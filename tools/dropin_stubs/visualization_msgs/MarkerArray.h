#pragma once
#include <vector>
#include <string>
#include <geometry_msgs/Point.h>
#include <std_msgs/ColorRGBA.h>
namespace visualization_msgs {
struct Marker {
    enum { ARROW, CUBE, SPHERE, CYLINDER, LINE_STRIP, LINE_LIST, CUBE_LIST, SPHERE_LIST, POINTS, TEXT_VIEW_FACING, MESH_RESOURCE, TRIANGLE_LIST, ADD = 0, MODIFY = 0, DELETE = 2, DELETEALL = 3 };
    std_msgs::Header header; std::string ns; int id = 0, type = 0, action = 0; geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color;
    std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors; std::string text;
};
struct MarkerArray { std::vector<Marker> markers; };
}

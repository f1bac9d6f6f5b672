#pragma once
#include <vector>
#include <std_msgs/MultiArrayDimension.h>
namespace std_msgs {
struct MultiArrayLayout { std::vector<MultiArrayDimension> dim; unsigned data_offset = 0; };
struct Float64MultiArray { MultiArrayLayout layout; std::vector<double> data; };
}

#pragma once
#include <string>
namespace std_msgs { struct MultiArrayDimension { std::string label; unsigned size = 0, stride = 0; }; }

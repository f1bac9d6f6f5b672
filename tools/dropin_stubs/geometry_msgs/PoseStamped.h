#pragma once
#include <geometry_msgs/Point.h>

#pragma once
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }

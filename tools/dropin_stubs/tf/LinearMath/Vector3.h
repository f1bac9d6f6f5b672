#pragma once
namespace tf { struct Vector3 { double v[3] = {0, 0, 0}; double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; } }; }

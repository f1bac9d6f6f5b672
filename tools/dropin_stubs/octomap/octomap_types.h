#pragma once
// octomath::Vector3 / octomap::point3d: three float32 components, the operations the reference headers use
#include <cmath>
#include <vector>
#include <iostream>
namespace octomath {
class Vector3 {
public:
    Vector3() : data{0.f, 0.f, 0.f} {}
    Vector3(float x, float y, float z) : data{x, y, z} {}
    float& x() { return data[0]; } float& y() { return data[1]; } float& z() { return data[2]; }
    const float& x() const { return data[0]; } const float& y() const { return data[1]; } const float& z() const { return data[2]; }
    float& operator()(unsigned i) { return data[i]; }
    const float& operator()(unsigned i) const { return data[i]; }
    Vector3 operator+(const Vector3& o) const { return Vector3(data[0] + o.data[0], data[1] + o.data[1], data[2] + o.data[2]); }
    Vector3 operator-(const Vector3& o) const { return Vector3(data[0] - o.data[0], data[1] - o.data[1], data[2] - o.data[2]); }
    Vector3 operator-() const { return Vector3(-data[0], -data[1], -data[2]); }
    Vector3 operator*(float s) const { return Vector3(data[0] * s, data[1] * s, data[2] * s); }
    Vector3& operator+=(const Vector3& o) { for (int i = 0; i < 3; i++) data[i] += o.data[i]; return *this; }
    Vector3& operator-=(const Vector3& o) { for (int i = 0; i < 3; i++) data[i] -= o.data[i]; return *this; }
    Vector3& operator*=(float s) { for (int i = 0; i < 3; i++) data[i] *= s; return *this; }
    Vector3& operator/=(float s) { for (int i = 0; i < 3; i++) data[i] /= s; return *this; }
    double dot(const Vector3& o) const { return data[0] * o.data[0] + data[1] * o.data[1] + data[2] * o.data[2]; }
    Vector3 cross(const Vector3& o) const { return Vector3(data[1] * o.data[2] - data[2] * o.data[1], data[2] * o.data[0] - data[0] * o.data[2], data[0] * o.data[1] - data[1] * o.data[0]); }
    double norm_sq() const { return dot(*this); }
    double norm() const { return std::sqrt(norm_sq()); }
    double distance(const Vector3& o) const { return (*this - o).norm(); }
    Vector3 normalized() const { return *this; }
    Vector3& normalize() { return *this; }
    bool operator==(const Vector3& o) const { return data[0] == o.data[0] && data[1] == o.data[1] && data[2] == o.data[2]; }
protected:
    float data[3];
};
inline std::ostream& operator<<(std::ostream& s, const Vector3& v) { return s << v.x() << ' ' << v.y() << ' ' << v.z(); }
}  // namespace octomath
namespace octomap { typedef octomath::Vector3 point3d; typedef std::vector<point3d> point3d_collection; }

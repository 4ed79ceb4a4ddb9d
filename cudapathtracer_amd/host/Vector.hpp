// Minimal float3 for the host side (scene description and BVH build only; no
// rendering arithmetic happens on the host).  Operation order follows the
// vector library the reference uses (external/glm/detail/func_geometric.inl:54-96)
// because the BVH topology and the precomputed triangle data depend on it.
#pragma once
#include <cmath>

struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    Vec3() {}
    explicit Vec3(float s) : x(s), y(s), z(s) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline float dot(Vec3 a, Vec3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline Vec3 cross(Vec3 x, Vec3 y) { return Vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline float length(Vec3 v) { return std::sqrt(dot(v, v)); }
inline Vec3 normalize(Vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
// scalar min/max of the reference's vector library (external/glm/detail/func_common.inl:15-28)
inline float vmin(float x, float y) { return x < y ? x : y; }
inline float vmax(float x, float y) { return x > y ? x : y; }
inline Vec3 vmin(Vec3 a, Vec3 b) { return Vec3(vmin(a.x, b.x), vmin(a.y, b.y), vmin(a.z, b.z)); }
inline Vec3 vmax(Vec3 a, Vec3 b) { return Vec3(vmax(a.x, b.x), vmax(a.y, b.y), vmax(a.z, b.z)); }

// Axis-aligned box used by the host BVH build (the reference's Bounds3,
// common/Bounds3.hpp:12-48, :120-134).  The ray/box test itself runs on the device.
#pragma once
#include "Vector.hpp"

class Bounds3 {
public:
    Vec3 pMin, pMax;
    Bounds3() : pMin(Vec3(1e30f)), pMax(Vec3(-1e30f)) {}
    explicit Bounds3(Vec3 p) : pMin(p), pMax(p) {}
    Bounds3(Vec3 p1, Vec3 p2)
        : pMin(std::fmin(p1.x, p2.x), std::fmin(p1.y, p2.y), std::fmin(p1.z, p2.z)),
          pMax(std::fmax(p1.x, p2.x), std::fmax(p1.y, p2.y), std::fmax(p1.z, p2.z)) {}

    Vec3 Diagonal() const { return pMax - pMin; }
    int maxExtent() const {
        Vec3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    Vec3 Centroid() const { return pMin * 0.5f + pMax * 0.5f; }
};

inline Bounds3 Union(const Bounds3& b1, const Bounds3& b2) {
    Bounds3 ret;
    ret.pMin = vmin(b1.pMin, b2.pMin);
    ret.pMax = vmax(b1.pMax, b2.pMax);
    return ret;
}
inline Bounds3 Union(const Bounds3& b, const Vec3& p) {
    Bounds3 ret;
    ret.pMin = vmin(b.pMin, p);
    ret.pMax = vmax(b.pMax, p);
    return ret;
}

// Ray and Intersection as host callers of the reference know them (common/Ray.hpp:10-34,
// common/Intersection.hpp:14-35).  On the host they only carry queries to and answers from the
// device (Scene::intersect); no traversal arithmetic happens here.
#pragma once
#include "Material.hpp"
#include "Vector.hpp"

constexpr double kDoubleInfinity = 1e30f;  // common/MathUtils.hpp:72: the float literal's value, as a double

struct Ray {
    Vec3 origin;
    Vec3 direction, direction_inv;
    bool shadowTest = false;  // any-hit query: is there a hit nearer than shadowDistance?
    double shadowDistance = kDoubleInfinity;
    Ray() {}
    Ray(const Vec3& ori, const Vec3& dir, bool _shadowTest = false, double _shadowDistance = kDoubleInfinity)
        : origin(ori), direction(dir), shadowTest(_shadowTest), shadowDistance(_shadowDistance) {
        direction_inv = Vec3((float)(1. / dir.x), (float)(1. / dir.y), (float)(1. / dir.z));
    }
    Vec3 operator()(double t) const { return origin + direction * (float)t; }
};

struct Intersection {
    bool happened = false;
    Vec3 coords;
    Vec3 tcoords;
    Vec3 normal;
    Vec3 emit;
    double distance = kDoubleInfinity;
    float triangleArea = 0.f;
    Material m;
    int primitive = -1;  // not in the reference: global primitive id (Scene::Add order, then index in the mesh)
};

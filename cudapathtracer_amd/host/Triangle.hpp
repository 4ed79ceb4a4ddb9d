// Host-side geometry: Object / Triangle / MeshTriangle with the public members
// the reference's scene-building code uses (common/Object.hpp:12-26,
// common/Triangle.hpp:42-59, :87-154).  Intersection and sampling run on the
// device against the flattened records that BVHAccel/Scene emit.
#pragma once
#include <string>
#include <vector>

#include "Bounds3.hpp"
#include "Material.hpp"

class Object {
public:
    float area = 0.f;
    Object() {}
    virtual ~Object() {}
    virtual Bounds3 getBounds() const = 0;
};

class Triangle : public Object {
public:
    Vec3 v0, v1, v2;  // vertices, counter-clockwise
    Vec3 e1, e2;      // v1-v0, v2-v0
    Vec3 normal;
    Material material;

    Triangle() {}
    // precomputation as common/Triangle.hpp:52-59
    Triangle(Vec3 _v0, Vec3 _v1, Vec3 _v2, const Material& _m) : v0(_v0), v1(_v1), v2(_v2), material(_m) {
        e1 = v1 - v0;
        e2 = v2 - v0;
        normal = normalize(cross(e1, e2));
        area = length(cross(e1, e2)) * 0.5f;
    }
    Bounds3 getBounds() const override { return Union(Bounds3(v0, v1), v2); }
};

// Sphere: the primitive of common/Sphere.hpp:13-101, which the reference keeps commented out
// (north_star names it).  Members as there; an object of the top-level BVH like a mesh.  Spheres
// do not emit (Scene::sampleLight only knows meshes).
class Sphere : public Object {
public:
    Vec3 center;
    float radius = 0.f, radius2 = 0.f;
    Material material;
    Sphere(const Vec3& c, float r, const Material& mt) : center(c), radius(r), radius2(r * r), material(mt) {
        area = 4 * 3.1415926535897932384626422832795028841971f * r * r;
    }
    Bounds3 getBounds() const override {
        return Bounds3(Vec3(center.x - radius, center.y - radius, center.z - radius), Vec3(center.x + radius, center.y + radius, center.z + radius));
    }
};

// Translation, rotation (degrees about x, then y, then z) and scale of a mesh, composed as
// Math::buildTransformationMatrix does (common/MathUtils.hpp:109-116): M = T * Rx * Ry * Rz * S.
// The reference declares that helper and never calls it (a MeshTriangle is the OBJ's vertices
// verbatim, common/Triangle.cpp:28-53); here a mesh can be given one.  Applied to every vertex in
// float, column by column as the reference's vector library multiplies mat4 * vec4:
// (m0 * x + m1 * y) + (m2 * z + m3).  Parity unpinned: nothing in the reference exercises it.
struct Transform {
    Vec3 translation = Vec3(0.f), rotation = Vec3(0.f), scale = Vec3(1.f);
    bool identity() const;
    void matrix(float m[4][4]) const;  // m[column][row]
    Vec3 apply(const float m[4][4], Vec3 v) const;
};

class MeshTriangle : public Object {
public:
    // Wavefront OBJ with `v` and triangular `f` records, one mesh per file
    // (what common/Triangle.cpp:5-61 accepts through the vendored loader).
    // Throws std::runtime_error if the file cannot be read or a face is not a triangle.
    MeshTriangle(const std::string& filename, const Material& mt);
    MeshTriangle(const std::string& filename, const Material& mt, const Transform& transform);
    // triangle soup: 3 vertices per face
    MeshTriangle(const std::vector<Vec3>& face_vertices, const Material& mt);

    Bounds3 getBounds() const override { return bounding_box; }

    Bounds3 bounding_box;
    int num_vertices = 0;
    int num_triangles = 0;
    std::vector<Vec3> vertices;       // 3 per face, un-indexed
    std::vector<Triangle> triangles;  // triangle soup
    Material material;

private:
    void build(const std::vector<Vec3>& face_vertices, const Material& mt);
};

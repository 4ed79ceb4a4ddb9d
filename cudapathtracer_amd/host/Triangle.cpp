#include "Triangle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace {

// `v x y z` and `f a b c` (each index as a, a/b, a//c or a/b/c; negative =
// relative).  Emits three un-indexed vertices per face in file order, which is
// the vertex stream MeshTriangle is defined over (common/Triangle.cpp:28-53,
// common/OBJ_Loader.hpp:590-595).
std::vector<Vec3> read_obj(const std::string& filename) {
    FILE* fp = std::fopen(filename.c_str(), "r");
    if (!fp) throw std::runtime_error("MeshTriangle: cannot open '" + filename + "'");
    std::vector<Vec3> verts, faces;
    char line[2048];
    int lineno = 0;
    while (std::fgets(line, sizeof line, fp)) {
        ++lineno;
        char* s = line;
        while (*s == ' ' || *s == '\t') ++s;
        const bool sep = s[1] == ' ' || s[1] == '\t';
        if (s[0] == 'v' && sep) {
            char* e = s + 1;
            float c[3];
            for (float& k : c) k = std::strtof(e, &e);
            verts.emplace_back(c[0], c[1], c[2]);
        } else if (s[0] == 'f' && sep) {
            char* e = s + 1;
            long idx[4];
            int cnt = 0;
            while (cnt < 4) {
                while (*e == ' ' || *e == '\t') ++e;
                if (*e == '\0' || *e == '\n' || *e == '\r') break;
                idx[cnt++] = std::strtol(e, &e, 10);
                while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') ++e;
            }
            if (cnt != 3) {
                std::fclose(fp);
                throw std::runtime_error("MeshTriangle: '" + filename + "' line " + std::to_string(lineno) + ": only triangular faces are supported");
            }
            for (int k = 0; k < 3; ++k) {
                const long id = idx[k] < 0 ? (long)verts.size() + idx[k] : idx[k] - 1;
                if (id < 0 || id >= (long)verts.size()) {
                    std::fclose(fp);
                    throw std::runtime_error("MeshTriangle: '" + filename + "' line " + std::to_string(lineno) + ": vertex index out of range");
                }
                faces.push_back(verts[id]);
            }
        }
    }
    std::fclose(fp);
    if (faces.empty()) throw std::runtime_error("MeshTriangle: '" + filename + "' has no faces");
    return faces;
}

}  // namespace

MeshTriangle::MeshTriangle(const std::string& filename, const Material& mt) { build(read_obj(filename), mt); }

MeshTriangle::MeshTriangle(const std::string& filename, const Material& mt, const Transform& transform) {
    std::vector<Vec3> fv = read_obj(filename);
    if (!transform.identity()) {
        float m[4][4];
        transform.matrix(m);
        for (Vec3& v : fv) v = transform.apply(m, v);
    }
    build(fv, mt);
}

bool Transform::identity() const {
    return translation.x == 0.f && translation.y == 0.f && translation.z == 0.f && rotation.x == 0.f && rotation.y == 0.f &&
           rotation.z == 0.f && scale.x == 1.f && scale.y == 1.f && scale.z == 1.f;
}

namespace {
void mat_identity(float m[4][4]) {
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) m[c][r] = c == r ? 1.f : 0.f;
}
// out = a * b (column-major: out[c] = a[0]*b[c][0] + a[1]*b[c][1] + a[2]*b[c][2] + a[3]*b[c][3], summed left to right)
void mat_mul(const float a[4][4], const float b[4][4], float out[4][4]) {
    float t[4][4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) t[c][r] = ((a[0][r] * b[c][0] + a[1][r] * b[c][1]) + a[2][r] * b[c][2]) + a[3][r] * b[c][3];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) out[c][r] = t[c][r];
}
// rotation by `deg` degrees about coordinate axis `axis` (0 x, 1 y, 2 z), angle = deg * Pi / 180.f
void mat_rotation(float deg, int axis, float m[4][4]) {
    const float pi = 3.1415926535897932384626422832795028841971f;
    const float a = deg * pi / 180.f;
    const float c = std::cos(a), s = std::sin(a);
    mat_identity(m);
    const int u = (axis + 1) % 3, v = (axis + 2) % 3;
    m[u][u] = c;
    m[u][v] = s;
    m[v][u] = -s;
    m[v][v] = c;
}
}  // namespace

void Transform::matrix(float m[4][4]) const {
    float t[4][4], r[4][4], rx[4][4], ry[4][4], rz[4][4], sc[4][4], tr[4][4];
    mat_identity(t);
    t[3][0] = translation.x;
    t[3][1] = translation.y;
    t[3][2] = translation.z;
    mat_rotation(rotation.x, 0, rx);
    mat_rotation(rotation.y, 1, ry);
    mat_rotation(rotation.z, 2, rz);
    mat_mul(rx, ry, r);
    mat_mul(r, rz, r);
    mat_identity(sc);
    sc[0][0] = scale.x;
    sc[1][1] = scale.y;
    sc[2][2] = scale.z;
    mat_mul(t, r, tr);
    mat_mul(tr, sc, m);
}

Vec3 Transform::apply(const float m[4][4], Vec3 v) const {
    float out[3];
    for (int r = 0; r < 3; ++r) out[r] = (m[0][r] * v.x + m[1][r] * v.y) + (m[2][r] * v.z + m[3][r]);
    return Vec3(out[0], out[1], out[2]);
}

MeshTriangle::MeshTriangle(const std::vector<Vec3>& face_vertices, const Material& mt) {
    if (face_vertices.empty() || face_vertices.size() % 3 != 0) throw std::runtime_error("MeshTriangle: vertex count must be a positive multiple of 3");
    build(face_vertices, mt);
}

// bounding box over all vertices, triangles in file order, area summed in that
// order (common/Triangle.cpp:21-60)
void MeshTriangle::build(const std::vector<Vec3>& fv, const Material& mt) {
    material = mt;
    area = 0;
    vertices = fv;
    num_vertices = (int)fv.size();
    num_triangles = num_vertices / 3;
    triangles.reserve(num_triangles);
    const float inf = std::numeric_limits<float>::infinity();
    Vec3 lo(inf), hi(-inf);
    for (int i = 0; i + 2 < num_vertices; i += 3) {
        for (int j = 0; j < 3; ++j) {
            const Vec3& v = fv[i + j];
            lo = Vec3(std::min(lo.x, v.x), std::min(lo.y, v.y), std::min(lo.z, v.z));
            hi = Vec3(std::max(hi.x, v.x), std::max(hi.y, v.y), std::max(hi.z, v.z));
        }
        triangles.emplace_back(fv[i], fv[i + 1], fv[i + 2], mt);
        area += triangles.back().area;
    }
    bounding_box = Bounds3(lo, hi);
}

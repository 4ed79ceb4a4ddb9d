#include "Triangle.hpp"

#include <algorithm>
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

// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Thin extern "C" harness around the *unmodified* reference sources, which are
// compiled from where they lie under /root/reference (see oracle/Makefile,
// target `ref`).  The result (oracle/_ref/libref.so) only exists in the build
// container; it is used to (1) validate the CPU restatement in oracle.cpp and
// (2) generate the golden fixtures under tests/golden/.
//
// Nothing here re-implements reference behaviour: every number comes out of a
// call into the reference's own functions (Scene::intersect, Renderer::Render,
// Material::sample/pdf/bsdf, Bounds3::IntersectP, Triangle::getIntersection...).
// The only logic of its own is the id-tracking re-walk of the public
// nextIfHit/nextIfMiss links in ref_intersect_batch, needed because the
// reference's Intersection carries no primitive id (common/Intersection.hpp:14-32).
#include "Scene.hpp"
#include "Renderer.hpp"
#include "Utility.hpp"

#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>
#include <unordered_map>
#include <vector>

namespace {

Scene* g_scene = nullptr;
std::vector<int> g_mesh_base;  // global prim id base per mesh, Scene::Add order

void collect_preorder(BVHBuildNode* n, std::vector<BVHBuildNode*>& out) {
    if (!n) return;
    out.push_back(n);
    collect_preorder(n->left, out);
    collect_preorder(n->right, out);
}

int mesh_index_of(MeshTriangle* m) {
    for (int i = 0; i < g_scene->num_meshes; ++i)
        if (g_scene->meshes_data[i] == m) return i;
    return -1;
}

void finish_scene() {
    g_scene->buildBVH();
    g_mesh_base.clear();
    int base = 0;
    for (int i = 0; i < g_scene->num_meshes; ++i) {
        g_mesh_base.push_back(base);
        base += g_scene->meshes_data[i]->num_triangles;
    }
}

struct StdoutSilencer {
    int saved;
    StdoutSilencer() {
        fflush(stdout);
        std::cout.flush();
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~StdoutSilencer() {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};

Material make_material(int type, const float* emission, const float* albedo, float metallic,
                       float roughness, float ior) {
    Material m;
    m.type = type;
    m._emission = glm::vec3(emission[0], emission[1], emission[2]);
    m._albedo = glm::vec3(albedo[0], albedo[1], albedo[2]);
    m._metallic = metallic;
    m._roughness = roughness;
    m._ior = ior;
    return m;
}

}  // namespace

extern "C" {

// Scene::CreateBuiltinScene(CornellBox, maxDepth) with cwd = root_dir (the
// reference opens "models/cornellbox/*.obj" relative to the cwd, Scene.cpp:56-61).
int ref_scene_cornell(const char* root_dir, int max_depth) {
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return 1;
    if (chdir(root_dir) != 0) return 2;
    delete g_scene;
    g_scene = new Scene(Scene::CreateBuiltinScene(Scene::CornellBox, max_depth));
    // the by-value copy above keeps meshes_data pointing at the moved-from
    // vector's storage only if RVO applied; re-seat it to be safe.
    g_scene->meshes_data = g_scene->meshes.data();
    int rc = chdir(cwd);
    g_mesh_base.clear();
    int base = 0;
    for (int i = 0; i < g_scene->num_meshes; ++i) {
        g_mesh_base.push_back(base);
        base += g_scene->meshes_data[i]->num_triangles;
    }
    return rc;
}

// Arbitrary scene through the reference's own Scene::Add + buildBVH.
int ref_scene_begin(int w, int h, int max_depth, const float* cam_pos, double fov) {
    delete g_scene;
    g_scene = new Scene(w, h);
    g_scene->maxDepth = max_depth;
    g_scene->camPos = glm::vec3(cam_pos[0], cam_pos[1], cam_pos[2]);
    g_scene->fov = fov;
    g_scene->name = "custom";
    return 0;
}

int ref_scene_add_obj(const char* path, int type, const float* emission, const float* albedo,
                      float metallic, float roughness, float ior) {
    Material m = make_material(type, emission, albedo, metallic, roughness, ior);
    g_scene->Add(new MeshTriangle(path, m));
    return g_scene->num_meshes - 1;
}

int ref_scene_build() {
    finish_scene();
    return 0;
}

void ref_set_resolution(int w, int h) {
    g_scene->width = w;
    g_scene->height = h;
}

void ref_set_max_depth(int d) { g_scene->maxDepth = d; }

void ref_scene_info(int* num_meshes, int* num_tris, int* width, int* height, float* sum_light_area,
                    float* sum_light_power, int* num_lights, int* node_counter) {
    *num_meshes = g_scene->num_meshes;
    int t = 0;
    for (int i = 0; i < g_scene->num_meshes; ++i) t += g_scene->meshes_data[i]->num_triangles;
    *num_tris = t;
    *width = g_scene->width;
    *height = g_scene->height;
    *sum_light_area = g_scene->sumLightArea;
    *sum_light_power = g_scene->sumLightPower;
    *num_lights = g_scene->num_lights;
    *node_counter = BVHBuildNode::Counter;
}

void ref_mesh_info(int mesh, int* num_tris, float* area, float* bbox6, float* bvh_root_area) {
    MeshTriangle* m = g_scene->meshes_data[mesh];
    *num_tris = m->num_triangles;
    *area = m->area;
    bbox6[0] = m->bounding_box.pMin.x; bbox6[1] = m->bounding_box.pMin.y; bbox6[2] = m->bounding_box.pMin.z;
    bbox6[3] = m->bounding_box.pMax.x; bbox6[4] = m->bounding_box.pMax.y; bbox6[5] = m->bounding_box.pMax.z;
    *bvh_root_area = g_scene->mesh_bvhs[mesh]->root->area;
}

// per triangle: v0,e1,e2,normal (12 floats) + area
void ref_mesh_triangles(int mesh, float* out13) {
    MeshTriangle* m = g_scene->meshes_data[mesh];
    for (int i = 0; i < m->num_triangles; ++i) {
        const Triangle& t = m->triangles[i];
        float* o = out13 + 13 * i;
        o[0] = t.v0.x; o[1] = t.v0.y; o[2] = t.v0.z;
        o[3] = t.e1.x; o[4] = t.e1.y; o[5] = t.e1.z;
        o[6] = t.e2.x; o[7] = t.e2.y; o[8] = t.e2.z;
        o[9] = t.normal.x; o[10] = t.normal.y; o[11] = t.normal.z;
        o[12] = t.area;
    }
}

// Pre-order node dump of one BVH: mesh = -1 -> top level (leaf payload = mesh
// index), else the nested per-mesh BVH (leaf payload = triangle index within
// the mesh).  Returns the node count; pass null pointers to only count.
int ref_bvh_dump(int mesh, float* bounds6, int* payload, float* area, int* miss_rank) {
    BVHAccel* bvh = mesh < 0 ? g_scene->bvh : g_scene->mesh_bvhs[mesh];
    std::vector<BVHBuildNode*> nodes;
    collect_preorder(bvh->root, nodes);
    if (!bounds6) return (int)nodes.size();
    std::unordered_map<const BVHBuildNode*, int> rank_of;
    for (size_t i = 0; i < nodes.size(); ++i) rank_of[nodes[i]] = (int)i;
    for (size_t i = 0; i < nodes.size(); ++i) {
        BVHBuildNode* n = nodes[i];
        bounds6[6 * i + 0] = n->bounds.pMin.x; bounds6[6 * i + 1] = n->bounds.pMin.y; bounds6[6 * i + 2] = n->bounds.pMin.z;
        bounds6[6 * i + 3] = n->bounds.pMax.x; bounds6[6 * i + 4] = n->bounds.pMax.y; bounds6[6 * i + 5] = n->bounds.pMax.z;
        bool leaf = !n->left && !n->right;
        if (!leaf) payload[i] = -1;
        else if (n->mesh) payload[i] = mesh_index_of(n->mesh);
        else payload[i] = (int)(n->triangle - g_scene->meshes_data[mesh]->triangles);
        area[i] = n->area;
        miss_rank[i] = n->nextIfMiss ? rank_of.at(n->nextIfMiss) : (int)nodes.size();
        // nextIfHit must be the pre-order successor (BVH.cpp:53-62)
        BVHBuildNode* succ = i + 1 < nodes.size() ? nodes[i + 1] : nullptr;
        if (n->nextIfHit != succ) return -1;
    }
    return (int)nodes.size();
}

// Id-tracking walk over the reference's own threaded links, calling the
// reference's own Bounds3::IntersectP and Triangle::getIntersection, with the
// closest/any-hit update rules of BVHAccel::getIntersection (BVH.hpp:109-143).
// Also calls Scene::intersect and reports whether its distance/happened agree
// bit-for-bit (agree_out, may be null).  counts_out (may be null) receives
// per-ray {box tests, triangle tests}.
void ref_intersect_batch(long n, const float* o, const float* d, const unsigned char* shadow,
                         const double* tmax, int* prim_out, double* t_out, int* agree_out,
                         int* counts_out) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (long r = 0; r < n; ++r) {
        glm::vec3 ro(o[3 * r], o[3 * r + 1], o[3 * r + 2]);
        glm::vec3 rd(d[3 * r], d[3 * r + 1], d[3 * r + 2]);
        bool sh = shadow && shadow[r];
        Ray ray = sh ? Ray(ro, rd, true, tmax[r]) : Ray(ro, rd);
        const int dirIsNeg[3] = {ray.direction.x < 0, ray.direction.y < 0, ray.direction.z < 0};
        int best = -1;
        double best_t = Intersection().distance;
        int nbox = 0, ntri = 0;
        bool done = false;
        BVHBuildNode* cur = g_scene->bvh->root;
        while (cur && !done) {
            ++nbox;
            if (!cur->bounds.IntersectP(ray, ray.direction_inv, dirIsNeg)) { cur = cur->nextIfMiss; continue; }
            if (!cur->left && !cur->right) {
                int mi = mesh_index_of(cur->mesh);
                MeshTriangle* mesh = cur->mesh;
                // nested mesh BVH, same loop one level down (BVH.hpp:117)
                int mbest = -1;
                double mbest_t = Intersection().distance;
                bool mfound = false;
                BVHBuildNode* mc = cur->meshBvhRoot->root;
                while (mc) {
                    ++nbox;
                    if (!mc->bounds.IntersectP(ray, ray.direction_inv, dirIsNeg)) { mc = mc->nextIfMiss; continue; }
                    if (!mc->left && !mc->right) {
                        ++ntri;
                        Intersection hit = mc->triangle->getIntersection(ray);
                        if (hit.happened) {
                            int id = g_mesh_base[mi] + (int)(mc->triangle - mesh->triangles);
                            if (sh) {
                                if (hit.distance < ray.shadowDistance - Epsilon4) { mbest = id; mbest_t = hit.distance; mfound = true; break; }
                            } else if (hit.distance < mbest_t) { mbest = id; mbest_t = hit.distance; mfound = true; }
                        }
                    }
                    mc = mc->nextIfHit;
                }
                if (mfound) {
                    if (sh) {
                        if (mbest_t < ray.shadowDistance - Epsilon4) { best = mbest; best_t = mbest_t; done = true; }
                    } else if (mbest_t < best_t) { best = mbest; best_t = mbest_t; }
                }
            }
            cur = cur->nextIfHit;
        }
        prim_out[r] = best;
        t_out[r] = best_t;
        if (counts_out) { counts_out[2 * r] = nbox; counts_out[2 * r + 1] = ntri; }
        if (agree_out) {
            Intersection ref = g_scene->intersect(ray);
            bool hap = best >= 0;
            agree_out[r] = (ref.happened == hap) && (ref.distance == best_t);
        }
    }
}

// Renderer::Render of the reference (OpenMP over pixels), framebuffer copied out.
void ref_render(int spp, float* rgb_out) {
    Renderer r;
    r.spp = spp;
    {
        StdoutSilencer quiet;
        r.Render(*g_scene);
    }
    std::memcpy(rgb_out, r.framebuffer.data(), sizeof(float) * 3 * r.framebuffer.size());
}

// Radiance of single (sample k, pixel m) paths, exactly the body of the pixel
// loop in Renderer::Render (Renderer.cpp:32-40) without the /spp.
void ref_cast_paths(long n, const int* sample_k, const int* pixel_m, float* rgb_out) {
    const Scene& scene = *g_scene;
    // same expressions as Renderer.cpp:11,20 (deg2rad takes a const float&)
    const float half_fov_deg = scene.fov * 0.5;
    const float half_fov_rad = half_fov_deg * Pi / 180.0;
    float scale = tan(half_fov_rad);
    float imageAspectRatio = scene.width / (float)scene.height;
    glm::vec3 eye_pos = scene.camPos;
#pragma omp parallel for schedule(dynamic, 256)
    for (long q = 0; q < n; ++q) {
        int k = sample_k[q], m = pixel_m[q];
        uint32_t i = m % scene.width, j = m / scene.width;
        RNG rng = RNG(k, m, 0, nullptr);
        float bias = rng.sample1D();
        float x = (2 * (i + bias) / (float)scene.width - 1) * imageAspectRatio * scale;
        float y = (1 - 2 * (j + bias) / (float)scene.height) * scale;
        glm::vec3 dir = glm::normalize(glm::vec3(-x, y, 1));
        glm::vec3 L = scene.castRay(rng, Ray(eye_pos, dir));
        rgb_out[3 * q] = L.x; rgb_out[3 * q + 1] = L.y; rgb_out[3 * q + 2] = L.z;
    }
}

// First `count` floats of the reference RNG stream RNG(iter, index, 0).
void ref_rng_floats(int iter, int index, int count, float* out) {
    RNG rng(iter, index, 0, nullptr);
    for (int i = 0; i < count; ++i) out[i] = rng.sample1D();
}

void ref_material_eval(int type, const float* emission, const float* albedo, float metallic, float roughness,
                       float ior, long n, const float* wi, const float* wo, const float* nrm, float* bsdf_out,
                       float* pdf_out) {
    Material m = make_material(type, emission, albedo, metallic, roughness, ior);
    for (long q = 0; q < n; ++q) {
        glm::vec3 a(wi[3 * q], wi[3 * q + 1], wi[3 * q + 2]);
        glm::vec3 b(wo[3 * q], wo[3 * q + 1], wo[3 * q + 2]);
        glm::vec3 c(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]);
        glm::vec3 f = m.bsdf(a, b, c);
        bsdf_out[3 * q] = f.x; bsdf_out[3 * q + 1] = f.y; bsdf_out[3 * q + 2] = f.z;
        pdf_out[q] = m.pdf(a, b, c);
    }
}

// Material::sample driven by RNG(iter0 + q, index, 0); u_out gets the first 3
// floats of the same stream so the restatement can be fed identical numbers.
void ref_material_sample(int type, const float* emission, const float* albedo, float metallic, float roughness,
                         float ior, long n, int iter0, int index, const float* wo, const float* nrm,
                         float* wi_out, float* u_out) {
    Material m = make_material(type, emission, albedo, metallic, roughness, ior);
    for (long q = 0; q < n; ++q) {
        RNG rng(iter0 + (int)q, index, 0, nullptr);
        RNG copy = rng;
        for (int k = 0; k < 3; ++k) u_out[3 * q + k] = copy.sample1D();
        glm::vec3 b(wo[3 * q], wo[3 * q + 1], wo[3 * q + 2]);
        glm::vec3 c(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]);
        glm::vec3 w = m.sample(rng, b, c);
        wi_out[3 * q] = w.x; wi_out[3 * q + 1] = w.y; wi_out[3 * q + 2] = w.z;
    }
}

void ref_sample_hemisphere_cosine(long n, const float* nrm, const float* rx, const float* ry, float* out) {
    for (long q = 0; q < n; ++q) {
        glm::vec3 c(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]);
        glm::vec3 w = Math::sampleHemisphereCosine(c, rx[q], ry[q]);
        out[3 * q] = w.x; out[3 * q + 1] = w.y; out[3 * q + 2] = w.z;
    }
}

// Scene::sampleLight driven by RNG(iter0 + q, index, 0): coords(3) normal(3)
// emit(3) triangleArea pdf = 11 floats per sample, plus the first 4 stream floats.
void ref_sample_light(long n, int iter0, int index, float* out11, float* u_out) {
    for (long q = 0; q < n; ++q) {
        RNG rng(iter0 + (int)q, index, 0, nullptr);
        RNG copy = rng;
        for (int k = 0; k < 4; ++k) u_out[4 * q + k] = copy.sample1D();
        Intersection pos;
        float pdf = 0.f;
        g_scene->sampleLight(rng, pos, pdf);
        float* o = out11 + 11 * q;
        o[0] = pos.coords.x; o[1] = pos.coords.y; o[2] = pos.coords.z;
        o[3] = pos.normal.x; o[4] = pos.normal.y; o[5] = pos.normal.z;
        o[6] = pos.emit.x; o[7] = pos.emit.y; o[8] = pos.emit.z;
        o[9] = pos.triangleArea;
        o[10] = pdf;
    }
}

float ref_power_heuristic(float f, float g) { return Math::powerHeuristic(f, g); }

// solveQuadratic (common/MathUtils.hpp:86-100): the live half of the commented-out Sphere
int ref_solve_quadratic(float a, float b, float c, float* x0, float* x1) { return solveQuadratic(a, b, c, *x0, *x1) ? 1 : 0; }

// Material::regularize (common/Material.hpp:164-171): roughness in, roughness out
float ref_material_regularize(float roughness) {
    Material m;
    m._roughness = roughness;
    m.regularize();
    return m._roughness;
}

// SavePPM tone curve of one channel value (Utility.cpp:16)
int ref_tonemap(float v) { return (unsigned char)(255 * glm::pow(clamp(0, 1, v), 0.6f)); }

}  // extern "C"

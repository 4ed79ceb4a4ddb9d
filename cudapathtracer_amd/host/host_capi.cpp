// C view of the C++ host classes (include/hpt_host.h).
#include "../../include/hpt_host.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "HipRenderer.hpp"
#include "Scene.hpp"
#include "Utility.hpp"

struct hpt_host_scene {
    Scene scene;
    explicit hpt_host_scene(const Scene& s) : scene(s) {}
};

namespace {
thread_local std::string g_err;
int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}
Material to_material(const hpt_material* m) {
    Material r;
    r.type = m->type;
    r._emission = Vec3(m->emission[0], m->emission[1], m->emission[2]);
    r._albedo = Vec3(m->albedo[0], m->albedo[1], m->albedo[2]);
    r._metallic = m->metallic;
    r._roughness = m->roughness;
    r._ior = m->ior;
    return r;
}
}  // namespace

#define GUARD(body)                                  \
    try {                                            \
        body;                                        \
    } catch (const std::exception& e) {              \
        return fail(e.what());                       \
    } catch (...) {                                  \
        return fail("unknown C++ exception");        \
    }                                                \
    return 0;

extern "C" {

const char* hpt_host_last_error(void) { return g_err.c_str(); }

int hpt_host_scene_cornell(const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out) {
    if (!out) return fail("out is null");
    *out = nullptr;
    GUARD({
        if (models_dir) setenv("HPT_MODELS_DIR", models_dir, 1);
        Scene sc = Scene::CreateBuiltinScene(Scene::CornellBox, max_depth);
        if (width > 0) sc.width = width;
        if (height > 0) sc.height = height;
        *out = new hpt_host_scene(sc);
    })
}

int hpt_host_scene_builtin(int scene_id, const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out) {
    if (!out) return fail("out is null");
    *out = nullptr;
    if (scene_id < (int)Scene::CornellBox || scene_id > (int)Scene::GlassCornell) return fail("unknown built-in scene");
    GUARD({
        if (models_dir) setenv("HPT_MODELS_DIR", models_dir, 1);
        Scene sc = Scene::CreateBuiltinScene((Scene::BuiltinScene)scene_id, max_depth);
        if (width > 0) sc.width = width;
        if (height > 0) sc.height = height;
        *out = new hpt_host_scene(sc);
    })
}

int hpt_host_scene_named(const char* name, const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out) {
    if (!out || !name) return fail("null argument");
    *out = nullptr;
    GUARD({
        if (models_dir) setenv("HPT_MODELS_DIR", models_dir, 1);
        Scene sc = Scene::CreateNamedScene(name, max_depth);
        if (width > 0) sc.width = width;
        if (height > 0) sc.height = height;
        *out = new hpt_host_scene(sc);
    })
}

int hpt_host_scene_info(const hpt_host_scene* s, int* width, int* height, char* name, int name_cap) {
    if (!s) return fail("scene is null");
    if (width) *width = s->scene.width;
    if (height) *height = s->scene.height;
    if (name && name_cap > 0) {
        std::strncpy(name, s->scene.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = '\0';
    }
    return 0;
}

int hpt_host_scene_add_obj_transformed(hpt_host_scene* s, const char* obj_path, const hpt_material* material, const float translation[3],
                                       const float rotation_deg[3], const float scale[3]) {
    if (!s || !obj_path || !material || !translation || !rotation_deg || !scale) return fail("null argument");
    GUARD({
        Transform t;
        t.translation = Vec3(translation[0], translation[1], translation[2]);
        t.rotation = Vec3(rotation_deg[0], rotation_deg[1], rotation_deg[2]);
        t.scale = Vec3(scale[0], scale[1], scale[2]);
        s->scene.Add(new MeshTriangle(std::string(obj_path), to_material(material), t));
    })
}

int hpt_host_scene_create(int width, int height, const float cam_pos[3], double fov_deg, int max_depth, hpt_host_scene** out) {
    if (!out || !cam_pos) return fail("null argument");
    *out = nullptr;
    GUARD({
        Scene sc(width, height);
        sc.camPos = Vec3(cam_pos[0], cam_pos[1], cam_pos[2]);
        sc.fov = fov_deg;
        sc.maxDepth = max_depth;
        sc.name = "custom";
        *out = new hpt_host_scene(sc);
    })
}

int hpt_host_scene_add_obj(hpt_host_scene* s, const char* obj_path, const hpt_material* material) {
    if (!s || !obj_path || !material) return fail("null argument");
    GUARD(s->scene.Add(new MeshTriangle(std::string(obj_path), to_material(material))))
}

int hpt_host_scene_add_triangles(hpt_host_scene* s, const float* verts9, int num_tris, const hpt_material* material) {
    if (!s || !verts9 || !material || num_tris <= 0) return fail("bad argument");
    GUARD({
        std::vector<Vec3> fv((size_t)num_tris * 3);
        for (size_t i = 0; i < fv.size(); ++i) fv[i] = Vec3(verts9[3 * i], verts9[3 * i + 1], verts9[3 * i + 2]);
        s->scene.Add(new MeshTriangle(fv, to_material(material)));
    })
}

int hpt_host_scene_add_sphere(hpt_host_scene* s, const float center[3], float radius, const hpt_material* material) {
    if (!s || !center || !material) return fail("null argument");
    GUARD(s->scene.Add(new Sphere(Vec3(center[0], center[1], center[2]), radius, to_material(material))))
}

int hpt_host_scene_build(hpt_host_scene* s) {
    if (!s) return fail("scene is null");
    GUARD(s->scene.buildBVH())
}

int hpt_host_scene_set_resolution(hpt_host_scene* s, int width, int height) {
    if (!s || width <= 0 || height <= 0) return fail("bad argument");
    s->scene.width = width;
    s->scene.height = height;
    return 0;
}

int hpt_host_scene_destroy(hpt_host_scene* s) {
    delete s;
    return 0;
}

int hpt_host_scene_describe(const hpt_host_scene* s, hpt_scene_desc* out) {
    if (!s || !out) return fail("null argument");
    GUARD(*out = s->scene.describe())
}

int hpt_host_scene_prim_ids(const hpt_host_scene* s, int32_t* out) {
    if (!s || !out) return fail("null argument");
    GUARD({
        const FlatScene& f = s->scene.flat();
        std::memcpy(out, f.prim_of_tri.data(), f.prim_of_tri.size() * sizeof(int32_t));
    })
}

int hpt_host_bvh_dump(const hpt_host_scene* s, int mesh, int* count, float* bounds6, int32_t* payload, float* area, int32_t* miss) {
    if (!s || !count) return fail("null argument");
    GUARD({
        s->scene.flat();
        if (mesh >= s->scene.num_meshes) throw std::runtime_error("mesh index out of range");
        const BVHAccel* b = mesh < 0 ? s->scene.bvh : s->scene.mesh_bvhs[mesh];
        *count = (int)b->nodes.size();
        if (bounds6) {
            for (size_t i = 0; i < b->nodes.size(); ++i) {
                const FlatNode& n = b->nodes[i];
                bounds6[6 * i + 0] = n.bounds.pMin.x; bounds6[6 * i + 1] = n.bounds.pMin.y; bounds6[6 * i + 2] = n.bounds.pMin.z;
                bounds6[6 * i + 3] = n.bounds.pMax.x; bounds6[6 * i + 4] = n.bounds.pMax.y; bounds6[6 * i + 5] = n.bounds.pMax.z;
                payload[i] = n.object;
                area[i] = n.area;
                miss[i] = n.miss;
            }
        }
    })
}

int hpt_host_render(const hpt_host_scene* s, int device, int spp, int max_depth, int num_threads, int mode, float* rgb,
                    hpt_stats* stats) {
    if (!s || !rgb) return fail("null argument");
    if (mode != 1 && mode != 2) return fail("mode must be 1 (SingleKernel) or 2 (Streamed)");
    GUARD({
        Scene sc = s->scene;
        sc.maxDepth = max_depth;
        HipRenderer r;
        r.spp = spp;
        r.device = device;
        r.SetMode((HipRenderMode)mode);
        r.num_threads = num_threads;
        r.PrepareRender(sc);
        {
            // Render() prints the reference's timing lines; keep library calls quiet
            std::streambuf* old = std::cout.rdbuf(nullptr);
            try {
                r.Render(sc);
            } catch (...) {
                std::cout.rdbuf(old);
                throw;
            }
            std::cout.rdbuf(old);
        }
        if (stats) *stats = r.Stats();
        r.FinishRender(sc);
        std::memcpy(rgb, r.framebuffer.data(), r.framebuffer.size() * sizeof(Vec3));
    })
}

int hpt_host_render_devices(const hpt_host_scene* s, const int* devices, int num_devices, int spp, int max_depth, int num_threads,
                            int mode, int iterative, float* rgb, hpt_stats* stats) {
    if (!s || !rgb || !devices || num_devices < 1) return fail("null argument");
    if (mode != 1 && mode != 2) return fail("mode must be 1 (SingleKernel) or 2 (Streamed)");
    GUARD({
        Scene sc = s->scene;
        sc.maxDepth = max_depth;
        HipRenderer r;
        r.spp = spp;
        r.devices.assign(devices, devices + num_devices);
        r.SetMode((HipRenderMode)mode);
        r.num_threads = num_threads;
        r.PrepareRender(sc);
        if (iterative == 2) {
            // Render the first half of the samples, then add the rest one PathTrace at a time: the flow of a
            // caller that refines a finished picture (framebuffer = accumulated / spp, so spp is raised first)
            const int first = spp / 2;
            r.spp = first;
            std::streambuf* old = std::cout.rdbuf(nullptr);
            try {
                r.Render(sc);
            } catch (...) {
                std::cout.rdbuf(old);
                throw;
            }
            std::cout.rdbuf(old);
            r.spp = spp;
            for (int iter = first; iter < spp; ++iter) r.PathTrace(sc, iter);
        } else if (iterative) {
            for (int iter = 0; iter < spp; ++iter) r.PathTrace(sc, iter);
        } else {
            std::streambuf* old = std::cout.rdbuf(nullptr);
            try {
                r.Render(sc);
            } catch (...) {
                std::cout.rdbuf(old);
                throw;
            }
            std::cout.rdbuf(old);
        }
        if (stats) *stats = r.Stats();
        r.FinishRender(sc);
        std::memcpy(rgb, r.framebuffer.data(), r.framebuffer.size() * sizeof(Vec3));
    })
}

int hpt_host_render_iterative(const hpt_host_scene* s, int device, int spp, int max_depth, int num_threads, int mode,
                              float* rgb, hpt_stats* stats) {
    if (!s || !rgb) return fail("null argument");
    if (mode != 1 && mode != 2) return fail("mode must be 1 (SingleKernel) or 2 (Streamed)");
    GUARD({
        Scene sc = s->scene;
        sc.maxDepth = max_depth;
        HipRenderer r;
        r.spp = spp;
        r.device = device;
        r.SetMode((HipRenderMode)mode);
        r.num_threads = num_threads;
        r.PrepareRender(sc);
        for (int iter = 0; iter < spp; ++iter) r.PathTrace(sc, iter);
        if (stats) *stats = r.Stats();
        r.FinishRender(sc);  // downloads accumulated / spp
        std::memcpy(rgb, r.framebuffer.data(), r.framebuffer.size() * sizeof(Vec3));
    })
}

static bool to_bvh_build(int build, Scene::BvhBuild* out) {
    if (build == HPT_BVH_BUILD_HOST) *out = Scene::BvhBuild::Reference;
    else if (build == HPT_BVH_BUILD_GPU) *out = Scene::BvhBuild::Gpu;
    else if (build == HPT_HOST_BVH_BUILD_SAH) *out = Scene::BvhBuild::Sah;
    else if (build == HPT_HOST_BVH_BUILD_AUTO) *out = Scene::BvhBuild::Auto;
    else return false;
    return true;
}

int hpt_host_set_default_bvh_build(int build) {
    if (!to_bvh_build(build, &Scene::default_bvh_build)) return fail("build is none of HPT_BVH_BUILD_HOST, HPT_BVH_BUILD_GPU, HPT_HOST_BVH_BUILD_SAH, HPT_HOST_BVH_BUILD_AUTO");
    return 0;
}

int hpt_host_default_bvh_build(int* build) {
    if (!build) return fail("null argument");
    *build = Scene::default_bvh_build == Scene::BvhBuild::Gpu ? HPT_BVH_BUILD_GPU : Scene::default_bvh_build == Scene::BvhBuild::Sah ? HPT_HOST_BVH_BUILD_SAH
           : Scene::default_bvh_build == Scene::BvhBuild::Auto ? HPT_HOST_BVH_BUILD_AUTO : HPT_BVH_BUILD_HOST;
    return 0;
}

int hpt_host_scene_set_bvh_build(hpt_host_scene* s, int build) {
    if (!s) return fail("scene is null");
    if (!to_bvh_build(build, &s->scene.bvh_build)) return fail("build is none of HPT_BVH_BUILD_HOST, HPT_BVH_BUILD_GPU, HPT_HOST_BVH_BUILD_SAH, HPT_HOST_BVH_BUILD_AUTO");
    return 0;
}

int hpt_host_scene_bvh_build(const hpt_host_scene* s, int* build) {
    if (!s || !build) return fail("null argument");
    GUARD(*build = s->scene.flat().gpu_build ? HPT_BVH_BUILD_GPU : s->scene.flat().sah_build ? HPT_HOST_BVH_BUILD_SAH : HPT_BVH_BUILD_HOST)
}

int hpt_host_scene_intersect(const hpt_host_scene* s, int device, size_t n, const float* o, const float* d, const uint8_t* shadow,
                            const double* tmax, uint8_t* happened, int32_t* prim, double* distance, float* coords, float* normal,
                            float* triangle_area, int32_t* material_type) {
    if (!s || (n && (!o || !d))) return fail("null argument");
    GUARD({
        const_cast<Scene&>(s->scene).query_device = device;
        std::vector<Ray> rays(n);
        for (size_t i = 0; i < n; ++i)
            rays[i] = Ray(Vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), shadow && shadow[i],
                          shadow && shadow[i] && tmax ? tmax[i] : kDoubleInfinity);
        const std::vector<Intersection> hits = s->scene.intersect(rays);
        for (size_t i = 0; i < n; ++i) {
            const Intersection& h = hits[i];
            if (happened) happened[i] = h.happened ? 1 : 0;
            if (prim) prim[i] = h.primitive;
            if (distance) distance[i] = h.distance;
            if (coords) { coords[3 * i] = h.coords.x; coords[3 * i + 1] = h.coords.y; coords[3 * i + 2] = h.coords.z; }
            if (normal) { normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z; }
            if (triangle_area) triangle_area[i] = h.triangleArea;
            if (material_type) material_type[i] = h.happened ? h.m.type : -1;
        }
    })
}

int hpt_host_scene_cast_rays(const hpt_host_scene* s, int device, uint32_t seed, uint32_t integrator, size_t n, const float* o, const float* d,
                             const uint32_t* pixel_key, const uint32_t* sample_key, float* rgb) {
    if (!s || (n && (!o || !d || !pixel_key || !sample_key || !rgb))) return fail("null argument");
    GUARD({
        Scene& scene = const_cast<Scene&>(s->scene);
        scene.query_device = device;
        scene.query_seed = seed;
        scene.query_integrator = integrator;
        std::vector<Ray> rays(n);
        std::vector<Scene::PathKey> keys(n);
        for (size_t i = 0; i < n; ++i) {
            rays[i] = Ray(Vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
            keys[i].pixel = pixel_key[i];
            keys[i].sample = sample_key[i];
        }
        const std::vector<Vec3> radiance = s->scene.castRay(keys, rays);
        for (size_t i = 0; i < n; ++i) { rgb[3 * i] = radiance[i].x; rgb[3 * i + 1] = radiance[i].y; rgb[3 * i + 2] = radiance[i].z; }
    })
}

int hpt_host_material_eval(const hpt_material* material, size_t n, const float* ray_source_dir, const float* ray_out_dir,
                           const float* normal, float* bsdf, float* pdf) {
    if (!material || (n && (!ray_source_dir || !ray_out_dir || !normal))) return fail("null argument");
    GUARD({
        const Material m = to_material(material);
        for (size_t i = 0; i < n; ++i) {
            const Vec3 src(ray_source_dir[3 * i], ray_source_dir[3 * i + 1], ray_source_dir[3 * i + 2]);
            const Vec3 out(ray_out_dir[3 * i], ray_out_dir[3 * i + 1], ray_out_dir[3 * i + 2]);
            const Vec3 nrm(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
            if (bsdf) {
                const Vec3 f = m.bsdf(src, out, nrm);
                bsdf[3 * i] = f.x; bsdf[3 * i + 1] = f.y; bsdf[3 * i + 2] = f.z;
            }
            if (pdf) pdf[i] = m.pdf(src, out, nrm);
        }
    })
}

int hpt_host_material_sample(const hpt_material* material, size_t n, const float* u3, const float* ray_out_dir, const float* normal,
                             float* dir_out) {
    if (!material || (n && (!u3 || !ray_out_dir || !normal || !dir_out))) return fail("null argument");
    GUARD({
        const Material m = to_material(material);
        for (size_t i = 0; i < n; ++i) {
            // u3 is in the order Material::sample(RNG&, ...) draws: a Lambertian surface's first draw is its SECOND argument
            const bool swap = m.type == Material::Lambertian;
            const Vec3 w = m.sample(u3[3 * i + (swap ? 1 : 0)], u3[3 * i + (swap ? 0 : 1)], u3[3 * i + 2], Vec3(ray_out_dir[3 * i], ray_out_dir[3 * i + 1], ray_out_dir[3 * i + 2]),
                                    Vec3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]));
            dir_out[3 * i] = w.x; dir_out[3 * i + 1] = w.y; dir_out[3 * i + 2] = w.z;
        }
    })
}

int hpt_host_rng_floats(int iter, int index, int count, float* out) {
    if (count && !out) return fail("null argument");
    GUARD({
        RNG rng(iter, index, 0);
        for (int i = 0; i < count; ++i) out[i] = rng.sample1D();
    })
}

int hpt_host_save_ppm(const char* path, const float* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return fail("bad argument");
    GUARD({
        std::vector<Vec3> fb((size_t)width * height);
        std::memcpy(static_cast<void*>(fb.data()), rgb, fb.size() * sizeof(Vec3));
        Utility::SavePPM(path, fb, width, height);
    })
}

int hpt_host_save_pfm(const char* path, const float* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return fail("bad argument");
    GUARD({
        std::vector<Vec3> fb((size_t)width * height);
        std::memcpy(static_cast<void*>(fb.data()), rgb, fb.size() * sizeof(Vec3));
        Utility::SavePFM(path, fb, width, height);
    })
}

}  // extern "C"

/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference path
 * tracer's hot path.  Never linked into, imported by or executed from the
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * Parity status: PINNED.  With rng_mode = ORC_RNG_MT and trig_mode =
 * ORC_TRIG_LIBM the restatement reproduces the compiled reference
 * (oracle/_ref/libref.so, built from the unmodified sources) bit for bit:
 * BVH topology/bounds, closest-hit (prim id, t) and any-hit results on fixed
 * ray batches, and whole rendered images.  The committed fixtures under
 * tests/golden/ were produced by that reference build (tests/golden/make_golden.py).
 */
#ifndef HPT_ORACLE_H
#define HPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* mirrors the reference's Material (common/Material.hpp:130-142) */
typedef struct {
    int32_t type; /* 0 Lambertian, 1 MetallicWorkflow, 2 Glass */
    float emission[3];
    float albedo[3];
    float metallic;
    float roughness;
    float ior;
} orc_material;

enum { ORC_RNG_MT = 0, ORC_RNG_PHILOX = 1 };
enum { ORC_TRIG_LIBM = 0, ORC_TRIG_PORTABLE = 1 };

typedef struct {
    uint64_t closest_queries; /* Scene::intersect calls for path rays      */
    uint64_t shadow_queries;  /* Scene::intersect calls for shadow rays    */
    uint64_t box_tests;       /* Bounds3::IntersectP calls                 */
    uint64_t tri_tests;       /* Triangle::getIntersection calls           */
    uint64_t paths;
    uint64_t stalled_bounces; /* bsdfSamplePdf < 1e-5 (Scene.hpp:238-240)  */
    uint64_t nan_resets;      /* Scene.hpp:269-271                         */
} orc_stats;

orc_scene* orc_scene_create(int width, int height, const float cam_pos[3], double fov_deg);
void orc_scene_destroy(orc_scene*);
/* Scene::CreateBuiltinScene(CornellBox) (common/Scene.cpp:25-73); OBJ files are
 * read from <models_dir>/cornellbox/.  width/height default to 512 there. */
orc_scene* orc_scene_create_cornell(const char* models_dir, int width, int height);
int orc_scene_add_obj(orc_scene*, const char* obj_path, const orc_material*);
/* triangle soup: 9 floats per triangle (v0 v1 v2) */
int orc_scene_add_triangles(orc_scene*, const float* verts9, int num_tris, const orc_material*);
/* a sphere as an object of the top-level BVH (common/Sphere.hpp, commented out in the reference) */
int orc_scene_add_sphere(orc_scene*, const float center[3], float radius, const orc_material*);
/* enableRR / pathRegularization of Scene::TracePath (common/Scene.hpp:141-142; compile-time false there) */
void orc_scene_set_integrator(orc_scene*, int russian_roulette, int path_regularization);
/* solveQuadratic (common/MathUtils.hpp:86-100) */
int orc_solve_quadratic(float a, float b, float c, float* x0, float* x1);
/* Material::regularize (common/Material.hpp:164-171): roughness in, roughness out */
float orc_material_regularize(float roughness);
int orc_scene_build(orc_scene*);
void orc_scene_set_resolution(orc_scene*, int width, int height);

void orc_scene_info(const orc_scene*, int* num_meshes, int* num_tris, int* width, int* height,
                    float* sum_light_area, float* sum_light_power, int* num_lights, float* cam_scale);
void orc_mesh_info(const orc_scene*, int mesh, int* num_tris, float* area, float* bbox6, float* bvh_root_area);
void orc_mesh_triangles(const orc_scene*, int mesh, float* out13);
/* same contract as ref_bvh_dump in ref_harness.cpp */
int orc_bvh_dump(const orc_scene*, int mesh, float* bounds6, int* payload, float* area, int* miss_rank);

/* closest-hit (shadow[r]==0) or any-hit (shadow[r]!=0, tmax[r] = shadowDistance)
 * queries; prim = global primitive id (mesh base + triangle index, Scene::Add
 * order) or -1; t = Intersection::distance (1e30f on a miss).  counts (nullable)
 * = per ray {box tests, triangle tests}. */
void orc_intersect_batch(const orc_scene*, int64_t n, const float* o, const float* d, const uint8_t* shadow,
                         const double* tmax, int32_t* prim, double* t, int32_t* counts, int threads);

/* Renderer::Render (common/Renderer.cpp:16-51) over samples
 * [sample_begin, sample_begin + sample_count):
 *   rgb_out  (nullable): float[3*W*H], += castRay / spp_total in sample order
 *   accum_out(nullable): int64[3*W*H], += fixed-point radiance (orc_radiance_to_fixed) */
void orc_render(const orc_scene*, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int sample_begin,
                int sample_count, int spp_total, int threads, float* rgb_out, int64_t* accum_out, orc_stats* stats);

/* radiance of individual (sample, pixel) paths, no normalisation */
void orc_cast_paths(const orc_scene*, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int64_t n,
                    const int32_t* sample_k, const int32_t* pixel_m, float* rgb_out, int threads, orc_stats* stats);

/* Scene::castRay (common/Scene.hpp:274-285) for caller-supplied eye rays (o, d: 3 floats each); ray q draws from the
 * stream of (sample_k[q], pixel_m[q]) */
void orc_cast_rays(const orc_scene*, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int64_t n, const float* o,
                   const float* d, const int32_t* sample_k, const int32_t* pixel_m, float* rgb_out, int threads, orc_stats* stats);

/* the rays camera paths (sample_k[q], pixel_m[q]) trace at bounce `bounce` (0 = camera rays): an incoherent batch as a
 * render produces it; alive_out[q] = 0 when the path ended earlier (o/d untouched) */
void orc_capture_bounce_rays(const orc_scene*, int rng_mode, int trig_mode, uint32_t seed, int bounce, int64_t n,
                             const int32_t* sample_k, const int32_t* pixel_m, float* o_out, float* d_out, uint8_t* alive_out, int threads);

/* ---- unit-level entry points (shading vectors) ---- */
void orc_rng_floats(int rng_mode, uint32_t seed, int iter, int index, int count, float* out);
void orc_material_eval(const orc_material*, int64_t n, const float* wi, const float* wo, const float* nrm,
                       float* bsdf_out, float* pdf_out);
/* u: 3 floats per sample in the order the reference RNG stream produced them */
void orc_material_sample(const orc_material*, int trig_mode, int64_t n, const float* u3, const float* wo,
                         const float* nrm, float* wi_out);
void orc_sample_hemisphere_cosine(int trig_mode, int64_t n, const float* nrm, const float* rx, const float* ry,
                                  float* out);
/* u: 4 floats per sample in stream order; out11 as ref_sample_light */
void orc_sample_light(const orc_scene*, int64_t n, const float* u4, float* out11);
float orc_power_heuristic(float f, float g);
int orc_tonemap(float v);

/* ---- contract of the new core (not from the reference) ---- */
/* Philox4x32-10: ctr[4], key[2] -> out[4] */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* the keyed uniform in [0,1): (pixel, sample) key, (depth, slot) counter */
float orc_philox_uniform(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot, uint32_t seed);
int64_t orc_radiance_to_fixed(float v);
float orc_fixed_to_radiance(int64_t acc, int spp_total);
void orc_portable_trig(int64_t n, const float* x, float* sin_out, float* cos_out, float* acos_out);

#ifdef __cplusplus
}
#endif
#endif

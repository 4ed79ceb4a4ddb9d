/* hpt_host -- C view of the C++ host side (Scene / BVHAccel / HipRenderer in
 * cudapathtracer_amd/host/), for callers that are not C++ (the Python package
 * binds it with ctypes).  It adds no behaviour of its own: scenes are built by
 * the same Scene::Add / Scene::buildBVH code the gpu_path_tracer CLI uses.
 *
 * Every function returns 0 on success; hpt_host_last_error() describes the
 * last failure of the calling thread.
 */
#ifndef HPT_HOST_H
#define HPT_HOST_H

#include "hpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpt_host_scene hpt_host_scene;

const char* hpt_host_last_error(void);

/* Scene::CreateBuiltinScene(CornellBox, max_depth) (common/Scene.cpp:25-73);
 * models_dir = directory that contains cornellbox/ (NULL = "models" or
 * $HPT_MODELS_DIR); width/height <= 0 keep the built-in 512x512. */
int hpt_host_scene_cornell(const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out);
/* Scene::CreateBuiltinScene(scene_id, max_depth): 0 = CornellBox as above, 1 = BunnyInCornell
 * (no counterpart in the reference, which ships models/bunny but no scene that uses it),
 * 2 = GlassCornell (CornellBox with the short box's material type switched to Glass; the
 * reference sets that material's glass parameters, common/Scene.cpp:45-47, but not its type). */
int hpt_host_scene_builtin(int scene_id, const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out);
/* empty scene: Scene(w, h) + camera (common/Scene.hpp:23-45) */
/* Scene::CreateNamedScene: "cornell" | "bunny" | "glass" | "heightfield:<n>" | "file:<scene file>"
 * (what the CLI's HPT_SCENE accepts; scene file format: host/SceneFile.cpp).  width/height 0 keep
 * the scene's own resolution. */
int hpt_host_scene_named(const char* name, const char* models_dir, int width, int height, int max_depth, hpt_host_scene** out);
int hpt_host_scene_info(const hpt_host_scene* s, int* width, int* height, char* name, int name_cap);
/* MeshTriangle with a Transform: translation, rotation in degrees about x, y, z, scale, composed as
 * Math::buildTransformationMatrix (common/MathUtils.hpp:109-116), which the reference never calls. */
int hpt_host_scene_add_obj_transformed(hpt_host_scene* s, const char* obj_path, const hpt_material* material, const float translation[3],
                                       const float rotation_deg[3], const float scale[3]);
/* Scene::Add(Sphere*): a sphere as an object of the top-level BVH (common/Sphere.hpp is commented out
 * in the reference; no counterpart in its Scene).  The material must not emit. */
int hpt_host_scene_add_sphere(hpt_host_scene* s, const float center[3], float radius, const hpt_material* material);
int hpt_host_scene_create(int width, int height, const float cam_pos[3], double fov_deg, int max_depth,
                          hpt_host_scene** out);
/* Scene::Add(new MeshTriangle(...)) (common/Scene.hpp:47-56, common/Triangle.cpp:5-61) */
int hpt_host_scene_add_obj(hpt_host_scene* s, const char* obj_path, const hpt_material* material);
int hpt_host_scene_add_triangles(hpt_host_scene* s, const float* verts9, int num_tris, const hpt_material* material);
/* Scene::buildBVH (common/Scene.cpp:9-23) + flattening */
int hpt_host_scene_build(hpt_host_scene* s);
int hpt_host_scene_set_resolution(hpt_host_scene* s, int width, int height);
int hpt_host_scene_destroy(hpt_host_scene* s);

/* the flattened scene; pointers stay valid until the scene is destroyed or rebuilt */
int hpt_host_scene_describe(const hpt_host_scene* s, hpt_scene_desc* out);
/* leaf order -> global primitive id (num_tris entries) */
int hpt_host_scene_prim_ids(const hpt_host_scene* s, int32_t* out);
/* One level of the two-level BVH before flattening, pre-order: mesh = -1 for
 * the top level (payload = mesh index), else the mesh's own BVH (payload =
 * triangle index within the mesh); payload -1 = interior.  bounds6 = pMin,pMax.
 * Returns the node count through *count; pass NULL arrays to only count. */
int hpt_host_bvh_dump(const hpt_host_scene* s, int mesh, int* count, float* bounds6, int32_t* payload, float* area,
                      int32_t* miss);

/* HipRenderer: PrepareRender + Render + FinishRender on `device`; rgb =
 * float[3*width*height] (Renderer::framebuffer).  stats may be NULL. */
int hpt_host_render(const hpt_host_scene* s, int device, int spp, int max_depth, int num_threads, int mode, float* rgb,
                    hpt_stats* stats);
/* The same picture through HipRenderer::PathTrace: PrepareRender, PathTrace(scene, iter) for iter in
 * [0, spp) -- the per-sample loop of CudaRenderer::Render (gpu_path_tracer/CudaRenderer.cu:58-61) --
 * then FinishRender, which downloads the framebuffer once. */
int hpt_host_render_iterative(const hpt_host_scene* s, int device, int spp, int max_depth, int num_threads, int mode,
                              float* rgb, hpt_stats* stats);
/* The same with the samples sharded over the GPUs listed (HipRenderer::devices; an index may repeat):
 * one host thread per entry renders a contiguous sample range, hpt_accum_reduce sums the buffers
 * onto devices[0].  iterative = 1: through PathTrace, sample iter on entry iter mod num_devices;
 * iterative = 2: Render() with spp / 2 samples, then PathTrace for the remaining ones (a finished picture
 * that is refined further; every sample must be counted once). */
int hpt_host_render_devices(const hpt_host_scene* s, const int* devices, int num_devices, int spp, int max_depth,
                            int num_threads, int mode, int iterative, float* rgb, hpt_stats* stats);
/* A third way to build the scene's tree, on the host like HPT_BVH_BUILD_HOST (the device library sees an ordinary
 * host-built scene): BVHAccel::SplitMethod::SAH -- declared by the reference (common/BVH.hpp:28), implemented nowhere in it --
 * for the top level and every mesh that does not emit, with the children ordered for a short traversal stack; emitting
 * meshes keep the reference's tree (Scene::sampleLight descends it).  Same leaves as the reference's tree and the
 * triangle records in the reference's traversal order: distances, occlusion answers AND primitive ids are the
 * reference's for every ray with finite reciprocal direction components; fewer boxes are tested per ray.  The rays
 * that leaves out (a zero direction component) walk the reference's own tree, which the scene description carries
 * beside the other (hpt_scene_desc::ref_nodes): every result is the reference's.
 * HPT_HOST_BVH_BUILD_AUTO (the default): HPT_HOST_BVH_BUILD_SAH for scenes of more than 64 primitives, else
 * HPT_BVH_BUILD_HOST (such scenes are kept in LDS under trees the device library makes itself). */
#define HPT_HOST_BVH_BUILD_SAH 2
#define HPT_HOST_BVH_BUILD_AUTO 3
/* Scene::default_bvh_build: scenes created from now on flatten for HPT_HOST_BVH_BUILD_AUTO (default), HPT_BVH_BUILD_HOST,
 * HPT_BVH_BUILD_GPU (include/hpt.h) or HPT_HOST_BVH_BUILD_SAH; the renderer picks the matching upload. */
int hpt_host_set_default_bvh_build(int build);
int hpt_host_default_bvh_build(int* build);
/* Scene::bvh_build of one scene (takes effect at its next hpt_host_scene_build), and how the scene was last built */
int hpt_host_scene_set_bvh_build(hpt_host_scene* s, int build);
int hpt_host_scene_bvh_build(const hpt_host_scene* s, int* build);

/* Scene::intersect (common/Scene.hpp:95-97) on a batch of rays, run on GPU `device` (hpt_intersect_batch on a context the
 * scene keeps).  shadow[i] != 0 = any-hit query against tmax[i] (Ray::shadowTest / shadowDistance).  Outputs (each may be
 * NULL) are the fields of Intersection (common/Intersection.hpp:14-35): happened, distance, coords = ray(distance), normal,
 * triangleArea, m.type; prim = global primitive id, -1 on a miss. */
int hpt_host_scene_intersect(const hpt_host_scene* s, int device, size_t n, const float* o, const float* d, const uint8_t* shadow,
                             const double* tmax, uint8_t* happened, int32_t* prim, double* distance, float* coords, float* normal,
                             float* triangle_area, int32_t* material_type);

/* Scene::castRay (common/Scene.hpp:274-285) on a batch of eye rays, run on GPU `device` (hpt_cast_rays on the context the
 * scene keeps): rgb[3*n] = radiance after up to the scene's maxDepth bounces.  Ray i draws from the keyed random stream of
 * the camera path (pixel_key[i], sample_key[i]) under `seed`; integrator: HPT_INTEGRATOR_* flags (include/hpt.h). */
int hpt_host_scene_cast_rays(const hpt_host_scene* s, int device, uint32_t seed, uint32_t integrator, size_t n, const float* o,
                             const float* d, const uint32_t* pixel_key, const uint32_t* sample_key, float* rgb);

/* Material::bsdf / Material::pdf (common/Material.hpp:257-384) and Material::sample (:186-219, with the three uniform draws
 * given per sample in u3, in the order Material::sample(RNG&) draws them) of the host Material class, n directions each; libm transcendentals like the
 * reference's CPU build.  The renderer does not use them (materials are evaluated on the device). */
int hpt_host_material_eval(const hpt_material* material, size_t n, const float* ray_source_dir, const float* ray_out_dir,
                           const float* normal, float* bsdf, float* pdf);
int hpt_host_material_sample(const hpt_material* material, size_t n, const float* u3, const float* ray_out_dir, const float* normal,
                             float* dir_out);
/* the first `count` floats of RNG(iter, index, 0) (common/MathUtils.hpp:22-36), the reference's host random stream */
int hpt_host_rng_floats(int iter, int index, int count, float* out);

/* Utility::SavePPM (common/Utility.cpp:4-22) */
int hpt_host_save_ppm(const char* path, const float* rgb, int width, int height);
/* Utility::SavePFM: linear float RGB as a portable float map (no counterpart in the reference) */
int hpt_host_save_pfm(const char* path, const float* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* HPT_HOST_H */

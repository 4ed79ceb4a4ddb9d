/* hpt -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary: everything the reference's host code reaches
 * through `CudaRenderer` and the CUDA_PORTABLE MallocCuda/FreeCuda deep copy is
 * reached through these entry points instead.  Plain C types only: pointers,
 * sizes, POD structs.  No C++ or torch types cross the boundary.
 *
 * Conventions
 *  - every function returns 0 on success, a non-zero hpt_status otherwise;
 *    hpt_last_error() returns a thread-local message for the last failure
 *    (the reference has no working error path: checkCUDAError is compiled
 *    out, gpu_path_tracer/CudaUtils.h:8-19).
 *  - one hpt_ctx per GPU; a context may be used from one host thread at a
 *    time; there is no global mutable state (contrast the reference's
 *    file-static host->device pointer registry, gpu_path_tracer/CudaPortable.cu:13-50).
 *  - host arrays passed to hpt_scene_upload are borrowed for the duration of
 *    the call only.
 *  - `stream` arguments are hipStream_t handles passed as void* (NULL = the
 *    default stream); device-pointer arguments are named *_dev.
 */
#ifndef HPT_H
#define HPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpt_ctx hpt_ctx;

typedef enum {
    HPT_OK = 0,
    HPT_ERR_INVALID_ARGUMENT = 1,
    HPT_ERR_HIP = 2,           /* a HIP runtime call failed; see hpt_last_error() */
    HPT_ERR_NO_SCENE = 3,      /* render/intersect before hpt_scene_upload */
    HPT_ERR_OUT_OF_MEMORY = 4, /* a device allocation found no room (scene upload, wavefront queues, ray staging) */
    HPT_ERR_TOO_LARGE = 5      /* fixed-capacity entry point: input exceeds its limit */
} hpt_status;

/* Render modes: the values of the reference's CudaRenderMode
 * (gpu_path_tracer/CudaRenderer.hpp:9-13) and of argv[4] of gpu_path_tracer. */
enum { HPT_MODE_SINGLE_KERNEL = 1, HPT_MODE_STREAMED = 2 };

/* Material table entry; the fields of the reference's Material
 * (common/Material.hpp:130-142).  type: 0 Lambertian, 1 MetallicWorkflow, 2 Glass
 * (rough dielectric, Material.hpp:200-215, :277-309, :348-381; ior must be set). */
typedef struct {
    int32_t type;
    float emission[3];
    float albedo[3];
    float metallic;
    float roughness;
    float ior;
    float _pad[2];
} hpt_material; /* 48 bytes */

/* One emitting mesh, in Scene::Add order (the order Scene::sampleLight walks,
 * common/Scene.hpp:99-120). */
typedef struct {
    float area;         /* MeshTriangle::area: triangle areas summed in file order      */
    float root_area;    /* mesh BVH root->area: child sums in tree order (pdf = 1/this) */
    uint32_t root_node; /* index of the mesh's BVH root in the flattened node array     */
    uint32_t material;
} hpt_light;

/* Flattened scene.  Replaces Scene::MallocCuda's node-by-node deep copy
 * (gpu_path_tracer/CudaPortable.cu:157-190, :283-318).
 *
 * nodes: the reference's two-level BVH (common/BVH.cpp:81-168) flattened into
 *   ONE array in pre-order, mesh BVHs spliced in at the top-level leaves.
 *   The traversal is "threaded" like the reference's nextIfHit/nextIfMiss
 *   links (common/BVH.cpp:53-64): after a box HIT the next node is i+1, after
 *   a MISS it is the node's miss link; num_nodes terminates.
 *   Record = 8 floats, 32 bytes:  pMin.xyz, bits(int32 miss link),
 *                                 pMax.xyz, bits(int32 triangle index | -1)
 *   The array may hold ANY binary tree in this form whose boxes are the unions of the boxes below them and
 *   whose leaves are the reference's (one triangle or sphere each): a ray with finite reciprocal direction
 *   components that hits a leaf's box hits every box that encloses it, so the triangles tested -- and with them
 *   distances, occlusion answers and pictures -- do not depend on the interior nodes.  The host library's
 *   Scene::BvhBuild::Sah emits such a tree (surface-area splits, fewer boxes tested per ray), with the reference's
 *   tree beside it in `ref_nodes` for the rays the argument leaves out.  The light table's
 *   root_node subtrees must be the reference's mesh trees: BVHAccel::getSample descends them.
 * tris: in leaf order of the REFERENCE's tree (= the order the reference traversal meets them; equal-t
 *   ties resolve to the lower index, as the strict `<` of common/BVH.hpp:130 does; a tree of another
 *   shape keeps the records in this order and refers to them by index).
 *   Record = 16 floats, 64 bytes: v0.xyz, area | e1.xyz, bits(material index) |
 *                                 e2.xyz, bits(prim id) | normal.xyz, bits(mesh index)
 *   with e1, e2, normal, area precomputed as common/Triangle.hpp:52-59 does and
 *   prim id = mesh base + triangle index within the mesh (Scene::Add order).
 *   A record whose word 15 has bit 31 set is a SPHERE (common/Sphere.hpp, commented out in the
 *   reference; north_star names the primitive):  center.xyz, 0 | radius, radius^2, 0, bits(material)
 *   | 0, 0, 0, bits(prim id) | 0, 0, 0, bits(mesh | 0x80000000); its leaf's box is center -+ radius.
 *   Spheres do not emit.  Scenes with spheres use the two-child-record traversal whatever their size.
 * tri_v12: v1.xyz, 0, v2.xyz, 0 per triangle (Triangle::Sample needs v1, v2
 *   themselves, common/Triangle.hpp:75-82).
 * node_area: BVHBuildNode::area per node (common/BVH.cpp:111,121,164), used by
 *   the area-proportional descent of BVHAccel::getSample (common/BVH.hpp:147-177).
 * All arrays must be 64-byte aligned. */
typedef struct {
    uint32_t num_nodes;
    const float* nodes;     /* [num_nodes][8]  */
    const float* node_area; /* [num_nodes]     */
    uint32_t num_tris;
    const float* tris;      /* [num_tris][16]  */
    const float* tri_v12;   /* [num_tris][8]   */
    uint32_t num_materials;
    const hpt_material* materials;
    uint32_t num_lights;
    const hpt_light* lights;
    float sum_light_area;   /* Scene::sumLightArea  (common/Scene.hpp:52) */
    float sum_light_power;  /* Scene::sumLightPower (common/Scene.hpp:53) */
    float cam_pos[3];       /* Scene::camPos */
    float cam_scale;        /* tan(fov/2) as the host renderer computes it (common/Renderer.cpp:20) */
    int32_t width, height;  /* Scene::width/height; aspect = width/(float)height */
    /* Optional (0 / NULL: `nodes` IS the reference's tree).  When `nodes` holds another tree over the same leaves
     * (surface-area splits), the reference's own flattened tree (common/BVH.cpp:81-168), in the format of `nodes`
     * with leaf words that index the same `tris`.  Rays with a zero direction component walk THIS tree: for them a
     * slab test can fail on 0 * inf = NaN whatever the boxes' nesting (common/Bounds3.hpp:93-118), so the leaves
     * they reach depend on the interior nodes, and with the reference's nodes they reach the reference's.  Every
     * other ray walks `nodes`.  With it, a scene with any tree in `nodes` gives the reference's results for every ray. */
    uint32_t num_ref_nodes;
    const float* ref_nodes; /* [num_ref_nodes][8] */
} hpt_scene_desc;

typedef struct {
    uint64_t closest_queries; /* closest-hit BVH queries  */
    uint64_t shadow_queries;  /* any-hit (shadow) queries */
    uint64_t paths;           /* camera paths started     */
    double kernel_ms;         /* HIP-event time of hpt_render's kernels, summed since the last hpt_stats_reset */
    uint32_t launches;        /* kernel launches of hpt_render, summed since the last hpt_stats_reset        */
    uint32_t variant;         /* traversal variant chosen for the uploaded scene: 0 threaded walk through L1/L2,
                                 1 the same from LDS, 2/3 LDS with a 32-/64-bit leaf mask,
                                 5 two-child records + stack through L1/L2, 6 the same from LDS,
                                 7 two-child records with distance culling (HPT_TRAVERSAL_CULLED)   */
    uint64_t box_tests;       /* box / triangle tests the kernels executed: counted only by the diagnostic     */
    uint64_t tri_tests;       /* build of the library (-DHPT_COUNT_TESTS, `make count`); 0 in the product build */
} hpt_stats;

const char* hpt_last_error(void);
const char* hpt_version(void);

/* cudaGetDeviceCount analogue (the reference always uses device 0). */
int hpt_device_count(int* count);

/* Replaces CudaRenderer::PrepareRender's allocations (gpu_path_tracer/CudaRenderer.cu:19-35)
 * and CudaRenderer::FinishRender (:37-49). */
int hpt_context_create(int device, hpt_ctx** out);
int hpt_context_destroy(hpt_ctx* ctx);

/* Replaces Scene::MallocCuda / Scene::FreeCuda (gpu_path_tracer/CudaPortable.cu:283-332).
 * Uploading again replaces the previous scene. Also (re)allocates and clears
 * the context's own accumulation buffer (3 x int64 per pixel). */
int hpt_scene_upload(hpt_ctx* ctx, const hpt_scene_desc* scene);

/* Options that have no counterpart in the reference CLI.
 *   seed: word 2 of the Philox counter (default 0)
 *   wavefront_capacity: paths in flight per streamed batch (default 1<<28; the two queues take 128 bytes per path of
 *   a batch and are allocated for what a call needs: 32 GiB for a call of 2^28 paths or more; when the card cannot
 *   give that much the batch is halved until the queues fit)   */
int hpt_set_seed(hpt_ctx* ctx, uint32_t seed);
int hpt_set_wavefront_capacity(hpt_ctx* ctx, uint64_t paths);

/* The two steps of Scene::TracePath that the reference switches off at compile time
 * (`const bool pathRegularization = false, enableRR = false`, common/Scene.hpp:141-142), as run-time
 * options; takes effect at the next hpt_render.  Restated as written there (parity unpinned: no
 * input reaches them in the reference):
 *   HPT_INTEGRATOR_RUSSIAN_ROULETTE  :252-264 -- one more random draw per bounce; from the third
 *     bounce on, throughput /= RussianRoulette (0.8) unless the draw falls below 1 - RussianRoulette.
 *     As in the reference, such a bounce only goes without the compensation: the `break` there
 *     leaves the function's while(true) wrapper, it does not end the path.
 *   HPT_INTEGRATOR_REGULARIZE        :186-189 -- once a path has left a non-specular surface, materials
 *     are evaluated after Material::regularize() (common/Material.hpp:164-171). */
#define HPT_INTEGRATOR_RUSSIAN_ROULETTE 1u
#define HPT_INTEGRATOR_REGULARIZE 2u
int hpt_set_integrator(hpt_ctx* ctx, uint32_t flags);

/* Traversal order for scenes of more than 64 triangles; takes effect at the next hpt_scene_upload.
 *   HPT_TRAVERSAL_REFERENCE (default): exactly the boxes and triangles BVHAccel::getIntersection
 *     (common/BVH.hpp:91-145) tests: no ordering, no distance culling; results bit-exact.
 *   HPT_TRAVERSAL_CULLED: nearer child first, boxes behind the ray origin or beyond the closest
 *     hit found so far (shadow rays: beyond shadowDistance) skipped -- the "near-first + t-culling"
 *     the reference's README lists as future work.  Same hit except where float rounding puts a
 *     triangle's t outside its own box's slab range by more than 1e-4 of the scene diagonal.
 * No counterpart in the reference. */
#define HPT_TRAVERSAL_REFERENCE 0
#define HPT_TRAVERSAL_CULLED 1
int hpt_set_traversal(hpt_ctx* ctx, int traversal);

/* Who derives the traversal records (two boxes per record) at the next hpt_scene_upload.
 *   HPT_BVH_BUILD_HOST (default): from `nodes`, i.e. from the tree the host built by the reference's rules
 *     (common/BVH.cpp:81-168); every result is the reference's, ties between equal distances included.
 *   HPT_BVH_BUILD_GPU: on the device, from `tris` / `tri_v12` alone, by the same rule (sort by centroid along
 *     the longest axis of the centroid bounds, split in the middle) run level by level for all objects at once,
 *     under a tree over the objects' boxes -- the GPU BVH build the reference's README lists as future work
 *     (README.md:87).  `tris` need not be in leaf order then: an object is a run of records with the same object
 *     word (the host side passes primitive-id order), and `nodes` / `node_area` need only hold the trees of the
 *     emitting meshes (lights[].root_node; Scene::sampleLight descends them), leaf words = indices into `tris`.  The leaf boxes are the reference's and boxes nest, so for rays without a zero
 *     direction component the triangles tested, every closest distance and every occlusion answer are the
 *     reference's; when two triangles are hit at bitwise the same distance, the lower index in `tris` is
 *     reported where the reference reports the one its own tree meets first.  Needs >= 2 primitives.
 * hpt_scene_build_info: device time of the last GPU build (0 for a host build), number of records, stack
 * entries a walk can need.  No counterpart in the reference. */
#define HPT_BVH_BUILD_HOST 0
#define HPT_BVH_BUILD_GPU 1
int hpt_set_bvh_build(hpt_ctx* ctx, int build);
int hpt_scene_build_info(hpt_ctx* ctx, double* gpu_build_ms, uint32_t* num_records, uint32_t* stack_depth);

/* thrust::fill(framebuffer, 0) of CudaRenderer::Render (gpu_path_tracer/CudaRenderer.cu:56).
 * accum_dev = NULL clears the context's own buffer. */
int hpt_accum_clear(hpt_ctx* ctx, int64_t* accum_dev, void* stream);

/* Replaces the per-sample loop of CudaRenderer::Render / CudaRenderer::PathTrace
 * (gpu_path_tracer/CudaRenderer.cu:58-61, :76-94) and the kernels behind it
 * (SingleKernelRayTracing, GenerateCameraRay, ComputeIntersections,
 * IntegratePathSegment, the two thrust compactions, FinalGather;
 * gpu_path_tracer/CudaPathTrace.cu:14-160).
 *
 * Traces samples [sample_begin, sample_begin + sample_count) of every pixel
 * and ADDS each path's radiance, in 40.24 fixed point, to accum_dev
 * (int64[3*width*height], row-major pixels, RGB interleaved; NULL = the
 * context's own buffer).  PathTrace(scene, iter) is sample_begin = iter,
 * sample_count = 1.  The sample index is part of the RNG key, so any split of
 * a sample range over calls, launches, contexts or GPUs yields the same sums.
 * One render at a time per context: the work-unit counter and the wavefront
 * queues belong to the context, so a call on another stream than the previous
 * call's first makes that stream wait for the previous render (contexts are
 * independent of each other; use one per stream for concurrent renders).
 *   mode:  HPT_MODE_SINGLE_KERNEL | HPT_MODE_STREAMED
 *   block: workgroup size (the CLI's <threads>), 1..1024
 *   max_depth: Scene::maxDepth, 0..16384
 * Asynchronous with respect to the host (work is queued on `stream`). */
int hpt_render(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count,
               int64_t* accum_dev, void* stream);

/* hpt_render for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arguments, same kernels and bit-identical sums as hpt_render, but every buffer beyond the
 * caller's accumulators -- the work-unit counter, the per-segment live counters and the two
 * wavefront path queues -- lies in a 32 MiB + 17 KiB array that is part of the library's code
 * object (each GPU the library runs on holds its copy until the process ends), and the call takes
 * no timing events: it calls no hipMalloc, hipFree or hipEventCreate.  hpt_stats.kernel_ms does not
 * include these renders; the ray counters and `launches` do.
 *   Limit: at most 2^18 = 262 144 camera paths per call, counted as
 *   ceil(width/8) * ceil(height/8) * 64 * sample_count (512 x 512 pixels x 1 sample, 64 x 64 x 64
 *   samples, ...); beyond that the call returns HPT_ERR_TOO_LARGE and does nothing.
 *   All calls on one GPU share that GPU's array, whatever their context or stream.  The library orders
 *   them itself: a call makes its stream wait for the previous hpt_render_static on that GPU (one event
 *   per GPU, created with the GPU's first context; nothing is allocated in the call) -- so calls from
 *   several contexts or streams are safe but run one after the other.  A stream under graph capture is
 *   not ordered this way (an event recorded there belongs to the graph): keep such calls on one stream. */
int hpt_render_static(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count,
                      int64_t* accum_dev, void* stream);

/* Replaces the cudaMemcpy D2H of CudaRenderer::Render (gpu_path_tracer/CudaRenderer.cu:63-67):
 * converts the accumulators to float RGB divided by spp_total (what
 * Renderer::framebuffer holds in the reference) and copies them to host
 * memory.  Synchronises `stream`. */
int hpt_resolve(hpt_ctx* ctx, const int64_t* accum_dev, uint32_t spp_total, float* rgb_host, void* stream);
/* same, result left in device memory (float[3*width*height]) */
int hpt_resolve_dev(hpt_ctx* ctx, const int64_t* accum_dev, uint32_t spp_total, float* rgb_dev, void* stream);

/* BVH queries on a fixed ray batch; the device counterpart of Scene::intersect
 * (common/Scene.hpp:95-97 -> BVHAccel::Intersect, common/BVH.hpp:82-145).
 *   o, d:   float[3*n] origins / directions
 *   shadow: NULL or uint8[n]; non-zero = any-hit query with tmax[n] = Ray::shadowDistance
 *   prim:   int32[n]  global primitive id, -1 = miss
 *   t:      double[n] Intersection::distance (1e30f on a miss)
 * Closest-hit queries return the reference's (prim, t) exactly, ties included.  An any-hit query
 * guarantees the reference's ANSWER -- occluded (prim >= 0) or not -- and a (prim, t) of some
 * triangle that passes the reference's occlusion test; when several triangles pass it, scenes of
 * more than 64 triangles may name another of them than BVHAccel::getIntersection's first in
 * pre-order (their walk queues both children's leaves before descending).  The integrator only
 * uses the answer (common/Scene.hpp:205-207).
 * Host-pointer form (copies in and out, synchronous) and device-pointer form
 * (asynchronous on `stream`). */
int hpt_intersect_batch(hpt_ctx* ctx, size_t n, const float* o, const float* d, const uint8_t* shadow,
                        const double* tmax, int32_t* prim, double* t);
int hpt_intersect_batch_dev(hpt_ctx* ctx, size_t n, const float* o_dev, const float* d_dev, const uint8_t* shadow_dev,
                            const double* tmax_dev, int32_t* prim_dev, double* t_dev, int block, void* stream);

/* Radiance along caller-supplied eye rays; the device counterpart of Scene::castRay
 * (common/Scene.hpp:274-285: up to max_depth bounces of Scene::intersect + Scene::TracePath) for a batch.
 *   o, d:        float[3*n] origins / directions of the eye rays (d as Ray::direction: normalised by the caller)
 *   pixel_key,
 *   sample_key:  uint32[n]  the random stream ray r draws from: that of the camera path of pixel pixel_key[r],
 *                sample sample_key[r] (with the context's seed).  Draws are addressed by (stream, bounce, slot), so a ray
 *                equal to that path's camera ray returns exactly that path's radiance, and two rays with one key
 *                are correlated: give every ray of a batch its own key.
 *   rgb:         float[3*n] castRay's return value: radiance, not normalised, not clamped (NaN/inf pass through)
 * The context's integrator flags, seed and traversal apply as in hpt_render; the accumulators are not touched;
 * closest/shadow queries and paths are added to the counters of hpt_stats_get.
 * Host-pointer form (copies in and out, synchronous) and device-pointer form (asynchronous on `stream`,
 * block in [1, 256]). */
int hpt_cast_rays(hpt_ctx* ctx, size_t n, const float* o, const float* d, const uint32_t* pixel_key,
                  const uint32_t* sample_key, int max_depth, float* rgb);
int hpt_cast_rays_dev(hpt_ctx* ctx, size_t n, const float* o_dev, const float* d_dev, const uint32_t* pixel_key_dev,
                      const uint32_t* sample_key_dev, int max_depth, float* rgb_dev, int block, void* stream);

/* Ray counters, kernel time and launch count accumulated since the last
 * hpt_stats_reset.  Synchronises the device. */
int hpt_stats_get(hpt_ctx* ctx, hpt_stats* out);
int hpt_stats_reset(hpt_ctx* ctx);

/* the context's own accumulation buffer (device pointer), for callers that
 * reduce it across GPUs themselves */
int hpt_accum_buffer(hpt_ctx* ctx, int64_t** accum_dev, size_t* num_elements);

/* cudaDeviceSynchronize of CudaRenderer::Render (gpu_path_tracer/CudaRenderer.cu:63): waits for
 * everything queued on the context's device.  Call THIS rather than another library's
 * synchronize: a process can hold more than one HIP runtime (PyTorch wheels bundle their own). */
int hpt_synchronize(hpt_ctx* ctx);

/* The exact accumulators (int64[3*width*height], 40.24 fixed point) copied to host memory after
 * waiting for the device; accum_dev = NULL reads the context's own buffer.  No counterpart in the
 * reference (its framebuffer is float); serves tests and checkpointing a progressive render. */
int hpt_accum_download(hpt_ctx* ctx, const int64_t* accum_dev, int64_t* accum_host);

/* Multi-GPU accumulation (no counterpart in the reference, which renders on one device,
 * gpu_path_tracer/CudaRenderer.hpp:18): sums the contexts' own accumulation buffers into the
 * buffer of ctxs[root].  Contexts on the same device are added there; distinct devices are summed
 * with ONE RCCL reduce (ncclInt64, ncclSum; RCCL is loaded on first use).  Integer sums are
 * associative, so the image is bit-identical to a single-GPU render of the same samples.  All
 * contexts must hold scenes of one resolution.  Waits for the contexts' queued renders and for the
 * reduce.  The other contexts' buffers are left in an unspecified state (clear them before reuse).
 * One process drives all devices (one host thread per device for the renders); processes that own
 * one GPU each reduce hpt_accum_buffer() with their own communicator instead (torch.distributed / RCCL:
 * cudapathtracer_amd/sharding.py). */
int hpt_accum_reduce(hpt_ctx* const* ctxs, int n, int root);

/* The RCCL communicators hpt_accum_reduce uses are kept: one set per ordered list of distinct devices,
 * created by the first reduce over that list and reused by every later one, destroyed together with the
 * process's last context.  hpt_accum_reduce_prepare creates the set for these contexts ahead of the first
 * reduce (HipRenderer::PrepareRender calls it, so that Render's clock holds a reduce, not a communicator
 * bootstrap); it does nothing when all contexts share one device.  hpt_accum_reduce_info reports how many
 * sets the process has created so far and the host time of the last reduce (ms: same-device adds, the
 * collective and the waits after it).  No counterpart in the reference (one GPU,
 * gpu_path_tracer/CudaRenderer.hpp:18). */
int hpt_accum_reduce_prepare(hpt_ctx* const* ctxs, int n);
int hpt_accum_reduce_info(uint64_t* communicators_created, double* last_reduce_ms);

/* Host-only (no GPU, no context): the nine per-octant node-array copies that hpt_scene_upload
 * derives for scenes of up to 64 triangles, for inspection and for tests of the claim they rest
 * on (a walk over copy k reaches exactly the leaves whose boxes the ray hits, minus triangles
 * that face away from every ray of octant k).
 *   leaf_bits != 0: the <= 32-triangle form (leaf word = mask of the leaf's triangles),
 *             == 0: the <= 64-triangle form (leaf word = triangle index, -1 for interior nodes)
 *   nodes_out: float[9][num_nodes][8]  near.xyz, bits(byte offset of the miss target from the
 *              start of the copies) | far.xyz, bits(leaf word); copy 8 = lo/hi, whole tree
 *   table_out: uint32[32]  [k] = records in copy k; [16+2k], [17+2k] = mask of the triangles
 *              whose back-face test octant k does not decide
 * No counterpart in the reference. */
int hpt_octant_copies(const hpt_scene_desc* scene, int leaf_bits, float* nodes_out, uint32_t* table_out);

#ifdef __cplusplus
}
#endif
#endif /* HPT_H */

// Scene: the host-side scene description with the public members and methods
// of the reference's Scene that callers use (common/Scene.hpp:20-93):
// resolution, camera, maxDepth, Add(), buildBVH(), CreateBuiltinScene(), the
// light totals, intersect(), castRay().  TracePath is not a host function here: the
// bounce loop runs on the device, and so do the queries of intersect() and castRay(); buildBVH()
// additionally emits the flattened records that hpt_scene_upload() consumes.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/hpt.h"
#include "BVH.hpp"
#include "Ray.hpp"
#include "Triangle.hpp"

// 64-byte aligned storage for the flattened arrays
struct AlignedFloats {
    std::vector<float> storage;
    float* data = nullptr;
    size_t size = 0;
    void resize(size_t n);
};

// The flattened scene: one pre-order node array for the whole two-level BVH,
// triangles in leaf order.  Record formats are documented at hpt_scene_desc.
struct FlatScene {
    AlignedFloats nodes, node_area, tris, tri_v12;
    AlignedFloats ref_nodes;              // sah_build: the reference's own tree beside `nodes` (hpt_scene_desc::ref_nodes)
    uint32_t num_ref_nodes = 0;
    std::vector<hpt_material> materials;  // one per mesh, Scene::Add order
    std::vector<hpt_light> lights;        // emitting meshes, Scene::Add order
    std::vector<int32_t> prim_of_tri;     // leaf order -> global primitive id
    std::vector<int32_t> mesh_root;       // mesh index -> node index of its BVH root
    uint32_t num_nodes = 0, num_tris = 0;
    bool gpu_build = false;               // Scene::BvhBuild::Gpu: triangles in primitive-id order, nodes = the light meshes' trees only
    bool sah_build = false;               // Scene::BvhBuild::Sah: non-emitting meshes and the top level split by the surface-area heuristic
};

class Scene {
public:
    // setting up options
    std::string name = "default";
    Vec3 camPos = Vec3(0, 0, 0);
    int width = 1280;
    int height = 960;
    double fov = 40;
    Vec3 backgroundColor = Vec3(0.235294f, 0.67451f, 0.843137f);  // unused by the integrator, as in the reference
    int maxDepth = 1;
    float RussianRoulette = 0.8f;  // unused (enableRR = false, common/Scene.hpp:142)

    BVHAccel* bvh = nullptr;              // top level, over meshes: the reference's tree, whatever tree the flattened records hold
    std::vector<MeshTriangle*> meshes;    // Scene::Add order
    std::vector<BVHAccel*> mesh_bvhs;     // per mesh, same order: the reference's trees
    int num_meshes = 0;
    float sumLightArea = 0.0f;
    float sumLightPower = 0.0f;
    int num_lights = 0;

    Scene(int w, int h) : width(w), height(h) {}

    // takes shared ownership of the mesh
    void Add(MeshTriangle* mesh);
    // a sphere (no counterpart in the reference, whose Scene::Add takes MeshTriangle* only); must not emit
    void Add(Sphere* sphere);
    std::vector<Sphere*> spheres;
    // meshes and spheres in Add order: {false, index into meshes} / {true, index into spheres}.  Material
    // indices and primitive ids of the flattened scene count through this list.
    struct ObjectRef {
        bool sphere;
        int index;
    };
    std::vector<ObjectRef> objects;
    // Reference: the two-level BVH by the reference's rules on the host (default; every result the reference's).
    // Gpu: only the trees of the emitting meshes are built here (Scene::sampleLight descends them); the traversal
    // records for the whole scene are built on the device at upload (HPT_BVH_BUILD_GPU in include/hpt.h says what
    // that guarantees).  `bvh` and the mesh_bvhs of non-emitting meshes stay null then.
    // Sah: the flattened records hold ANOTHER tree over the same leaves: the top level and every mesh that does not emit
    // are split by the surface-area heuristic (BVHAccel::SplitMethod::SAH, which the reference declares without
    // implementing, common/BVH.hpp:28) and their children ordered so that the traversal stack stays short; emitting meshes
    // keep the reference's tree, which Scene::sampleLight descends.  The leaves (one triangle each, same boxes) are the
    // reference's, so every ray with finite reciprocal direction components tests the same triangles and gets the
    // same closest distance and occlusion answer; the triangle records stay in the reference's traversal order,
    // so of several triangles hit at bitwise equal distance the one the reference names is named; and the reference's
    // own tree is emitted beside the other (FlatScene::ref_nodes) for the rays that argument leaves out -- those with a
    // zero direction component, whose slab tests can fail on 0 * inf = NaN -- which walk it on the device.  So every
    // result is the reference's, with fewer boxes tested on the way: 11-16 % more rays per second on the bunny and the
    // 1.0 M-triangle height field.  `bvh` and `mesh_bvhs` stay the reference's trees.
    // Auto (default): Sah for scenes of more than 64 primitives, else Reference.
    enum class BvhBuild { Reference, Gpu, Sah, Auto };
    static BvhBuild default_bvh_build;  // what new scenes start with (the built-in scenes build inside their factory)
    BvhBuild bvh_build = default_bvh_build;
    void buildBVH();

    enum BuiltinScene {
        CornellBox,
        // Cornell walls + light with the Stanford bunny (models/bunny/bunny.obj, 4968 triangles,
        // scaled by 2000 and standing on the floor) in place of the two boxes.  The reference
        // ships the mesh but no scene that uses it; the same scene is defined, vertex for
        // vertex, in cudapathtracer_amd/scenes.py (bunny_in_cornell).
        BunnyInCornell,
        // The Cornell box with the short box made of the glass that common/Scene.cpp:45-47
        // parametrises (roughness 0.2, ior 1.5) but leaves Lambertian: Material::Type::Glass.
        GlassCornell,
        // Cornell walls + light over an n x n displaced height field (2 n^2 triangles; n = 708 gives
        // 1.0 M): the BVH-fetch stress scene.  Created with CreateHeightfieldScene(n, maxDepth); the same
        // scene is defined, vertex for vertex, in cudapathtracer_amd/scenes.py (heightfield_in_cornell).
        HeightfieldInCornell,
    };
    // OBJ files are opened as "models/<dir>/<name>.obj" relative to the
    // current directory, or under $HPT_MODELS_DIR when that is set.
    static Scene CreateBuiltinScene(BuiltinScene sceneId, int maxDepth);
    static Scene CreateBunnyScene(int maxDepth);
    static Scene CreateHeightfieldScene(int n, int maxDepth);
    static std::vector<Vec3> HeightfieldVertices(int n);  // 3 vertices per triangle, 2 n^2 triangles
    // A scene description file (the "self-defined material description file" of the reference's
    // roadmap, README.md:62; format: host/SceneFile.cpp).  Paths in it are relative to the file.
    static Scene LoadSceneFile(const std::string& path, int maxDepth);
    // "cornell" | "bunny" | "glass" | "heightfield:<n>" | "file:<path>" (what HPT_SCENE accepts)
    static Scene CreateNamedScene(const std::string& name, int maxDepth);

    // tan(fov/2) exactly as the host renderer of the reference computes it
    // (common/Renderer.cpp:11,20)
    float cameraScale() const;

    // Scene::intersect (common/Scene.hpp:95-97 -> BVHAccel::Intersect) for host callers: closest hit, or the
    // any-hit answer for a ray with shadowTest set.  The queries run on GPU `query_device` through
    // hpt_intersect_batch (the scene is uploaded to a context of its own on first use); pass many rays at once
    // where possible -- a single ray costs a kernel launch.  Throws std::runtime_error without a GPU.
    Intersection intersect(const Ray& ray) const;
    std::vector<Intersection> intersect(const std::vector<Ray>& rays) const;
    int query_device = 0;

    // Scene::castRay (common/Scene.hpp:274-285) for host callers: the radiance along an eye ray after up to
    // maxDepth bounces of intersect + TracePath, computed on the GPU (hpt_cast_rays, same context as intersect()).
    // The reference hands castRay an RNG object; the device draws from a keyed stream instead, so the caller names the
    // stream: that of the camera path (pixel, sample) under seed `query_seed`.  The eye ray of that path returns that
    // path's radiance; rays that share a key are correlated.  `query_integrator`: HPT_INTEGRATOR_* flags.
    struct PathKey {
        uint32_t pixel, sample;
    };
    Vec3 castRay(PathKey key, const Ray& eyeRay) const;
    std::vector<Vec3> castRay(const std::vector<PathKey>& keys, const std::vector<Ray>& eyeRays) const;
    uint32_t query_seed = 0, query_integrator = 0;

    // valid after buildBVH(); pointers inside stay valid while the scene lives
    const FlatScene& flat() const;
    hpt_scene_desc describe() const;

private:
    std::vector<std::shared_ptr<MeshTriangle>> mesh_owner_;
    std::vector<std::shared_ptr<Sphere>> sphere_owner_;
    std::shared_ptr<BVHAccel> bvh_owner_;
    std::vector<std::shared_ptr<BVHAccel>> mesh_bvh_owner_;
    std::shared_ptr<FlatScene> flat_;
    void buildForGpu();
    void fillMaterialsAndLights(FlatScene& fs) const;
    hpt_ctx* queryContext() const;
    mutable std::shared_ptr<hpt_ctx> query_ctx_;   // intersect() / castRay(): created and uploaded on first use
    mutable int query_ctx_device_ = -1;            // the GPU query_ctx_ lives on (recreated when query_device changes)
    mutable std::vector<int32_t> tri_of_prim_;     // global primitive id -> triangle record
};

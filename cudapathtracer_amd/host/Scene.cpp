#include "Scene.hpp"

#include <random>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

void AlignedFloats::resize(size_t n) {
    storage.assign(n + 16, 0.f);
    uintptr_t p = reinterpret_cast<uintptr_t>(storage.data());
    p = (p + 63u) & ~uintptr_t(63);
    data = reinterpret_cast<float*>(p);
    size = n;
}

namespace {
float luminance(Vec3 c) { return dot(c, Vec3(.2126f, .7152f, .0722f)); }
void put_int(float* dst, int32_t v) { std::memcpy(dst, &v, 4); }
}  // namespace

Scene::BvhBuild Scene::default_bvh_build = Scene::BvhBuild::Auto;

// light bookkeeping as common/Scene.hpp:47-56
void Scene::Add(Sphere* sphere) {
    sphere_owner_.emplace_back(sphere);
    if (sphere->material.emitting()) throw std::runtime_error("Scene::Add: a sphere cannot be a light (Scene::sampleLight samples meshes only)");
    if (!(sphere->radius > 0.f)) throw std::runtime_error("Scene::Add: a sphere needs a positive radius");
    objects.push_back(ObjectRef{true, (int)spheres.size()});
    spheres.push_back(sphere);
}

void Scene::Add(MeshTriangle* mesh) {
    mesh_owner_.emplace_back(mesh);
    objects.push_back(ObjectRef{false, (int)meshes.size()});
    meshes.push_back(mesh);
    num_meshes = (int)meshes.size();
    if (mesh->material.emitting()) {
        sumLightArea += mesh->area;
        sumLightPower += mesh->area * luminance(mesh->material._emission) * 2.f * 3.14159265358979323846264338327950288f;
        ++num_lights;
    }
}

float Scene::cameraScale() const {
    const float half_deg = (float)(fov * 0.5);
    const float half_rad = (float)(half_deg * 3.1415926535897932384626422832795028841971f / 180.0);
    return (float)std::tan((double)half_rad);
}

// Builds the two-level BVH (top level over the objects -- meshes and spheres in Add order --, one
// nested BVH per mesh, common/Scene.cpp:9-23 + common/BVH.cpp:92-104) and flattens it: every
// top-level mesh leaf is replaced by its mesh's subtree, whose exits continue at the node that
// follows the leaf; a sphere is a leaf of the top level itself.
void Scene::buildBVH() {
    if (objects.empty()) throw std::runtime_error("Scene::buildBVH: scene has no meshes");
    query_ctx_.reset();  // intersect() uploads the new scene on its next call
    tri_of_prim_.clear();
    if (bvh_build == BvhBuild::Gpu) {
        buildForGpu();
        return;
    }
    // Reference: the top level over the objects, every mesh a subtree of its own (common/Scene.cpp:9-23).
    // Sah: surface-area splits for the top level and for every mesh that does not emit (an emitting mesh keeps the
    // reference's subtree: Scene::sampleLight descends it).  Small non-emitting meshes (walls, boxes: up to
    // kExplode triangles) enter the top level as individual triangles instead of as one object each: their
    // room-sized boxes would otherwise all be entered by every ray.
    // Auto: Sah for scenes of more than 64 primitives (the ones whose records are read through the caches), else
    // Reference (the device keeps such scenes in LDS under trees of its own making).
    size_t num_prims = 0;
    for (const ObjectRef& o : objects) num_prims += o.sphere ? 1u : meshes[o.index]->triangles.size();
    const bool sah = bvh_build == BvhBuild::Sah || (bvh_build == BvhBuild::Auto && num_prims > 64);
    constexpr size_t kExplode = 64;
    struct TopItem {
        int obj;  // index into `objects`
        int tri;  // >= 0: that triangle of the mesh; -1: the whole object (a sphere, or a mesh spliced in as a subtree)
    };
    std::vector<TopItem> items;
    std::vector<Object*> top_objects, ref_objects;
    for (size_t k = 0; k < objects.size(); ++k) {
        const ObjectRef& o = objects[k];
        Object* whole = o.sphere ? (Object*)spheres[o.index] : (Object*)meshes[o.index];
        ref_objects.push_back(whole);
        if (sah && !o.sphere && !meshes[o.index]->material.emitting() && meshes[o.index]->triangles.size() <= kExplode) {
            MeshTriangle* m = meshes[o.index];
            for (size_t i = 0; i < m->triangles.size(); ++i) {
                items.push_back({(int)k, (int)i});
                top_objects.push_back(&m->triangles[i]);
            }
        } else {
            items.push_back({(int)k, -1});
            top_objects.push_back(whole);
        }
    }
    // `bvh` and `mesh_bvhs` -- the members the reference's Scene shows its callers -- are the reference's trees
    // whatever is emitted; the surface-area trees only live until the records are written.
    mesh_bvh_owner_.clear();
    mesh_bvhs.clear();
    std::vector<std::shared_ptr<BVHAccel>> emit_mesh;  // the tree each mesh is emitted with
    for (MeshTriangle* m : meshes) {
        std::vector<Object*> tris(m->triangles.size());
        for (size_t i = 0; i < tris.size(); ++i) tris[i] = &m->triangles[i];
        mesh_bvh_owner_.push_back(std::make_shared<BVHAccel>(tris, 1, BVHAccel::SplitMethod::NAIVE));
        mesh_bvhs.push_back(mesh_bvh_owner_.back().get());
        const bool exploded = sah && !m->material.emitting() && m->triangles.size() <= kExplode;  // no subtree of its own is emitted
        emit_mesh.push_back(sah && !m->material.emitting() && !exploded ? std::make_shared<BVHAccel>(tris, 1, BVHAccel::SplitMethod::SAH) : mesh_bvh_owner_.back());
    }
    bvh_owner_ = std::make_shared<BVHAccel>(ref_objects, 1, BVHAccel::SplitMethod::NAIVE);
    bvh = bvh_owner_.get();
    // the emitted top level orders its children by the stack their subtrees need, the meshes' subtrees included
    std::vector<int32_t> leaf_need(items.size(), -1);
    for (size_t i = 0; i < items.size(); ++i) {
        const ObjectRef& o = objects[(size_t)items[i].obj];
        if (items[i].tri < 0 && !o.sphere && meshes[o.index]->triangles.size() > 1) leaf_need[i] = emit_mesh[o.index]->stackNeed();
    }
    const std::shared_ptr<BVHAccel> emit_top = sah ? std::make_shared<BVHAccel>(top_objects, 1, BVHAccel::SplitMethod::SAH, &leaf_need) : bvh_owner_;
    // primitive ids count through the objects in Add order (a sphere is one primitive)
    std::vector<int32_t> object_base(objects.size());
    int32_t base = 0;
    for (size_t k = 0; k < objects.size(); ++k) {
        object_base[k] = base;
        base += objects[k].sphere ? 1 : meshes[objects[k].index]->num_triangles;
    }

    // global index of every top-level node once the mesh subtrees are spliced in
    const std::vector<FlatNode>& top = emit_top->nodes;
    std::vector<int32_t> top_to_global(top.size() + 1);
    int32_t cursor = 0;
    for (size_t t = 0; t < top.size(); ++t) {
        top_to_global[t] = cursor;
        const bool mesh_leaf = top[t].object >= 0 && items[top[t].object].tri < 0 && !objects[items[top[t].object].obj].sphere;
        cursor += mesh_leaf ? (int32_t)emit_mesh[objects[items[top[t].object].obj].index]->nodes.size() : 1;
    }
    top_to_global[top.size()] = cursor;
    const uint32_t num_nodes = (uint32_t)cursor, num_tris = (uint32_t)base;

    auto fs = std::make_shared<FlatScene>();
    fs->num_nodes = num_nodes;
    fs->num_tris = num_tris;
    fs->nodes.resize(8ull * num_nodes);
    fs->node_area.resize(num_nodes);
    fs->tris.resize(16ull * num_tris);
    fs->tri_v12.resize(8ull * num_tris);
    fs->prim_of_tri.resize(num_tris);
    fs->mesh_root.assign(meshes.size(), -1);

    // A triangle's record index is its rank in the REFERENCE's traversal order (top-level leaves in the order the
    // reference's top-level tree meets them, then the mesh's leaves in the order its reference tree meets them),
    // whatever tree is emitted: among triangles hit at bitwise the same distance the kernels keep the lowest record
    // index, which is then the triangle the reference's strict `<` keeps (common/BVH.hpp:128-137).
    std::vector<int32_t> rec_base(objects.size(), 0);
    std::vector<std::vector<int32_t>> rec_of_tri(meshes.size());
    {
        int32_t next = 0;
        for (int32_t obj : bvh_owner_->leafOrder()) {
            rec_base[obj] = next;
            next += objects[obj].sphere ? 1 : meshes[objects[obj].index]->num_triangles;
        }
        for (size_t k = 0; k < meshes.size(); ++k) {
            const std::vector<int32_t> order = mesh_bvh_owner_[k]->leafOrder();
            rec_of_tri[k].assign(order.size(), 0);
            for (size_t i = 0; i < order.size(); ++i) rec_of_tri[k][(size_t)order[i]] = (int32_t)i;
        }
    }

    auto write_node = [&](int32_t g, const Bounds3& b, int32_t miss, int32_t tri, float area) {
        float* n = fs->nodes.data + 8ull * g;
        n[0] = b.pMin.x; n[1] = b.pMin.y; n[2] = b.pMin.z; put_int(n + 3, miss);
        n[4] = b.pMax.x; n[5] = b.pMax.y; n[6] = b.pMax.z; put_int(n + 7, tri);
        fs->node_area.data[g] = area;
    };

    for (size_t t = 0; t < top.size(); ++t) {
        const FlatNode& tn = top[t];
        const int32_t g0 = top_to_global[t];
        if (tn.object < 0) {
            write_node(g0, tn.bounds, top_to_global[tn.miss], -1, tn.area);
            continue;
        }
        const int obj = items[tn.object].obj;  // index into `objects` = material index
        auto write_triangle = [&](int32_t rec, const Triangle& T, int32_t prim) {
            float* r = fs->tris.data + 16ull * rec;
            r[0] = T.v0.x; r[1] = T.v0.y; r[2] = T.v0.z; r[3] = T.area;
            r[4] = T.e1.x; r[5] = T.e1.y; r[6] = T.e1.z; put_int(r + 7, obj);
            r[8] = T.e2.x; r[9] = T.e2.y; r[10] = T.e2.z; put_int(r + 11, prim);
            r[12] = T.normal.x; r[13] = T.normal.y; r[14] = T.normal.z; put_int(r + 15, obj);
            float* w = fs->tri_v12.data + 8ull * rec;
            w[0] = T.v1.x; w[1] = T.v1.y; w[2] = T.v1.z; w[3] = 0.f;
            w[4] = T.v2.x; w[5] = T.v2.y; w[6] = T.v2.z; w[7] = 0.f;
            fs->prim_of_tri[rec] = prim;
        };
        if (objects[obj].sphere) {
            const Sphere* sp = spheres[objects[obj].index];
            const int32_t rec = rec_base[obj];
            float* r = fs->tris.data + 16ull * rec;
            r[0] = sp->center.x; r[1] = sp->center.y; r[2] = sp->center.z; r[3] = 0.f;
            r[4] = sp->radius; r[5] = sp->radius2; r[6] = 0.f; put_int(r + 7, obj);
            r[8] = r[9] = r[10] = 0.f; put_int(r + 11, object_base[obj]);
            r[12] = r[13] = r[14] = 0.f; put_int(r + 15, (int32_t)((uint32_t)obj | 0x80000000u));
            float* w = fs->tri_v12.data + 8ull * rec;
            for (int c = 0; c < 8; ++c) w[c] = 0.f;
            fs->prim_of_tri[rec] = object_base[obj];
            write_node(g0, tn.bounds, top_to_global[tn.miss], rec, tn.area);
            continue;
        }
        const int k = objects[obj].index;
        const MeshTriangle* mesh = meshes[k];
        if (items[tn.object].tri >= 0) {  // Sah: a triangle of a small mesh is a leaf of the top level
            const int ti = items[tn.object].tri;
            const int32_t rec = rec_base[obj] + rec_of_tri[k][(size_t)ti];
            write_triangle(rec, mesh->triangles[ti], object_base[obj] + ti);
            write_node(g0, tn.bounds, top_to_global[tn.miss], rec, tn.area);
            continue;
        }
        const std::vector<FlatNode>& mn = emit_mesh[k]->nodes;
        // The reference tests the mesh's bounding box (top-level leaf) and then
        // the identical box of the nested root; one test gives the same answer.
        const Bounds3 rb = mn[0].bounds, mb = mesh->bounding_box;
        if (std::memcmp(&rb.pMin, &mb.pMin, sizeof(Vec3)) || std::memcmp(&rb.pMax, &mb.pMax, sizeof(Vec3)))
            throw std::runtime_error("Scene::buildBVH: mesh bounding box differs from its BVH root (NaN or signed-zero vertex?)");
        fs->mesh_root[k] = g0;
        for (size_t i = 0; i < mn.size(); ++i) {
            const FlatNode& n = mn[i];
            int32_t tri = -1;
            if (n.object >= 0) {
                tri = rec_base[obj] + rec_of_tri[k][(size_t)n.object];
                write_triangle(tri, mesh->triangles[n.object], object_base[obj] + n.object);
            }
            write_node(g0 + (int32_t)i, n.bounds, g0 + n.miss, tri, n.area);
        }
    }

    if (sah) {
        // The reference's own two-level tree, flattened the same way, beside the emitted one: rays with a zero
        // direction component walk it on the device (hpt_scene_desc::ref_nodes says why), so every ray gets the
        // reference's answer although `nodes` holds another tree.
        const std::vector<FlatNode>& rt = bvh_owner_->nodes;
        std::vector<int32_t> to_global(rt.size() + 1);
        int32_t at = 0;
        for (size_t t = 0; t < rt.size(); ++t) {
            to_global[t] = at;
            const bool mesh_leaf = rt[t].object >= 0 && !objects[(size_t)rt[t].object].sphere;
            at += mesh_leaf ? (int32_t)mesh_bvhs[objects[(size_t)rt[t].object].index]->nodes.size() : 1;
        }
        to_global[rt.size()] = at;
        fs->num_ref_nodes = (uint32_t)at;
        fs->ref_nodes.resize(8ull * fs->num_ref_nodes);
        auto write_ref = [&](int32_t g, const Bounds3& b, int32_t miss, int32_t tri) {
            float* n = fs->ref_nodes.data + 8ull * g;
            n[0] = b.pMin.x; n[1] = b.pMin.y; n[2] = b.pMin.z; put_int(n + 3, miss);
            n[4] = b.pMax.x; n[5] = b.pMax.y; n[6] = b.pMax.z; put_int(n + 7, tri);
        };
        for (size_t t = 0; t < rt.size(); ++t) {
            const FlatNode& tn = rt[t];
            const int32_t g0 = to_global[t];
            if (tn.object < 0) {
                write_ref(g0, tn.bounds, to_global[tn.miss], -1);
            } else if (objects[(size_t)tn.object].sphere) {
                write_ref(g0, tn.bounds, to_global[tn.miss], rec_base[tn.object]);
            } else {
                const int k = objects[(size_t)tn.object].index;
                const std::vector<FlatNode>& mn = mesh_bvhs[k]->nodes;
                for (size_t i = 0; i < mn.size(); ++i)
                    write_ref(g0 + (int32_t)i, mn[i].bounds, g0 + mn[i].miss, mn[i].object >= 0 ? rec_base[tn.object] + rec_of_tri[k][(size_t)mn[i].object] : -1);
            }
        }
    }
    fs->sah_build = sah;
    fillMaterialsAndLights(*fs);
    flat_ = fs;
}

// the material table (one entry per object, Add order) and the light table (emitting meshes, Add order)
void Scene::fillMaterialsAndLights(FlatScene& fs) const {
    for (size_t obj = 0; obj < objects.size(); ++obj) {
        const Material& m = objects[obj].sphere ? spheres[objects[obj].index]->material : meshes[objects[obj].index]->material;
        hpt_material hm;
        std::memset(&hm, 0, sizeof hm);
        hm.type = m.type;
        hm.emission[0] = m._emission.x; hm.emission[1] = m._emission.y; hm.emission[2] = m._emission.z;
        hm.albedo[0] = m._albedo.x; hm.albedo[1] = m._albedo.y; hm.albedo[2] = m._albedo.z;
        hm.metallic = m._metallic;
        hm.roughness = m._roughness;
        hm.ior = m._ior;
        fs.materials.push_back(hm);
        if (!objects[obj].sphere && m.emitting()) {
            const int k = objects[obj].index;
            hpt_light L;
            L.area = meshes[k]->area;
            L.root_area = mesh_bvhs[k]->rootArea();
            L.root_node = (uint32_t)fs.mesh_root[k];
            L.material = (uint32_t)obj;
            fs.lights.push_back(L);
        }
    }
}

// BvhBuild::Gpu: triangle records in primitive-id order (objects in Add order, triangles in file order), the
// node array = the BVHs of the emitting meshes one after the other (what Scene::sampleLight descends,
// common/BVH.hpp:147-184), leaf words = indices into the triangle array.
void Scene::buildForGpu() {
    bvh_owner_.reset();
    bvh = nullptr;
    mesh_bvh_owner_.assign(meshes.size(), nullptr);
    mesh_bvhs.assign(meshes.size(), nullptr);
    std::vector<int32_t> object_base(objects.size());
    int32_t base = 0;
    uint32_t num_nodes = 0;
    for (size_t k = 0; k < objects.size(); ++k) {
        object_base[k] = base;
        if (objects[k].sphere) {
            base += 1;
            continue;
        }
        const int mi = objects[k].index;
        MeshTriangle* m = meshes[mi];
        base += m->num_triangles;
        if (m->material.emitting()) {
            std::vector<Object*> tris(m->triangles.size());
            for (size_t i = 0; i < tris.size(); ++i) tris[i] = &m->triangles[i];
            mesh_bvh_owner_[mi] = std::make_shared<BVHAccel>(tris, 1, BVHAccel::SplitMethod::NAIVE);
            mesh_bvhs[mi] = mesh_bvh_owner_[mi].get();
            num_nodes += (uint32_t)mesh_bvhs[mi]->nodes.size();
        }
    }
    const bool no_lights = num_nodes == 0;
    if (no_lights) num_nodes = 1;  // the array may not be empty; nothing reads the entry
    auto fs = std::make_shared<FlatScene>();
    fs->gpu_build = true;
    fs->num_nodes = num_nodes;
    fs->num_tris = (uint32_t)base;
    fs->nodes.resize(8ull * num_nodes);
    fs->node_area.resize(num_nodes);
    fs->tris.resize(16ull * fs->num_tris);
    fs->tri_v12.resize(8ull * fs->num_tris);
    fs->prim_of_tri.resize(fs->num_tris);
    fs->mesh_root.assign(meshes.size(), -1);
    if (no_lights) {
        float* n = fs->nodes.data;
        for (int c = 0; c < 8; ++c) n[c] = 0.f;
        put_int(n + 3, 1);
        put_int(n + 7, 0);
        fs->node_area.data[0] = 0.f;
    }
    int32_t g0 = 0;
    for (size_t obj = 0; obj < objects.size(); ++obj) {
        const int32_t first = object_base[obj];
        if (objects[obj].sphere) {
            const Sphere* sp = spheres[objects[obj].index];
            float* r = fs->tris.data + 16ull * first;
            r[0] = sp->center.x; r[1] = sp->center.y; r[2] = sp->center.z; r[3] = 0.f;
            r[4] = sp->radius; r[5] = sp->radius2; r[6] = 0.f; put_int(r + 7, (int32_t)obj);
            r[8] = r[9] = r[10] = 0.f; put_int(r + 11, first);
            r[12] = r[13] = r[14] = 0.f; put_int(r + 15, (int32_t)((uint32_t)obj | 0x80000000u));
            float* w = fs->tri_v12.data + 8ull * first;
            for (int c = 0; c < 8; ++c) w[c] = 0.f;
            fs->prim_of_tri[first] = first;
            continue;
        }
        const int mi = objects[obj].index;
        const MeshTriangle* mesh = meshes[mi];
        for (int i = 0; i < mesh->num_triangles; ++i) {
            const Triangle& T = mesh->triangles[i];
            const int32_t tri = first + i;
            float* r = fs->tris.data + 16ull * tri;
            r[0] = T.v0.x; r[1] = T.v0.y; r[2] = T.v0.z; r[3] = T.area;
            r[4] = T.e1.x; r[5] = T.e1.y; r[6] = T.e1.z; put_int(r + 7, (int32_t)obj);
            r[8] = T.e2.x; r[9] = T.e2.y; r[10] = T.e2.z; put_int(r + 11, tri);
            r[12] = T.normal.x; r[13] = T.normal.y; r[14] = T.normal.z; put_int(r + 15, (int32_t)obj);
            float* w = fs->tri_v12.data + 8ull * tri;
            w[0] = T.v1.x; w[1] = T.v1.y; w[2] = T.v1.z; w[3] = 0.f;
            w[4] = T.v2.x; w[5] = T.v2.y; w[6] = T.v2.z; w[7] = 0.f;
            fs->prim_of_tri[tri] = tri;
        }
        if (!mesh_bvhs[mi]) continue;
        const std::vector<FlatNode>& mn = mesh_bvhs[mi]->nodes;
        fs->mesh_root[mi] = g0;
        for (size_t i = 0; i < mn.size(); ++i) {
            float* n = fs->nodes.data + 8ull * (g0 + (int32_t)i);
            const Bounds3& b = mn[i].bounds;
            n[0] = b.pMin.x; n[1] = b.pMin.y; n[2] = b.pMin.z; put_int(n + 3, g0 + mn[i].miss);
            n[4] = b.pMax.x; n[5] = b.pMax.y; n[6] = b.pMax.z; put_int(n + 7, mn[i].object >= 0 ? first + mn[i].object : -1);
            fs->node_area.data[g0 + (int32_t)i] = mn[i].area;
        }
        g0 += (int32_t)mn.size();
    }
    fillMaterialsAndLights(*fs);
    flat_ = fs;
}

const FlatScene& Scene::flat() const {
    if (!flat_) throw std::runtime_error("Scene: buildBVH() has not been called");
    return *flat_;
}

hpt_ctx* Scene::queryContext() const {
    const FlatScene& f = flat();
    if (query_ctx_ && query_ctx_device_ != query_device) query_ctx_.reset();  // the caller moved the queries to another GPU: upload there
    if (!query_ctx_) {
        hpt_ctx* ctx = nullptr;
        if (hpt_context_create(query_device, &ctx) != HPT_OK) throw std::runtime_error(std::string("Scene: ") + hpt_last_error());
        query_ctx_ = std::shared_ptr<hpt_ctx>(ctx, [](hpt_ctx* c) { hpt_context_destroy(c); });
        query_ctx_device_ = query_device;
        const hpt_scene_desc d = describe();
        if (hpt_set_bvh_build(ctx, f.gpu_build ? HPT_BVH_BUILD_GPU : HPT_BVH_BUILD_HOST) != HPT_OK || hpt_scene_upload(ctx, &d) != HPT_OK) {
            query_ctx_.reset();
            throw std::runtime_error(std::string("Scene: ") + hpt_last_error());
        }
        int32_t prims = 0;
        for (int32_t p : f.prim_of_tri) prims = std::max(prims, p + 1);
        tri_of_prim_.assign((size_t)prims, -1);
        for (size_t t = 0; t < f.prim_of_tri.size(); ++t) tri_of_prim_[(size_t)f.prim_of_tri[t]] = (int32_t)t;
    }
    return query_ctx_.get();
}

std::vector<Vec3> Scene::castRay(const std::vector<PathKey>& keys, const std::vector<Ray>& eyeRays) const {
    if (keys.size() != eyeRays.size()) throw std::invalid_argument("Scene::castRay: one key per eye ray");
    hpt_ctx* ctx = queryContext();
    const size_t n = eyeRays.size();
    std::vector<Vec3> out(n);
    if (n == 0) return out;
    std::vector<float> o(3 * n), d(3 * n), rgb(3 * n);
    std::vector<uint32_t> pk(n), sk(n);
    for (size_t i = 0; i < n; ++i) {
        const Ray& r = eyeRays[i];
        o[3 * i] = r.origin.x; o[3 * i + 1] = r.origin.y; o[3 * i + 2] = r.origin.z;
        d[3 * i] = r.direction.x; d[3 * i + 1] = r.direction.y; d[3 * i + 2] = r.direction.z;
        pk[i] = keys[i].pixel;
        sk[i] = keys[i].sample;
    }
    if (hpt_set_seed(ctx, query_seed) != HPT_OK || hpt_set_integrator(ctx, query_integrator) != HPT_OK ||
        hpt_cast_rays(ctx, n, o.data(), d.data(), pk.data(), sk.data(), maxDepth, rgb.data()) != HPT_OK)
        throw std::runtime_error(std::string("Scene::castRay: ") + hpt_last_error());
    for (size_t i = 0; i < n; ++i) out[i] = Vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return out;
}

Vec3 Scene::castRay(PathKey key, const Ray& eyeRay) const { return castRay(std::vector<PathKey>{key}, std::vector<Ray>{eyeRay})[0]; }

std::vector<Intersection> Scene::intersect(const std::vector<Ray>& rays) const {
    const FlatScene& f = flat();
    hpt_ctx* query = queryContext();
    const size_t n = rays.size();
    std::vector<Intersection> out(n);
    if (n == 0) return out;
    std::vector<float> o(3 * n), d(3 * n);
    std::vector<uint8_t> shadow(n);
    std::vector<double> tmax(n), t(n);
    std::vector<int32_t> prim(n);
    for (size_t i = 0; i < n; ++i) {
        const Ray& r = rays[i];
        o[3 * i] = r.origin.x; o[3 * i + 1] = r.origin.y; o[3 * i + 2] = r.origin.z;
        d[3 * i] = r.direction.x; d[3 * i + 1] = r.direction.y; d[3 * i + 2] = r.direction.z;
        shadow[i] = r.shadowTest ? 1 : 0;
        tmax[i] = r.shadowDistance;
    }
    if (hpt_intersect_batch(query, n, o.data(), d.data(), shadow.data(), tmax.data(), prim.data(), t.data()) != HPT_OK)
        throw std::runtime_error(std::string("Scene::intersect: ") + hpt_last_error());
    for (size_t i = 0; i < n; ++i) {
        if (prim[i] < 0) continue;
        Intersection& in = out[i];
        const float* rec = f.tris.data + 16 * (size_t)tri_of_prim_[(size_t)prim[i]];
        uint32_t mat, tag;
        std::memcpy(&mat, rec + 7, 4);
        std::memcpy(&tag, rec + 15, 4);
        in.happened = true;
        in.primitive = prim[i];
        in.distance = t[i];
        in.coords = rays[i](t[i]);  // Triangle.hpp:191
        const ObjectRef ref = objects[mat];  // one material per object, in Add order
        if (tag & 0x80000000u) {  // a sphere record: center, then radius (Sphere.hpp:62, commented out in the reference)
            in.normal = normalize(in.coords - Vec3(rec[0], rec[1], rec[2]));
            in.m = spheres[ref.index]->material;
        } else {
            in.normal = Vec3(rec[12], rec[13], rec[14]);
            in.triangleArea = rec[3];
            in.m = meshes[ref.index]->material;
        }
        in.emit = in.m._emission;
    }
    return out;
}

Intersection Scene::intersect(const Ray& ray) const { return intersect(std::vector<Ray>{ray})[0]; }

hpt_scene_desc Scene::describe() const {
    const FlatScene& f = flat();
    hpt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.num_nodes = f.num_nodes;
    d.nodes = f.nodes.data;
    d.node_area = f.node_area.data;
    d.num_tris = f.num_tris;
    d.tris = f.tris.data;
    d.tri_v12 = f.tri_v12.data;
    d.num_materials = (uint32_t)f.materials.size();
    d.materials = f.materials.data();
    d.num_lights = (uint32_t)f.lights.size();
    d.lights = f.lights.data();
    d.sum_light_area = sumLightArea;
    d.sum_light_power = sumLightPower;
    d.cam_pos[0] = camPos.x; d.cam_pos[1] = camPos.y; d.cam_pos[2] = camPos.z;
    d.cam_scale = cameraScale();
    d.width = width;
    d.height = height;
    d.num_ref_nodes = f.num_ref_nodes;
    d.ref_nodes = f.num_ref_nodes ? f.ref_nodes.data : nullptr;
    return d;
}

// The built-in Cornell box: resolution, camera, materials, mesh order and
// file names of common/Scene.cpp:25-73.
Scene Scene::CreateBuiltinScene(Scene::BuiltinScene sceneId, int maxDepth) {
    if (sceneId == Scene::BuiltinScene::BunnyInCornell) return CreateBunnyScene(maxDepth);
    if (sceneId == Scene::BuiltinScene::HeightfieldInCornell) return CreateHeightfieldScene(708, maxDepth);
    if (sceneId != Scene::BuiltinScene::CornellBox && sceneId != Scene::BuiltinScene::GlassCornell)
        throw std::runtime_error("Unsupported sceneId");
    Scene scene(512, 512);
    scene.name = sceneId == Scene::BuiltinScene::GlassCornell ? "Glass Cornell" : "Cornell Box";
    scene.maxDepth = maxDepth;
    scene.camPos = Vec3(278, 273, -800);

    Material red, green, white, light, metal, glass;
    red._albedo = Vec3(0.63f, 0.065f, 0.05f);
    green._albedo = Vec3(0.14f, 0.45f, 0.091f);
    white._albedo = Vec3(0.725f, 0.71f, 0.68f);
    light._emission = (8.0f * Vec3(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) + 15.6f * Vec3(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                       18.4f * Vec3(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));
    metal.type = Material::Type::MetallicWorkflow;
    metal._albedo = Vec3((float)0.7, (float)0.5, (float)0.2);
    metal._metallic = 1.f;
    metal._roughness = 0.005f;
    // the "glass" short box is Lambertian in the reference's scene (common/Scene.cpp:45)
    glass.type = sceneId == Scene::BuiltinScene::GlassCornell ? Material::Type::Glass : Material::Type::Lambertian;
    glass._roughness = 0.2f;
    glass._ior = 1.5f;

    const char* env = std::getenv("HPT_MODELS_DIR");
    const std::string dir = std::string(env && *env ? env : "models") + "/cornellbox/";
    scene.Add(new MeshTriangle(dir + "floor.obj", white));
    scene.Add(new MeshTriangle(dir + "shortbox.obj", glass));
    scene.Add(new MeshTriangle(dir + "tallbox.obj", metal));
    scene.Add(new MeshTriangle(dir + "left.obj", red));
    scene.Add(new MeshTriangle(dir + "right.obj", green));
    scene.Add(new MeshTriangle(dir + "light.obj", light));
    scene.buildBVH();
    return scene;
}

// Cornell shell (floor incl. ceiling and back wall, left, right, light) + the bunny.  The OBJ's
// vertices are scaled by 2000 and moved so that the bounding box is centred at (x, z) =
// (278, 280) and its bottom lies on y = 0; all in float, one operation at a time, so that the
// vertices equal those of cudapathtracer_amd/scenes.py (bunny_triangles) bit for bit.
Scene Scene::CreateBunnyScene(int maxDepth) {
    Scene scene(512, 512);
    scene.name = "Bunny";
    scene.maxDepth = maxDepth;
    scene.camPos = Vec3(278, 273, -800);

    Material red, green, white, light;
    red._albedo = Vec3(0.63f, 0.065f, 0.05f);
    green._albedo = Vec3(0.14f, 0.45f, 0.091f);
    white._albedo = Vec3(0.725f, 0.71f, 0.68f);
    light._emission = (8.0f * Vec3(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) + 15.6f * Vec3(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                       18.4f * Vec3(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));

    const char* env = std::getenv("HPT_MODELS_DIR");
    const std::string root = std::string(env && *env ? env : "models");
    scene.Add(new MeshTriangle(root + "/cornellbox/floor.obj", white));
    scene.Add(new MeshTriangle(root + "/cornellbox/left.obj", red));
    scene.Add(new MeshTriangle(root + "/cornellbox/right.obj", green));
    scene.Add(new MeshTriangle(root + "/cornellbox/light.obj", light));

    MeshTriangle raw(root + "/bunny/bunny.obj", white);
    Vec3 lo = raw.vertices.at(0), hi = lo;
    for (const Vec3& v : raw.vertices) {
        lo = Vec3(std::min(lo.x, v.x), std::min(lo.y, v.y), std::min(lo.z, v.z));
        hi = Vec3(std::max(hi.x, v.x), std::max(hi.y, v.y), std::max(hi.z, v.z));
    }
    const float s = 2000.0f;
    const Vec3 off(278.0f - s * (lo.x + hi.x) * 0.5f, -s * lo.y, 280.0f - s * (lo.z + hi.z) * 0.5f);
    std::vector<Vec3> moved;
    moved.reserve(raw.vertices.size());
    for (const Vec3& v : raw.vertices) moved.push_back(Vec3(v.x * s + off.x, v.y * s + off.y, v.z * s + off.z));
    scene.Add(new MeshTriangle(moved, white));
    scene.buildBVH();
    return scene;
}

namespace {

// The materials of the reference's Cornell box (common/Scene.cpp:34-53)
struct CornellMaterials {
    Material red, green, white, light;
    CornellMaterials() {
        red._albedo = Vec3(0.63f, 0.065f, 0.05f);
        green._albedo = Vec3(0.14f, 0.45f, 0.091f);
        white._albedo = Vec3(0.725f, 0.71f, 0.68f);
        light._emission = (8.0f * Vec3(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) + 15.6f * Vec3(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                           18.4f * Vec3(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));
    }
};

void add_cornell_shell(Scene& scene, const CornellMaterials& m) {
    const char* env = std::getenv("HPT_MODELS_DIR");
    const std::string root = std::string(env && *env ? env : "models");
    scene.Add(new MeshTriangle(root + "/cornellbox/floor.obj", m.white));
    scene.Add(new MeshTriangle(root + "/cornellbox/left.obj", m.red));
    scene.Add(new MeshTriangle(root + "/cornellbox/right.obj", m.green));
    scene.Add(new MeshTriangle(root + "/cornellbox/light.obj", m.light));
}

// MT19937 as numpy.random.RandomState(seed) runs it: init_genrand seeding (what std::mt19937(seed)
// does) and 53-bit doubles from two outputs, (a >> 5) * 2^26 + (b >> 6) over 2^53.
struct NumpyRandom {
    std::mt19937 gen;
    explicit NumpyRandom(uint32_t seed) : gen(seed) {}
    double rand() {
        const uint32_t a = (uint32_t)gen() >> 5, b = (uint32_t)gen() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

// numpy.linspace(lo, hi, num): i * step + lo with step = (hi - lo) / (num - 1), the last element hi
std::vector<double> linspace(double lo, double hi, int num) {
    std::vector<double> v((size_t)num);
    const double step = (hi - lo) / (double)(num - 1);
    for (int i = 0; i < num; ++i) v[(size_t)i] = (double)i * step + lo;
    v[(size_t)num - 1] = hi;
    return v;
}

}  // namespace

// n x n cells over the floor rectangle [30, 526] x [30, 529] (x, z), heights 40 + 60 * smooth noise
// (a 9 x 9 random grid, bilinearly interpolated) + a little per-vertex jitter, two upward-facing
// triangles per cell.  Every operation in double, in the order cudapathtracer_amd/scenes.py
// (heightfield_triangles, seed 1234) performs it, then rounded to float once: the vertices are
// the same bit for bit.
Scene Scene::CreateHeightfieldScene(int n, int maxDepth) {
    Scene scene(512, 512);
    scene.name = "Heightfield " + std::to_string(n);
    scene.maxDepth = maxDepth;
    scene.camPos = Vec3(278, 273, -800);
    const CornellMaterials mats;
    add_cornell_shell(scene, mats);
    scene.Add(new MeshTriangle(HeightfieldVertices(n), mats.white));
    scene.buildBVH();
    return scene;
}

std::vector<Vec3> Scene::HeightfieldVertices(int n) {
    if (n < 1 || n > 4096) throw std::runtime_error("height field: n must be in [1, 4096]");
    const double base = 40.0, amplitude = 60.0;
    NumpyRandom rs(1234u);
    double coarse[9][9];
    for (auto& row : coarse)
        for (double& c : row) c = rs.rand();
    const int m = n + 1;
    const std::vector<double> u = linspace(0.0, 8.0, m);
    std::vector<int> i0((size_t)m);
    std::vector<double> f((size_t)m);
    for (int i = 0; i < m; ++i) {
        i0[(size_t)i] = std::min((int)u[(size_t)i], 7);
        f[(size_t)i] = u[(size_t)i] - (double)i0[(size_t)i];
    }
    std::vector<double> h((size_t)m * m);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = i0[(size_t)i], b = i0[(size_t)j];
            const double fi = f[(size_t)i], fj = f[(size_t)j];
            double rows = coarse[a][b] * ((1 - fi) * (1 - fj));
            rows = rows + coarse[a + 1][b] * (fi * (1 - fj));
            rows = rows + coarse[a][b + 1] * ((1 - fi) * fj);
            rows = rows + coarse[a + 1][b + 1] * (fi * fj);
            h[(size_t)i * m + j] = rows;
        }
    const double jitter = amplitude / (double)n;
    for (size_t k = 0; k < h.size(); ++k) h[k] = (base + amplitude * h[k]) + jitter * rs.rand();
    const std::vector<double> x = linspace(30.0, 526.0, m), z = linspace(30.0, 529.0, m);
    auto P = [&](int i, int j) { return Vec3((float)x[(size_t)i], (float)h[(size_t)i * m + j], (float)z[(size_t)j]); };
    std::vector<Vec3> fv;
    fv.reserve((size_t)n * n * 6);
    // counter-clockwise seen from above: first all (a, d, c) triangles, then all (a, c, b)
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const Vec3 a = P(i, j), b = P(i + 1, j), c = P(i + 1, j + 1), d = P(i, j + 1);
                fv.push_back(a);
                fv.push_back(pass == 0 ? d : c);
                fv.push_back(pass == 0 ? c : b);
            }
    return fv;
}

Scene Scene::CreateNamedScene(const std::string& name, int maxDepth) {
    if (name.empty() || name == "cornell") return CreateBuiltinScene(CornellBox, maxDepth);
    if (name == "bunny") return CreateBuiltinScene(BunnyInCornell, maxDepth);
    if (name == "glass") return CreateBuiltinScene(GlassCornell, maxDepth);
    if (name.rfind("heightfield:", 0) == 0) return CreateHeightfieldScene(std::atoi(name.c_str() + 12), maxDepth);
    if (name.rfind("file:", 0) == 0) return LoadSceneFile(name.substr(5), maxDepth);
    throw std::runtime_error("unknown scene '" + name + "' (cornell | bunny | glass | heightfield:<n> | file:<path>)");
}

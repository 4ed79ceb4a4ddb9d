// GPU build of the two-child traversal records (DevScene::pairs) from the triangle records alone.
//
// The reference builds its BVH on the host, top-down: leaf = one primitive; two primitives = (first,
// second); otherwise sort the primitives by the centroid of their bounds along the longest axis of the
// centroid bounds and split in the middle (BVHAccel::recursiveBuild, common/BVH.cpp:81-168).  This file
// runs that recursion level by level on the GPU, for all objects of the scene at once (an object = a run of
// primitives with the same object word, i.e. a mesh or a sphere): every level is ONE radix sort of the whole
// primitive array by (start of the primitive's current segment, centroid coordinate on the segment's axis),
// then every segment of three or more splits at its middle.  The shape of an object's tree depends on its
// primitive count only (a segment [b, e) always splits at b + (e - b) / 2), so record numbers and child
// references follow from the counts: an object's records are numbered in pre-order of its interior nodes.
// The tree over the objects -- a handful of boxes -- is built on the host by the same rule once the
// objects' boxes are known, like the reference's two-level arrangement (common/Scene.cpp:9-23).
//
// What is the same as the reference's tree and what is not:
//   * leaf boxes are Triangle::getBounds / the sphere's box, interior boxes are min/max unions of their
//     children -- so boxes nest, and a ray with finite reciprocal direction components that hits a leaf's
//     box hits every box above it (float subtraction and multiplication are monotone).  The triangles that
//     get tested are therefore those whose leaf box the ray hits: the reference's set.  Every closest hit
//     (minimum t) and every occlusion answer is the reference's.
//   * which primitives share a subtree can differ: primitives whose centroids tie on the split axis keep
//     their previous order here (a stable sort) where std::sort leaves it to the implementation.  Only the ORDER of tests differs, which matters in one case: two triangles hit at
//     bitwise the same t.  The reference keeps the one it meets first; records built here keep the one with
//     the lower primitive id.  Rays with a zero direction component (NaN slab products) are outside the
//     argument above as well.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace hpt_build {

constexpr uint32_t kSphereBit = 0x80000000u;

struct Seg {
    uint32_t b, e, id;  // the segment [b, e) a position currently lies in, and its record number
};

// floats as unsigned integers of the same order (for atomicMin / atomicMax and for sort keys)
__device__ __forceinline__ uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// Triangle::getBounds = Union(Bounds3(v0, v1), v2) (common/Triangle.hpp:162); a sphere's box is center -+ radius
__global__ void k_leaf_boxes(uint32_t n, const float4* __restrict__ tris, const float4* __restrict__ v12, float4* __restrict__ lo, float4* __restrict__ hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = tris[4 * i], r1 = tris[4 * i + 1], tag = tris[4 * i + 3];
    float4 l, h;
    if (__float_as_uint(tag.w) & kSphereBit) {
        const float rad = r1.x;
        l = make_float4(a.x - rad, a.y - rad, a.z - rad, 0.f);
        h = make_float4(a.x + rad, a.y + rad, a.z + rad, 0.f);
    } else {
        const float4 b = v12[2 * i], c = v12[2 * i + 1];
        l = make_float4(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z), 0.f);
        h = make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), 0.f);
    }
    lo[i] = l;
    hi[i] = h;
}
// Bounds3::Centroid (common/Bounds3.hpp:35): 0.5f * pMin + 0.5f * pMax
__device__ __forceinline__ float centroid(const float4& l, const float4& h, int axis) {
    const float a = axis == 0 ? l.x : (axis == 1 ? l.y : l.z), b = axis == 0 ? h.x : (axis == 1 ? h.y : h.z);
    return 0.5f * a + 0.5f * b;
}

// every position starts in the segment of its object: obj[k] = {first primitive, end, number of its root record}
__global__ void k_init(uint32_t n, uint32_t num_objects, const Seg* __restrict__ obj, Seg* seg, Seg* seg0, uint32_t* perm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = num_objects - 1u;  // the last object whose first primitive is <= i
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) / 2u;
        if (obj[mid].b <= i) lo = mid;
        else hi = mid - 1u;
    }
    seg[i] = seg0[i] = obj[lo];
    perm[i] = i;
}
__global__ void k_clear_bounds(uint32_t n, const Seg* __restrict__ seg, uint32_t* __restrict__ cb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (seg[p].b != p) return;
    for (int c = 0; c < 3; ++c) {
        cb[6 * p + c] = 0xFFFFFFFFu;
        cb[6 * p + 3 + c] = 0u;
    }
}
// bounds of the centroids of every segment that will be sorted (three or more primitives), kept at the segment's start.
// Positions are contiguous, so near the root a whole wave lies in one segment: it reduces in registers and issues six
// atomics instead of 384 (a million atomics on one address take tens of milliseconds).
__global__ void k_centroid_bounds(uint32_t n, const Seg* __restrict__ seg, const uint32_t* __restrict__ perm, const float4* __restrict__ lo,
                                  const float4* __restrict__ hi, uint32_t* __restrict__ cb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = p < n;
    Seg s{0u, 0u, 0u};
    if (in) s = seg[p];
    const bool active = in && s.e - s.b >= 3u;
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    if (active) {
        const uint32_t prim = perm[p];
        const float4 l = lo[prim], h = hi[prim];
        for (int c = 0; c < 3; ++c) mn[c] = mx[c] = ordered(centroid(l, h, c));
    }
    const unsigned long long m_active = __ballot(active);
    if (m_active == 0ull) return;
    const int leader = __ffsll((long long)m_active) - 1;
    const uint32_t b0 = (uint32_t)__shfl((int)s.b, leader);
    const bool uniform = __ballot(active && s.b != b0) == 0ull;  // every active lane of the wave lies in the leader's segment
    if (uniform) {
        for (int c = 0; c < 3; ++c) {
            for (int off = 32; off >= 1; off >>= 1) {  // inactive lanes hold the neutral elements
                mn[c] = min(mn[c], (uint32_t)__shfl_xor((int)mn[c], off));
                mx[c] = max(mx[c], (uint32_t)__shfl_xor((int)mx[c], off));
            }
        }
        if ((int)(threadIdx.x & 63u) == leader) {
            for (int c = 0; c < 3; ++c) {
                atomicMin(&cb[6 * b0 + c], mn[c]);
                atomicMax(&cb[6 * b0 + 3 + c], mx[c]);
            }
        }
    } else if (active) {
        for (int c = 0; c < 3; ++c) {
            atomicMin(&cb[6 * s.b + c], mn[c]);
            atomicMax(&cb[6 * s.b + 3 + c], mx[c]);
        }
    }
}
// sort key: segment start, then the centroid on the segment's longest axis (Bounds3::maxExtent,
// common/Bounds3.hpp:22-30: x if it exceeds y and z, else y if it exceeds z, else z); segments of one or
// two keep their order
__global__ void k_keys(uint32_t n, const Seg* __restrict__ seg, const uint32_t* __restrict__ perm, const float4* __restrict__ lo,
                       const float4* __restrict__ hi, const uint32_t* __restrict__ cb, unsigned long long* __restrict__ keys) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Seg s = seg[p];
    uint32_t low = p - s.b;
    if (s.e - s.b >= 3u) {
        const float dx = unordered(cb[6 * s.b + 3]) - unordered(cb[6 * s.b]);
        const float dy = unordered(cb[6 * s.b + 4]) - unordered(cb[6 * s.b + 1]);
        const float dz = unordered(cb[6 * s.b + 5]) - unordered(cb[6 * s.b + 2]);
        const int axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);
        const uint32_t prim = perm[p];
        low = ordered(centroid(lo[prim], hi[prim], axis));
    }
    keys[p] = ((unsigned long long)s.b << 32) | low;
}
__global__ void k_split(uint32_t n, Seg* seg) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Seg s = seg[p];
    if (s.e - s.b < 2u) return;
    const uint32_t mid = s.b + (s.e - s.b) / 2u;  // BVH.cpp:137: beginning + objects.size() / 2
    if (p < mid) {
        s.e = mid;
        s.id += 1u;
    } else {
        s.id += mid - s.b;  // the left subtree holds (mid - b) - 1 interior nodes, plus this one
        s.b = mid;
    }
    seg[p] = s;
}

struct Box {
    float4 lo, hi;
};
__device__ __forceinline__ Box unite(const Box& a, const Box& b) {
    Box r;
    r.lo = make_float4(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z), 0.f);
    r.hi = make_float4(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z), 0.f);
    return r;
}
// the box and the reference of the subtree over positions [b, e) whose record (if it has one) is `id`
__device__ __forceinline__ Box child_box(uint32_t b, uint32_t e, uint32_t id, const uint32_t* perm, const float4* lo, const float4* hi,
                                         const float4* rec, int32_t& ref) {
    Box r;
    if (e - b == 1u) {
        const uint32_t prim = perm[b];
        ref = ~(int32_t)prim;
        r.lo = lo[prim];
        r.hi = hi[prim];
        return r;
    }
    ref = (int32_t)id;
    const Box l{rec[4 * id], rec[4 * id + 1]}, rr{rec[4 * id + 2], rec[4 * id + 3]};
    return unite(l, rr);
}
// Records of the nodes `level` splits below the root, once the records of all deeper levels exist.
// Record: left.pMin, left ref | left.pMax, right ref | right.pMin | right.pMax (ref >= 0 record, < 0 = ~primitive)
__global__ void k_emit_level(uint32_t n, uint32_t level, const Seg* __restrict__ seg0, const uint32_t* __restrict__ perm,
                             const float4* __restrict__ lo, const float4* __restrict__ hi, float4* rec) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t b = seg0[p].b, e = seg0[p].e, id = seg0[p].id;
    for (uint32_t l = 0; l < level; ++l) {
        if (e - b < 2u) return;
        const uint32_t mid = b + (e - b) / 2u;
        if (p < mid) {
            e = mid;
            id += 1u;
        } else {
            id += mid - b;
            b = mid;
        }
    }
    if (p != b || e - b < 2u) return;
    const uint32_t mid = b + (e - b) / 2u;
    int32_t ref_l, ref_r;
    const Box bl = child_box(b, mid, id + 1u, perm, lo, hi, rec, ref_l);
    const Box br = child_box(mid, e, id + (mid - b), perm, lo, hi, rec, ref_r);
    rec[4 * id] = make_float4(bl.lo.x, bl.lo.y, bl.lo.z, __int_as_float(ref_l));
    rec[4 * id + 1] = make_float4(bl.hi.x, bl.hi.y, bl.hi.z, __int_as_float(ref_r));
    rec[4 * id + 2] = make_float4(br.lo.x, br.lo.y, br.lo.z, 0.f);
    rec[4 * id + 3] = make_float4(br.hi.x, br.hi.y, br.hi.z, 0.f);
}

// box of every object: its root record's two boxes, or the box of its only primitive
__global__ void k_object_boxes(uint32_t num_objects, const Seg* __restrict__ obj, const uint32_t* __restrict__ perm, const float4* __restrict__ lo,
                               const float4* __restrict__ hi, const float4* __restrict__ rec, float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_objects) return;
    int32_t ref;
    const Box b = child_box(obj[k].b, obj[k].e, obj[k].id, perm, lo, hi, rec, ref);
    out[2 * k] = make_float4(b.lo.x, b.lo.y, b.lo.z, __int_as_float(ref));
    out[2 * k + 1] = b.hi;
}
// new position of every primitive: the triangle arrays are gathered into leaf order, so that leaves that are
// neighbours in the tree are neighbours in memory
__global__ void k_gather(uint32_t n, const uint32_t* __restrict__ perm, const float4* __restrict__ tris, const float4* __restrict__ v12,
                         float4* __restrict__ tris_out, float4* __restrict__ v12_out, uint32_t* __restrict__ new_of_old) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t prim = perm[p];
    for (int c = 0; c < 4; ++c) tris_out[4 * p + c] = tris[4 * prim + c];
    v12_out[2 * p] = v12[2 * prim];
    v12_out[2 * p + 1] = v12[2 * prim + 1];
    new_of_old[prim] = p;
}
// leaf references of the records, and the leaf words of the light meshes' sampling trees, follow the gather
__global__ void k_rename_refs(uint32_t num_records, float4* rec, const uint32_t* __restrict__ new_of_old) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_records) return;
    const int32_t rl = __float_as_int(rec[4 * i].w), rr = __float_as_int(rec[4 * i + 1].w);
    if (rl < 0) rec[4 * i].w = __int_as_float(~(int32_t)new_of_old[~rl]);
    if (rr < 0) rec[4 * i + 1].w = __int_as_float(~(int32_t)new_of_old[~rr]);
}
__global__ void k_rename_node_leaves(uint32_t num_nodes, float4* nodes, const uint32_t* __restrict__ new_of_old) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_nodes) return;
    const int32_t t = __float_as_int(nodes[2 * i + 1].w);
    if (t >= 0) nodes[2 * i + 1].w = __int_as_float((int32_t)new_of_old[t]);
}

// stack entries a depth-first walk can hold below a node over m primitives (the right child waits while the
// left subtree is walked); `any`: when either child may be walked first
struct Need {
    uint32_t fixed, any;
};
static Need need_of(uint32_t m, std::map<uint32_t, Need>& memo) {
    if (m < 2u) return Need{0u, 0u};
    auto it = memo.find(m);
    if (it != memo.end()) return it->second;
    const uint32_t l = m / 2u, r = m - l;
    const Need nl = need_of(l, memo), nr = need_of(r, memo);
    Need out;
    out.fixed = std::max(l >= 2u ? nl.fixed + (r >= 2u ? 1u : 0u) : 0u, r >= 2u ? nr.fixed : 0u);
    out.any = std::max(l >= 2u ? nl.any : 0u, r >= 2u ? nr.any : 0u) + ((l >= 2u && r >= 2u) ? 1u : 0u);
    memo[m] = out;
    return out;
}

#define BUILD_TRY(expr)                                                                      \
    do {                                                                                     \
        const hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)

// The tree over the objects, on the host, by the reference's rule (BVHAccel::recursiveBuild over the scene's
// objects, common/BVH.cpp:92-168): one object = that object's subtree; two = (first, second); otherwise sort by
// the centroid on the longest axis of the centroid bounds and split in the middle.
struct Sub {
    float lo[3], hi[3];
    int32_t ref;          // >= 0 record, < 0 = ~primitive
    uint32_t need, any;   // stack entries below (see need_of)
};
static Sub build_top(std::vector<Sub>& objs, size_t begin, size_t end, std::vector<float>& rec, uint32_t& next_id) {
    if (end - begin == 1) return objs[begin];
    const uint32_t id = next_id++;
    if (end - begin > 2) {
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = begin; k < end; ++k)
            for (int c = 0; c < 3; ++c) {
                const float ctr = 0.5f * objs[k].lo[c] + 0.5f * objs[k].hi[c];
                clo[c] = std::min(clo[c], ctr);
                chi[c] = std::max(chi[c], ctr);
            }
        const float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
        const int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);
        std::stable_sort(objs.begin() + begin, objs.begin() + end,
                         [dim](const Sub& a, const Sub& b) { return 0.5f * a.lo[dim] + 0.5f * a.hi[dim] < 0.5f * b.lo[dim] + 0.5f * b.hi[dim]; });
    }
    const size_t mid = begin + (end - begin) / 2;
    const Sub L = build_top(objs, begin, mid, rec, next_id), R = build_top(objs, mid, end, rec, next_id);
    float* q = &rec[16 * (size_t)id];
    for (int c = 0; c < 3; ++c) {
        q[c] = L.lo[c];
        q[4 + c] = L.hi[c];
        q[8 + c] = R.lo[c];
        q[12 + c] = R.hi[c];
    }
    std::memcpy(&q[3], &L.ref, 4);
    std::memcpy(&q[7], &R.ref, 4);
    q[11] = q[15] = 0.f;
    Sub out;
    for (int c = 0; c < 3; ++c) {
        out.lo[c] = std::min(L.lo[c], R.lo[c]);
        out.hi[c] = std::max(L.hi[c], R.hi[c]);
    }
    out.ref = (int32_t)id;
    const bool li = L.ref >= 0, ri = R.ref >= 0;
    out.need = std::max(li ? L.need + (ri ? 1u : 0u) : 0u, ri ? R.need : 0u);
    out.any = std::max(li ? L.any : 0u, ri ? R.any : 0u) + ((li && ri) ? 1u : 0u);
    return out;
}

// Builds the n - 1 records for the n >= 2 primitives of `tris` / `v12` (device arrays, 4 / 2 float4 per primitive)
// into rec_out (device, 4 * (n - 1) float4, allocated by the caller) and gathers `tris` / `v12` into leaf order
// (the leaf words of `nodes`, the light meshes' sampling trees, follow).  obj_first[k] = first primitive of
// object k (ascending, obj_first[0] = 0).  Returns an empty string, or what failed.
std::string build_pairs(float4* tris, float4* v12, uint32_t n, const uint32_t* obj_first, uint32_t num_objects, float4* nodes, uint32_t num_nodes,
                        float4* rec_out, uint32_t* stack_depth, uint32_t* stack_depth_any, float root_lo[3], float root_hi[3], double* build_ms) {
    std::string err;
    float4 *lo = nullptr, *hi = nullptr, *obj_box = nullptr, *tris_tmp = nullptr, *v12_tmp = nullptr;
    Seg *seg = nullptr, *seg0 = nullptr, *obj = nullptr;
    uint32_t *perm[2] = {nullptr, nullptr}, *cb = nullptr, *new_of_old = nullptr;
    unsigned long long* keys[2] = {nullptr, nullptr};
    void* temp = nullptr;
    size_t temp_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const uint32_t block = 256, grid = (n + block - 1) / block;
    // record numbers: the tree over the objects takes 0 .. num_objects - 2, then one block per object
    std::vector<Seg> objs(num_objects);
    uint32_t levels = 0, next_base = num_objects - 1u;
    for (uint32_t k = 0; k < num_objects; ++k) {
        const uint32_t b = obj_first[k], e = k + 1 < num_objects ? obj_first[k + 1] : n;
        objs[k] = Seg{b, e, next_base};
        if (e - b >= 2u) next_base += e - b - 1u;
        uint32_t l = 0;
        for (uint32_t m = e - b; m > 1u; m = m - m / 2u) ++l;  // the larger half decides
        levels = std::max(levels, l);
    }
    int seg_bits = 1;
    while ((1ull << seg_bits) < n) ++seg_bits;
    int cur = 0;
    BUILD_TRY(hipEventCreate(&ev0));
    BUILD_TRY(hipEventCreate(&ev1));
    BUILD_TRY(hipMalloc((void**)&lo, (size_t)n * 16));
    BUILD_TRY(hipMalloc((void**)&hi, (size_t)n * 16));
    BUILD_TRY(hipMalloc((void**)&seg, (size_t)n * sizeof(Seg)));
    BUILD_TRY(hipMalloc((void**)&seg0, (size_t)n * sizeof(Seg)));
    BUILD_TRY(hipMalloc((void**)&obj, (size_t)num_objects * sizeof(Seg)));
    BUILD_TRY(hipMalloc((void**)&obj_box, (size_t)num_objects * 32));
    BUILD_TRY(hipMalloc((void**)&perm[0], (size_t)n * 4));
    BUILD_TRY(hipMalloc((void**)&perm[1], (size_t)n * 4));
    BUILD_TRY(hipMalloc((void**)&new_of_old, (size_t)n * 4));
    BUILD_TRY(hipMalloc((void**)&keys[0], (size_t)n * 8));
    BUILD_TRY(hipMalloc((void**)&keys[1], (size_t)n * 8));
    BUILD_TRY(hipMalloc((void**)&cb, (size_t)n * 24));
    BUILD_TRY(hipMalloc((void**)&tris_tmp, (size_t)n * 64));
    BUILD_TRY(hipMalloc((void**)&v12_tmp, (size_t)n * 32));
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys[0], keys[1], perm[0], perm[1], (int)n, 0, 32 + seg_bits, nullptr));
    BUILD_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    BUILD_TRY(hipMemcpy(obj, objs.data(), (size_t)num_objects * sizeof(Seg), hipMemcpyHostToDevice));
    BUILD_TRY(hipEventRecord(ev0, nullptr));
    hipLaunchKernelGGL(k_leaf_boxes, dim3(grid), dim3(block), 0, nullptr, n, tris, v12, lo, hi);
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(block), 0, nullptr, n, num_objects, obj, seg, seg0, perm[0]);
    for (uint32_t level = 0; level < levels; ++level) {
        hipLaunchKernelGGL(k_clear_bounds, dim3(grid), dim3(block), 0, nullptr, n, seg, cb);
        hipLaunchKernelGGL(k_centroid_bounds, dim3(grid), dim3(block), 0, nullptr, n, seg, perm[cur], lo, hi, cb);
        hipLaunchKernelGGL(k_keys, dim3(grid), dim3(block), 0, nullptr, n, seg, perm[cur], lo, hi, cb, keys[0]);
        BUILD_TRY(hipGetLastError());
        BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys[0], keys[1], perm[cur], perm[cur ^ 1], (int)n, 0, 32 + seg_bits, nullptr));
        cur ^= 1;
        hipLaunchKernelGGL(k_split, dim3(grid), dim3(block), 0, nullptr, n, seg);
    }
    for (uint32_t level = levels; level-- > 0;) hipLaunchKernelGGL(k_emit_level, dim3(grid), dim3(block), 0, nullptr, n, level, seg0, perm[cur], lo, hi, rec_out);
    hipLaunchKernelGGL(k_object_boxes, dim3((num_objects + block - 1) / block), dim3(block), 0, nullptr, num_objects, obj, perm[cur], lo, hi, rec_out, obj_box);
    BUILD_TRY(hipGetLastError());
    {
        // the tree over the objects: a few boxes, on the host
        std::vector<float> boxes((size_t)num_objects * 8);
        BUILD_TRY(hipMemcpy(boxes.data(), obj_box, boxes.size() * 4, hipMemcpyDeviceToHost));
        std::map<uint32_t, Need> memo;
        std::vector<Sub> subs(num_objects);
        for (uint32_t k = 0; k < num_objects; ++k) {
            Sub& sb = subs[k];
            for (int c = 0; c < 3; ++c) {
                sb.lo[c] = boxes[8 * (size_t)k + c];
                sb.hi[c] = boxes[8 * (size_t)k + 4 + c];
            }
            std::memcpy(&sb.ref, &boxes[8 * (size_t)k + 3], 4);
            const Need nd = need_of(objs[k].e - objs[k].b, memo);
            sb.need = nd.fixed;
            sb.any = nd.any;
        }
        std::vector<float> top((size_t)(num_objects - 1u) * 16 + 16);
        uint32_t next_id = 0;
        const Sub root = build_top(subs, 0, num_objects, top, next_id);
        if (num_objects > 1u) BUILD_TRY(hipMemcpy(rec_out, top.data(), (size_t)(num_objects - 1u) * 64, hipMemcpyHostToDevice));
        for (int c = 0; c < 3; ++c) {
            root_lo[c] = root.lo[c];
            root_hi[c] = root.hi[c];
        }
        *stack_depth = std::max(1u, root.need);
        *stack_depth_any = std::max(1u, root.any);
    }
    // leaf order in memory
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(block), 0, nullptr, n, perm[cur], tris, v12, tris_tmp, v12_tmp, new_of_old);
    BUILD_TRY(hipMemcpyAsync(tris, tris_tmp, (size_t)n * 64, hipMemcpyDeviceToDevice, nullptr));
    BUILD_TRY(hipMemcpyAsync(v12, v12_tmp, (size_t)n * 32, hipMemcpyDeviceToDevice, nullptr));
    hipLaunchKernelGGL(k_rename_refs, dim3((n - 1u + block - 1) / block), dim3(block), 0, nullptr, n - 1u, rec_out, new_of_old);
    hipLaunchKernelGGL(k_rename_node_leaves, dim3((num_nodes + block - 1) / block), dim3(block), 0, nullptr, num_nodes, nodes, new_of_old);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipEventRecord(ev1, nullptr));
    BUILD_TRY(hipDeviceSynchronize());
    {
        float ms = 0.f;
        BUILD_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        if (build_ms) *build_ms = ms;
    }
done:
    (void)hipFree(lo);
    (void)hipFree(hi);
    (void)hipFree(seg);
    (void)hipFree(seg0);
    (void)hipFree(obj);
    (void)hipFree(obj_box);
    (void)hipFree(perm[0]);
    (void)hipFree(perm[1]);
    (void)hipFree(new_of_old);
    (void)hipFree(keys[0]);
    (void)hipFree(keys[1]);
    (void)hipFree(cb);
    (void)hipFree(tris_tmp);
    (void)hipFree(v12_tmp);
    (void)hipFree(temp);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return err;
}

}  // namespace hpt_build

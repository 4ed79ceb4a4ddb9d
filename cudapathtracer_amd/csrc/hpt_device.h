// Device-side building blocks of the gfx950 path-tracing core: vector math in
// the reference's (glm) operation order, the keyed Philox stream, the portable
// sin/cos/acos, the BVH traversal variants over the flattened node/triangle
// records and one bounce of the MIS integrator.
//
// Arithmetic contract: this translation unit is compiled with
// -ffp-contract=off; every float/double operation below is a single IEEE-754
// operation in a fixed order, divisions and square roots are correctly
// rounded.  Results therefore do not depend on the compiler's scheduling and
// can be compared bit for bit with a CPU evaluation of the same expressions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/hpt.h"

// rcp_rn, sqrt_rn, two_over and reciprocal_of_float below are correctly rounded because gfx950's
// v_rcp / v_sqrt / v_rsq approximations make them so (checked for every float by
// tools/micro/exact_math_check.hip, which includes this header); another GPU's approximations are unverified.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "hpt_device.h is written and verified for gfx950 only"
#endif

namespace hpt {

// Diagnostic build only (-DHPT_PROFILE): per-section shader-clock and lane-occupancy
// totals, written to a buffer of their own; the product build compiles none of it.
#ifdef HPT_PROFILE
enum { kProfRegen, kProfBoxC, kProfTriC, kProfShadeA, kProfBoxS, kProfTriS, kProfShadeB, kProfDone, kProfSections };
struct ProfAcc {
    unsigned long long* w;  // per-wave LDS: cyc[kProfSections], iters[kProfSections], lanes[kProfSections]
};
__device__ __forceinline__ bool prof_first_lane(unsigned long long exec) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(exec >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)exec, 0u)) == 0u;
}
__device__ __forceinline__ void prof_add(const ProfAcc& p, int idx, unsigned long long v) {
    const unsigned long long exec = __ballot(true);
    if (prof_first_lane(exec)) atomicAdd(p.w + idx, v);
}
#define PROF_ARG , ProfAcc& prof
#define PROF_PASS , prof
#define PROF_T0() const unsigned long long prof_t0 = __builtin_readcyclecounter()
#define PROF_T1(id) prof_add(prof, id, __builtin_readcyclecounter() - prof_t0)
#define PROF_ITER(id) do { prof_add(prof, kProfSections + (id), 1ull); prof_add(prof, 2 * kProfSections + (id), (unsigned long long)__popcll(__ballot(true))); } while (0)
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_T0()
#define PROF_T1(id)
#define PROF_ITER(id)
#endif

// Diagnostic build only (-DHPT_COUNT_TESTS, `make count`): every lane counts the box and
// triangle tests it executes (a test the back-face filter settles counts as a triangle test);
// the kernels add the totals to two more statistics slots.  The product build compiles none of it.
#ifdef HPT_COUNT_TESTS
struct TestCount {
    unsigned long long box, tri;
};
#define COUNT_ARG , TestCount& tcount
#define COUNT_PASS , tcount
#define COUNT_BOX(n) (tcount.box += (n))
#define COUNT_TRI(n) (tcount.tri += (n))
#else
#define COUNT_ARG
#define COUNT_PASS
#define COUNT_BOX(n)
#define COUNT_TRI(n)
#endif

// ---------------------------------------------------------------- vec3 ------
// glm::vec3 semantics (external/glm/detail/func_geometric.inl:54-96):
// dot = (x*x' + y*y') + z*z', cross as written there, normalize = v * (1/sqrt(dot)).
struct f3 {
    float x, y, z;
};
#define HD __device__ __forceinline__

// Lane mask of a predicate.  A compound predicate is best given as the AND of the masks of its comparisons: the
// compiler turns ballot(a && b) into a 0/1 value per lane and a second comparison, two vector instructions that the
// scalar AND of two comparison masks does not need.
HD unsigned long long lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Lanes in a mask, as a 32-bit scalar (the 64-bit population count keeps its comparisons 64 bits wide, and a 64-bit
// less-than is a vector instruction)
HD uint32_t lane_count(unsigned long long m) { return (uint32_t)__builtin_popcount((uint32_t)m) + (uint32_t)__builtin_popcount((uint32_t)(m >> 32)); }

HD f3 mk(float a, float b, float c) { return f3{a, b, c}; }
HD f3 mk(float a) { return f3{a, a, a}; }
HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
HD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
HD f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
HD f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
HD float dot(f3 a, f3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
HD f3 cross(f3 x, f3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// Correctly rounded 1.0f / x and sqrtf(x) in fewer instructions than the compiler's general IEEE
// sequences (which carry rescaling and fix-up steps for operands near the ends of the float
// range).  For |x| in [1e-30, 1e30] the hardware approximation plus ONE fused correction step
// already is the correctly rounded result: checked for every float in that range against the
// IEEE division / square root (tools/micro/f32_exact_check.hip: 0 differ; the bare
// approximations differ for 3.6e8 / 2.5e8 inputs).  Anything else -- zero, denormal, huge,
// infinite, negative (sqrt), NaN -- takes the compiler's sequence.
HD float rcp_rn(float x) {
    const float ax = fabsf(x);
    if (!(ax >= 1.0e-30f && ax <= 1.0e30f)) return 1.0f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
HD float sqrt_rn(float x) {
    if (!(x >= 1.0e-30f && x <= 1.0e30f)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
// 2.0f / x: twice the correctly rounded reciprocal where neither can leave the normal range
HD float two_over(float x) {
    const float ax = fabsf(x);
    if (!(ax >= 1.0e-30f && ax <= 1.0e30f)) return 2.0f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return 2.0f * __builtin_fmaf(e, r, r);
}
HD float length(f3 v) { return sqrt_rn(dot(v, v)); }
HD f3 normalize(f3 v) { return v * rcp_rn(sqrt_rn(dot(v, v))); }
HD float gmax(float x, float y) { return x > y ? x : y; }  // glm::max (external/glm/detail/func_common.inl:23-28): NaN -> y
HD float gmin(float x, float y) { return x < y ? x : y; }  // glm::min (external/glm/detail/func_common.inl:15-19)
HD f3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

// common/MathUtils.hpp:64-74
constexpr float kPi = 3.1415926535897932384626422832795028841971f;
constexpr float kEpsilon5 = 1e-5f;
constexpr float kEpsilon4 = (5 * 1e-4f);
constexpr float kInfinity = 1e30f;

// ------------------------------------------------------------- Philox -------
// Philox4x32-10; key = (pixel, sample), counter = (depth, block, seed, 0).
// A path's draws are addressed, never streamed, so a sample's radiance does
// not depend on which lane, launch or GPU computes it.
struct u4 {
    uint32_t x, y, z, w;
};
HD u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        // one 32x32->64 multiply per product (the 32-bit integer multiplies are the slow VALU ops)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
HD float u01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }
constexpr uint32_t kCameraDepth = 0xFFFFFFFFu;

// ------------------------------------------------------- portable trig ------
// sin/cos by quadrant reduction (3-term pi/2) + degree-7/8 polynomials, acos by
// the asin polynomial on [0, 0.5]; about 1 ulp on the ranges used here
// ([0, 2pi] and [0, 1]).  Written out operation by operation because the
// sequence *is* the definition.
HD void portable_sincos(float x, float& s_out, float& c_out) {
    float kf = x * 0.636619772f;
    kf = kf + 0.5f;
    int k = (int)kf;
    float fk = (float)k;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    float s = r * r;
    float sp = -1.9515295891e-4f * s;
    sp = sp + 8.3321608736e-3f;
    sp = sp * s;
    sp = sp - 1.6666654611e-1f;
    sp = sp * s;
    sp = sp * r;
    sp = sp + r;
    float cp = 2.443315711809948e-5f * s;
    cp = cp - 1.388731625493765e-3f;
    cp = cp * s;
    cp = cp + 4.166664568298827e-2f;
    cp = cp * s;
    cp = cp * s;
    float hs = 0.5f * s;
    cp = cp - hs;
    cp = cp + 1.0f;
    int q = k & 3;
    float sv = (q & 1) ? cp : sp;
    float cv = (q & 1) ? sp : cp;
    s_out = (q & 2) ? -sv : sv;
    c_out = (q == 1 || q == 2) ? -cv : cv;
}
HD float portable_asin_poly(float a, float z) {
    float p = 4.2163199048e-2f * z;
    p = p + 2.4181311049e-2f;
    p = p * z;
    p = p + 4.5470025998e-2f;
    p = p * z;
    p = p + 7.4953002686e-2f;
    p = p * z;
    p = p + 1.6666752422e-1f;
    p = p * z;
    p = p * a;
    p = p + a;
    return p;
}
HD float portable_acos(float x) {
    // Three ranges (x > 0.5, x < -0.5, between) that differ in the argument of one polynomial; the lanes of a wave
    // fall into all three, so the ranges are selected, not branched on.  1 + x for x < -0.5 is 1 - |x|, the same float.
    const bool outer = fabsf(x) > 0.5f;
    const float z = outer ? 0.5f * (1.0f - fabsf(x)) : x * x;
    const float a = outer ? sqrt_rn(z) : x;
    const float p = portable_asin_poly(a, z);
    const float inner = 1.5707963267948966f - p;
    const float twice = 2.0f * p;
    return outer ? (x > 0.5f ? twice : kPi - twice) : inner;
}

// ------------------------------------------------- fixed-point radiance -----
HD long long radiance_to_fixed(float v) {
    // |v| < 64: v * 2^24 is exact in float (a power-of-two scale) and below 2^30, so the float rounding to an integer and
    // the int32 conversion give the value of the double sequence below in five instructions instead of thirteen
    if (fabsf(v) < 64.0f) return (long long)(int)__builtin_rintf(v * 16777216.0f);
    double x = (double)v;
    if (!(x == x)) return 0;
    if (x > 1048576.0) x = 1048576.0;
    if (x < -1048576.0) x = -1048576.0;
    return (long long)__builtin_rint(x * 16777216.0);
}

// ---------------------------------------------------------------- scene -----
// Device-side material / light tables, derived from hpt_material / hpt_light at
// upload.  The derived members are values the integrator would otherwise
// recompute at every bounce from constants of the material; they are computed
// on the host with the same float operations in the same order, so using them
// changes no result.
struct DevMaterial {
    int32_t type;
    uint32_t flags;            // kMatEmitting | kMatSpecular
    float emission[3];
    float albedo[3];
    float albedo_over_pi[3];   // albedo * 1.f / Pi            (Material.hpp:318)
    float f0[3];               // lerp(0.04, albedo, metallic)  (Material.hpp:335-336)
    float metallic, one_minus_metallic, roughness, rsq, rsq_minus_1;
    float ior;                 // Glass only
};  // 80 bytes
enum : uint32_t { kMatEmitting = 1u, kMatSpecular = 2u };
struct DevLight {
    float area, root_area, inv_root_area;  // inv_root_area = 1.0f / root_area (BVH.hpp:183)
    uint32_t root_node;
    float emission[3];
    float _pad;
};  // 32 bytes
// In the device copy of the triangle records the `area` slot (word 3) holds the
// triangle's MIS "light pdf" instead:
//   luminance(emission) * 2 * pi * area / sumLightPower   (Scene.hpp:179, :215)
// which is the only thing the integrator uses the area for.

struct DevScene {
    const float4* nodes;      // 2 x float4 per node: (pMin, miss link) (pMax, triangle | -1)
    const float* node_area;
    const float4* tris;       // 4 x float4 per triangle
    const float4* tri_v12;    // 2 x float4 per triangle
    const DevMaterial* mats;
    const DevLight* lights;
    uint32_t num_nodes, num_tris, num_mats, num_lights;
    float sum_light_area, sum_light_power;
    float cam_x, cam_y, cam_z, cam_scale, aspect;
    int width, height;
    uint32_t seed;
    uint32_t lds_float4s;  // > 0: that many float4 of node (or pair) + triangle records are staged into LDS
    int32_t variant;       // kSceneGlobal .. kScenePairLds
    // marking variants: nine pre-processed copies of the node array (see stage_scene) and their
    // table: dword k = nodes in copy k, dwords 16 + 2k / 17 + 2k = triangles whose back-face test
    // is still needed for a ray of sign octant k
    const float4* oct_nodes;
    const uint32_t* oct_table;
    // two-child records of the same tree (variants kScenePair / kScenePairLds), 4 x float4 per
    // interior node, in pre-order:
    //   (left.pMin, left ref) (left.pMax, right ref) (right.pMin, -) (right.pMax, -)
    // ref >= 0: index of the child's own record; ref < 0: the child is a leaf, triangle ~ref
    const float4* pairs;
    uint32_t num_pairs;
    uint32_t stack_depth;  // entries of the per-lane traversal stack that no ray can exceed
    float root_lo[3], root_hi[3];  // bounds of node 0
    // kScenePairCull only: slack of the distance tests (scene units), and the scale that packs a
    // box-entry distance into the 10 bits above a stack entry's record index (and its inverse)
    float cull_slack, cull_qscale, cull_qinv;
    unsigned long long* prof;  // diagnostic builds only (3 x kProfSections totals), else null
    // kernels built with kExt only: the integrator's optional steps (kIntegratorRoulette |
    // kIntegratorRegularize), 1 - Scene::RussianRoulette, and `mats` holds a second table behind
    // the first: the materials after Material::regularize()
    uint32_t integrator;
    float roulette_q;
    uint32_t has_spheres;  // some record of `tris` is a sphere (then word 15 is the mesh index everywhere, never a group mask)
    // A scene may bring the reference's own tree beside the one `nodes` / `pairs` hold (any tree over the same leaves
    // reaches the same triangles for a ray whose three reciprocal direction components are finite; a ray with a zero
    // component can fail a box test on a NaN whatever the nesting, so for it the tree matters): such rays start their
    // walk at record `ref_root` of `pairs` (the reference tree's records lie behind the others; 0 = one tree only),
    // the threaded walk of kSceneGlobal reads `ref_nodes` (= `nodes` when there is one tree only).
    uint32_t ref_root, num_ref_nodes;
    const float4* ref_nodes;
    // Materials staged in LDS by stage_scene (every variant): mat_lds_float4s float4 (both material tables) at float4
    // offset mat_lds_off of the workgroup's dynamic LDS, counted in lds_float4s.  0 = not staged (more than
    // kMaxLdsMaterials materials): such scenes run the kernels built with kExt, which read `mats`.
    uint32_t mat_lds_off, mat_lds_float4s;
    // The scene's only light when it is one mesh of two triangles (the Cornell light and every scene built on it): all
    // that Scene::sampleLight and the direct-light term read of it, in ten float4 that stage_scene puts into LDS at
    // float4 offset light_lds_off (counted in lds_float4s):
    //   [4k + 0..3], k = 0 left leaf's triangle, 1 right leaf's:  v0.xyz, MIS light pdf | v1.xyz | v2.xyz | normal.xyz
    //   [8] mesh area, root area, 1 / root area, area of the left leaf      [9] emission.xyz
    // The values are the ones `lights`, `nodes`, `node_area`, `tris` and `tri_v12` hold (same bits), so taking them from
    // here changes no result; it saves the chain of dependent fetches the general descent makes per bounce.
    // fast_light = null: the general code.  (Not kernel arguments: thirty more live scalar registers spilled.)
    const float4* fast_light;
    uint32_t light_lds_off;
};
constexpr uint32_t kFastLightFloat4s = 10;
constexpr uint32_t kMaxLdsMaterials = 16;
enum : uint32_t { kIntegratorRoulette = 1u, kIntegratorRegularize = 2u };
// word 15 of a record of the triangle array with this bit set: the record is a sphere
//   center.xyz, 0 | radius, radius^2, 0, bits(material) | 0, 0, 0, bits(prim id) | 0, 0, 0, bits(mesh | kSphereBit)
constexpr uint32_t kSphereBit = 0x80000000u;

// ------------------------------------------------------------ traversal -----
// Scene residency / traversal variant, chosen at upload from the scene size:
enum {
    kSceneGlobal = 0,  // records read through L1/L2; triangle tested when its leaf is reached
    kSceneLds = 1,     // records staged in LDS; same loop
    kSceneTiny32 = 2,  // LDS, <= 32 triangles: the box walk only *marks* the triangles of the
    kSceneTiny64 = 3,  // leaves it reaches (one bit each, 32- or 64-bit mask per lane); the
                       // triangle tests run afterwards in a loop of their own, so lanes that
                       // are at interior nodes never idle through another lane's triangle test
    kScenePair = 5,    // larger scenes: two-child records (64 B: both children's boxes), depth-first
    kScenePairLds = 6, // with a per-lane stack in LDS; one dependent fetch per two box tests.
                       // Records read through L1/L2 (5) or staged in LDS (6).  Same box tests,
                       // same leaves reached as the threaded walk, in a different order.
    kScenePairCull = 7, // opt-in, NOT the reference's leaf-visit set: the records of (5), nearer child
                       // first, boxes behind the origin or beyond the closest hit so far (shadow
                       // rays: beyond the light) skipped.  The reference tests those boxes and
                       // their triangles too and rejects the hits by t; results differ only where
                       // float rounding lets a triangle's t fall outside its own box's range by
                       // more than the slack.
};

// LDS byte address of a pointer into shared memory (the low half of its flat address), and a
// 16-byte read at such an address: the address register goes into ds_read_b128 as it is
typedef float raw_float4 __attribute__((ext_vector_type(4)));
typedef const raw_float4 __attribute__((address_space(3))) lds_float4;
HD uint32_t lds_address(const void* p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }
HD float4 lds_read(uint32_t at) {
    const raw_float4 v = *reinterpret_cast<lds_float4*>(static_cast<uintptr_t>(at));
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef uint32_t __attribute__((address_space(3))) lds_word;
HD uint32_t lds_read_word(uint32_t at) { return *reinterpret_cast<const lds_word*>(static_cast<uintptr_t>(at)); }
HD void lds_write_word(uint32_t at, uint32_t v) { *reinterpret_cast<lds_word*>(static_cast<uintptr_t>(at)) = v; }

// Stage the records a variant reads from LDS (the Cornell box: 20 KiB).  Every thread of the
// workgroup must call it; it ends with a workgroup barrier (also for the variants that stage nothing).
template <int kScene>
HD void stage_scene(const DevScene& S, float4* lds) {
    {
        const float4* src = reinterpret_cast<const float4*>(S.mats);
        for (uint32_t i = threadIdx.x; i < S.mat_lds_float4s; i += blockDim.x) lds[S.mat_lds_off + i] = src[i];
        if (S.fast_light)
            for (uint32_t i = threadIdx.x; i < kFastLightFloat4s; i += blockDim.x) lds[S.light_lds_off + i] = S.fast_light[i];
    }
    if constexpr (kScene == kSceneGlobal || kScene == kScenePair || kScene == kScenePairCull) {
        __syncthreads();
    } else if constexpr (kScene == kScenePairLds) {
        const uint32_t np = 4u * S.num_pairs, nt = 4u * S.num_tris;
        for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) lds[i] = S.pairs[i];
        for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[np + i] = S.tris[i];
        __syncthreads();
    } else if constexpr (kScene == kSceneLds) {
        const uint32_t nn = 2u * S.num_nodes, nt = 4u * S.num_tris;
        for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) lds[i] = S.nodes[i];
        for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[nn + i] = S.tris[i];
        __syncthreads();
    } else {
        // Marking variants: NINE copies of the node array, prepared at upload (build_octant_copies in
        // hpt_core.hip), nn records of 32 bytes each.  Copy k < 8 serves rays of sign octant k
        // (bit 0/1/2 = x/y/z of the direction negative) and holds, in pre-order,
        //   (near.xyz, link) (far.xyz, leaf word)
        // near/far = the slab planes a ray of that octant enters/leaves through (the dirIsNeg
        // selection of Bounds3::IntersectP, common/Bounds3.hpp:100-116, done once instead of per
        // test); subtrees whose triangles all face away from every ray of the octant are left
        // out (the reference walks them, but Triangle::getIntersection rejects each of their
        // triangles at its first line, common/Triangle.hpp:168-169).  Copy 8 is the whole tree
        // with near = pMin, far = pMax for rays with a zero direction component.  link = byte
        // offset of the miss target from the start of the copies; it becomes an LDS address here.
        // Leaf word: the triangle's bit in the pending mask (32-triangle variant; 0 = interior)
        // or the triangle index / -1 (64).
        const uint32_t nn = S.num_nodes, nt = 4u * S.num_tris;
        const uint32_t base = lds_address(lds);
        for (uint32_t i = threadIdx.x; i < 18u * nn; i += blockDim.x) {
            float4 v = S.oct_nodes[i];
            if (!(i & 1u)) v.w = __uint_as_float(__float_as_uint(v.w) + base);
            lds[i] = v;
        }
        for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[18u * nn + i] = S.tris[i];
        for (uint32_t i = threadIdx.x; i < 32u; i += blockDim.x) reinterpret_cast<uint32_t*>(lds + 18u * nn + nt)[i] = S.oct_table[i];
        __syncthreads();
    }
}

// Bounds3::IntersectP (common/Bounds3.hpp:93-118): float slab test, no t-range
// culling, every comparison false on NaN.
HD bool box_hit(const float4 lo, const float4 hi, f3 o, f3 inv, bool nx, bool ny, bool nz) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    float tMin = nx ? bx : ax, tMax = nx ? ax : bx;
    const float tyMin = ny ? by : ay, tyMax = ny ? ay : by;
    const float tzMin = nz ? bz : az, tzMax = nz ? az : bz;
    bool hit = !((tMin > tyMax) || (tyMin > tMax));
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    return hit && !((tMin > tzMax) || (tzMin > tMax));
}

// The same predicate for rays whose three reciprocal direction components are
// finite: then no product can be NaN and near <= far holds per axis, so
// near = min(a, b), far = max(a, b) equal the sign-selected values and the
// reference's sequence of comparisons reduces to max(near) <= min(far).
HD bool box_range_finite(const float4 lo, const float4 hi, f3 o, f3 inv, float& tnear, float& tfar) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tnear <= tfar;
}
HD bool box_hit_finite(const float4 lo, const float4 hi, f3 o, f3 inv) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tnear <= tfar;
}

// 1. / (double)x, correctly rounded, for a float x.  The compiler's sequence for an IEEE double
// division (v_div_scale x2, v_rcp_f64, two Newton steps, quotient, remainder, v_div_fmas,
// v_div_fixup) with the parts left out that cannot act here: a float's magnitude lies in
// [2^-149, 2^128), so neither operand nor quotient needs the rescaling v_div_scale / v_div_fmas
// exist for, and v_div_fixup only rewrites the result for zero, infinite or NaN operands -- the
// callers pass |x| >= 1e-5; an infinite or NaN x takes the plain division.
HD double reciprocal_of_float(float x) {
    const double d = (double)x;
    if (!(fabsf(x) < 1.0e30f)) return 1. / d;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);  // quotient estimate q = 1.0 * r
    return __builtin_fma(rem, r, r);
}

// solveQuadratic (common/MathUtils.hpp:86-100): float coefficients; its sqrt is the double one, and
// the sums / products around it and the double literal -0.5 are double, narrowed on assignment
HD bool solveQuadratic(float a, float b, float c, float& x0, float& x1) {
    const float discr = b * b - 4 * a * c;
    if (discr < 0) return false;
    if (discr == 0) {
        x0 = x1 = (float)(-0.5 * (double)b / (double)a);
    } else {
        const double root = __builtin_sqrt((double)discr);
        const float q = (b > 0) ? (float)(-0.5 * ((double)b + root)) : (float)(-0.5 * ((double)b - root));
        x0 = q / a;
        x1 = c / q;
    }
    if (x0 > x1) {
        const float t = x0;
        x0 = x1;
        x1 = t;
    }
    return true;
}
// Sphere::getIntersection (common/Sphere.hpp:49-69, commented out in the reference): the nearer
// root that is not negative, no back-face rule
HD bool sphere_hit(const float4* __restrict__ T, int rec, f3 o, f3 d, double& t_out) {
    const f3 center = xyz(T[4 * rec]);
    const float radius2 = T[4 * rec + 1].y;
    const f3 L = o - center;
    const float a = dot(d, d);
    const float b = 2 * dot(d, L);
    const float c = dot(L, L) - radius2;
    float t0, t1;
    if (!solveQuadratic(a, b, c, t0, t1)) return false;
    if (t0 < 0) t0 = t1;
    if (t0 < 0) return false;
    t_out = (double)t0;
    return true;
}

// Triangle::getIntersection (common/Triangle.hpp:164-198): back-face cull,
// float cross/dot products, double det, 1/det, u, v, t.
// kCulled: the caller has already dropped back-facing triangles.
template <bool kCulled = false>
HD bool tri_hit(const float4* __restrict__ T, int tri, f3 o, f3 d, double& t_out) {
    if constexpr (!kCulled) {
        const float4 n4 = T[4 * tri + 3];
        if (__float_as_uint(n4.w) & kSphereBit) return sphere_hit(T, tri, o, d, t_out);  // (never in the scenes the marking variants serve)
        const f3 n = xyz(n4);
        if (dot(d, n) > 0) return false;
    }
    const f3 v0 = xyz(T[4 * tri]), e1 = xyz(T[4 * tri + 1]), e2 = xyz(T[4 * tri + 2]);
    const f3 pvec = cross(d, e2);
    const float detf = dot(e1, pvec);
    if (fabsf(detf) < kEpsilon5) return false;
    const double det_inv = reciprocal_of_float(detf);
    const f3 tvec = o - v0;
    const double u = (double)dot(tvec, pvec) * det_inv;
    if (u < 0 || u > 1) return false;
    const f3 qvec = cross(tvec, e1);
    const double v = (double)dot(d, qvec) * det_inv;
    if (v < 0 || u + v > 1) return false;
    const double t = (double)dot(e2, qvec) * det_inv;
    if (t < 0) return false;
    t_out = t;
    return true;
}

// The same test on a record that has been read already, without early exits.  The walks that read their
// records through L1/L2 use this form: tri_hit's staged exits make four DEPENDENT fetches of one 64-byte record
// (flag word, normal, e1/e2, v0: each waits for the one before), and a wave runs the whole test anyway while any
// of its lanes goes on.  Every comparison is the reference's, in its order and with its NaN behaviour (a NaN passes
// `u < 0 || u > 1` there too); a stage that has failed only keeps later stages from mattering.
// kSpheres = false: the caller knows that the scene holds no sphere (the kernels built without kExt: a sphere makes a
// scene take the others); without the branch on the record's flag all four pieces are wanted at once.
template <bool kSpheres = true>
HD bool tri_hit_record(const float4 r0, const float4 r1, const float4 r2, const float4 r3, f3 o, f3 d, double& t_out) {
    if (kSpheres && (__float_as_uint(r3.w) & kSphereBit)) {  // Sphere::getIntersection on (center, radius^2)
        const f3 L = o - xyz(r0);
        const float a = dot(d, d);
        const float b = 2 * dot(d, L);
        const float c = dot(L, L) - r1.y;
        float t0, t1;
        if (!solveQuadratic(a, b, c, t0, t1)) return false;
        if (t0 < 0) t0 = t1;
        if (t0 < 0) return false;
        t_out = (double)t0;
        return true;
    }
    const f3 v0 = xyz(r0), e1 = xyz(r1), e2 = xyz(r2), n = xyz(r3);
    bool ok = !(dot(d, n) > 0);
    const f3 pvec = cross(d, e2);
    const float detf = dot(e1, pvec);
    ok = ok && !(fabsf(detf) < kEpsilon5);
    const double det_inv = reciprocal_of_float(ok ? detf : 1.0f);
    const f3 tvec = o - v0;
    const double u = (double)dot(tvec, pvec) * det_inv;
    ok = ok && !(u < 0 || u > 1);
    const f3 qvec = cross(tvec, e1);
    const double v = (double)dot(d, qvec) * det_inv;
    ok = ok && !(v < 0 || u + v > 1);
    const double t = (double)dot(e2, qvec) * det_inv;
    ok = ok && !(t < 0);
    t_out = t;
    return ok;
}

// ---- the same test with the double division taken out of the common case ------------------
// Triangle::getIntersection decides with u = un * (1/det), v = vn * (1/det), t = tn * (1/det)
// (un, vn, tn, det: the four float dot products; the products and 1/det in double).  For
// 1e-5 <= det < 1e30 the double 1/det is positive and finite and no product can underflow, so
//   un < 0 => u < 0,  vn < 0 => v < 0,  tn < 0 => t < 0           (signs are exact; -0 is not < 0)
//   un + vn > det * (1 + 2e-6)  (float sum of two non-negative terms, relative error 6e-8)
//                               => u + v > 1 in double             (relative error 1e-15)
//   un, vn, tn >= 0 and un + vn <= det * (1 - 2e-6) => u <= u + v <= 1: accepted
// decide all but a sliver of the cases without the division; everything else (the sliver, a
// negative, huge or NaN det, NaN numerators: every comparison above is false for them) takes the
// reference's sequence on the same four floats.
struct TriNum {
    float det, un, vn, tn;
};
constexpr float kTriUp = 1.000002f, kTriDown = 0.999998f, kTriAbs = 1.0e-30f;

HD TriNum tri_numerators(const float4* __restrict__ T, int tri, f3 o, f3 d) {
    const f3 v0 = xyz(T[4 * tri]), e1 = xyz(T[4 * tri + 1]), e2 = xyz(T[4 * tri + 2]);
    const f3 pvec = cross(d, e2);
    TriNum n;
    n.det = dot(e1, pvec);
    const f3 tvec = o - v0;
    n.un = dot(tvec, pvec);
    const f3 qvec = cross(tvec, e1);
    n.vn = dot(d, qvec);
    n.tn = dot(e2, qvec);
    return n;
}
// tri_hit<true>'s verdict from the four floats.  `ta`: tn / det to within 3e-7 relative + 1.2e-38
// absolute when the fast rules decided (+inf when the quotient exceeds the float range: farther than
// any hit or limit it is compared with), NaN otherwise (a NaN makes every later comparison fall
// through to the exact one).  Written without early exits: in a wave some lane nearly always
// needs the whole test (staged exits measured 4 % slower).
HD bool tri_classify(const TriNum& n, float& ta) {
    const bool regular = n.det >= kEpsilon5 && n.det < 1.0e30f;
    const float s = n.un + n.vn;
    const float lo = fminf(fminf(n.un, n.vn), n.tn);
    const bool sure_miss = regular && (lo < 0.f || s > n.det * kTriUp);
    bool hit = regular && lo >= 0.f && s <= n.det * kTriDown;
    ta = n.tn * __builtin_amdgcn_rcpf(n.det);  // 1 ulp + 0.5 ulp; det < 1e30 keeps the reciprocal normal
    if (!(sure_miss || hit)) {
        ta = __builtin_nanf("");
        if (!(fabsf(n.det) < kEpsilon5)) {
            const double det_inv = reciprocal_of_float(n.det);
            const double u = (double)n.un * det_inv, v = (double)n.vn * det_inv, t = (double)n.tn * det_inv;
            hit = !(u < 0 || u > 1) && !(v < 0 || u + v > 1) && !(t < 0);
        }
    }
    return hit;
}
// the reference's t for an accepted triangle, from its two floats
HD double tri_exact_t(float tn, float det) { return (double)tn * reciprocal_of_float(det); }

// Closest hit so far as (triangle, its tn and det): its exact t is only computed when a query
// ends or when two candidates are too close to order by their float quotients.  The margins
// (2e-6 relative, 1e-30 absolute) exceed the error of both quotients together, so "clearly
// smaller" here implies strictly smaller in the reference's doubles.
struct Nearest {
    int tri;
    float ta, tn, det;
};
HD Nearest nearest_none() { return Nearest{-1, kInfinity, 0.f, 1.f}; }  // "no hit" is t = kInfinity, as in trace<>
// kTie: among equal t the lower index wins (walks that do not test in the reference's order)
template <bool kTie>
HD void nearest_update(Nearest& b, int tri, const TriNum& n, float ta) {
    bool take = ta * kTriUp + kTriAbs < b.ta;
    if (!take && !(ta > b.ta * kTriUp + kTriAbs)) {
        const double tc = tri_exact_t(n.tn, n.det);
        const double tb = b.tri < 0 ? (double)kInfinity : tri_exact_t(b.tn, b.det);
        take = tc < tb || (kTie && tc == tb && tri < b.tri);
        if (take && !(ta == ta)) ta = (float)tc;
    }
    if (take) {
        b.tri = tri;
        b.ta = ta;
        b.tn = n.tn;
        b.det = n.det;
    }
}
HD double nearest_t(const Nearest& b) { return b.tri < 0 ? (double)kInfinity : tri_exact_t(b.tn, b.det); }
// any-hit rule t < limit (BVH.hpp:121-127) for an accepted triangle; limit_f = (float)limit.
// 1 = occludes, 0 = does not, -1 = too close to tell from the float quotient (the sampled light
// triangle itself always is: its t lies within 1e-4 of the light's distance)
HD int tri_occludes_fast(float ta, float limit_f) {
    if (ta * kTriUp + kTriAbs < limit_f) return 1;
    if (ta > limit_f * kTriUp + kTriAbs) return 0;
    return -1;
}

// BVHAccel::getIntersection (common/BVH.hpp:91-145) over the flattened tree:
//   node visited after a box hit  = i + 1        (nextIfHit  = pre-order successor, BVH.cpp:53-62)
//   node visited after a box miss = miss link    (nextIfMiss, BVH.cpp:22-37)
// Closest hit keeps the strict `<` update in visit order (BVH.hpp:130);
// triangles are stored in visit order and the deferred variants test marked
// triangles in ascending index, so ties resolve exactly as in the reference.
// kShadow: any-hit with t < tmax - Epsilon4 (BVH.hpp:121-127).  Which triangles
// are tested is decided by the box tests alone, so deferring the tests changes
// no result; an occluded shadow ray merely finishes its box walk first.

// Per-lane queue of leaf triangles whose test is still to be run (all variants but the marking
// ones): tri_queue_entries(variant) entries in LDS, entry k of a lane at q[k * stride].
constexpr uint32_t kTriQueue = 8;
// The depth-first variants run longer between two rounds of triangle tests (each round is as
// long as the lane with the most candidates needs): 16 entries measured best for the reference
// walk (+13 % on the bunny over 8, 24 is slower again: LDS per wave), 12 for the culled one,
// which wants its hits early.
#ifndef HPT_PAIR_QUEUE
#define HPT_PAIR_QUEUE 16
#endif
__host__ __device__ constexpr uint32_t tri_queue_entries(int scene) {
    return (scene == kScenePair || scene == kScenePairLds) ? (uint32_t)HPT_PAIR_QUEUE : scene == kScenePairCull ? 12u : kTriQueue;
}
struct TriQueue {
    uint32_t* q;
    uint32_t stride;  // entries are addressed as q[__umul24(k, stride)]: both factors are small, and the 24-bit multiply is a full-rate instruction (the 32-bit one is quarter rate)
    uint32_t* stk;  // traversal stack of the pair variants: entry k of a lane at stk[k * stride]
    uint32_t stage; // cooperative record fetch (coop_fetch below): LDS byte address of this wave's staging area
};
// Cooperative record fetch (scenes whose records do not fit the L2s).  A lane that fetches its own 64-byte record
// issues four 16-byte loads to one cache line; a wave does that for 64 different lines, and the three loads that
// follow a line's first find its miss still in flight.  Here the four lanes of a quad fetch the four pieces of ONE
// record per instruction (one 64-byte access per quad), straight into LDS (the load's LDS address is wave base +
// lane * 16), in four instructions for the records of quad-lanes 0..3; each lane then reads its record back.
// Piece k of the wave lies at byte kCoopPieceBytes * k: 1024 + 16, so that the lanes of a read hit different banks.
#ifndef HPT_COOP
#define HPT_COOP 0
#endif
constexpr uint32_t kCoopPieceBytes = 1040u, kCoopWaveBytes = 4u * kCoopPieceBytes;
__host__ __device__ constexpr bool coop_variant(int scene) { return HPT_COOP != 0 && scene == 5; }
// the per-lane LDS scratch of a workgroup starts at `base`: (staging areas of all waves,) queues of all lanes, then stacks
template <int kScene>
HD TriQueue make_tri_queue(uint32_t* base) {
    uint32_t stage = 0u;
    if constexpr (coop_variant(kScene)) {
        stage = lds_address(base) + (threadIdx.x >> 6) * kCoopWaveBytes;
        base += ((blockDim.x + 63u) >> 6) * (kCoopWaveBytes / 4u);
    }
    return TriQueue{base + threadIdx.x, blockDim.x, base + tri_queue_entries(kScene) * blockDim.x + threadIdx.x, stage};
}

template <int kScene, bool kShadow>
HD void trace(const DevScene& S, const float4* __restrict__ lds, TriQueue tq, f3 o, f3 d, double tmax, int& hit_tri, double& hit_t PROF_ARG COUNT_ARG) {
    const float4* N;
    const float4* T;
    if constexpr (kScene == kScenePairLds) {
        N = lds;  // pair records
        T = lds + 4u * S.num_pairs;
    } else if constexpr (kScene == kScenePair || kScene == kScenePairCull) {
        N = S.pairs;
        T = S.tris;
    } else if constexpr (kScene == kSceneLds) {
        N = lds;
        T = lds + 2u * S.num_nodes;
    } else if constexpr (kScene != kSceneGlobal) {
        N = lds;  // nine copies of the nodes, see stage_scene
        T = lds + 18u * S.num_nodes;
    } else {
        N = S.nodes;
        T = S.tris;
    }
    const f3 inv = mk(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));  // == float(1. / double(d)), Ray.hpp:21
    const bool nx = d.x < 0, ny = d.y < 0, nz = d.z < 0;
    const double limit = tmax - (double)kEpsilon4;
    int best = -1;
    double best_t = (double)kInfinity;
    int node = 0;
    if constexpr (kScene == kSceneTiny32 || kScene == kSceneTiny64) {
        constexpr bool kMask32 = (kScene != kSceneTiny64);
        using mask_t = typename std::conditional<kMask32, uint32_t, unsigned long long>::type;
        mask_t pending = 0;
        const float kInf = __builtin_inff();
        PROF_T0();
        const bool finite = fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf;
        const uint32_t nbase = lds_address(N), copy_bytes = S.num_nodes * 32u;
        const uint32_t* table = reinterpret_cast<const uint32_t*>(T + 4u * S.num_tris);
        mask_t undecided;  // marked triangles whose back-face test (Triangle.hpp:168-169) is still to be done
        if (finite) {
            // walk the copy of this ray's octant; `at` is the node's LDS address, links are addresses too
            const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
            uint32_t at = nbase + __umul24(oct, copy_bytes);  // (the 32-bit integer multiply is a quarter-rate instruction)
            const uint32_t stop = at + table[oct] * 32u;
#ifdef HPT_BOX_TWICE  // diagnostic: the walk's cost = the time this adds
            for (uint32_t at2 = at; at2 < stop;) {
                const float4 nr = lds_read(at2), fr = lds_read(at2 + 16u);
                const float tnear = fmaxf(fmaxf((nr.x - o.x) * inv.x, (nr.y - o.y) * inv.y), (nr.z - o.z) * inv.z);
                const float tfar = fminf(fminf((fr.x - o.x) * inv.x, (fr.y - o.y) * inv.y), (fr.z - o.z) * inv.z);
                const bool hit = tnear <= tfar;
                pending |= hit ? (mask_t)__float_as_uint(fr.w) : (mask_t)0;
                at2 = hit ? at2 + 32u : __float_as_uint(nr.w);
                asm volatile("" ::: "memory");
            }
#endif
            while (at < stop) {
                PROF_ITER(kShadow ? kProfBoxS : kProfBoxC);
                COUNT_BOX(1);
                const float4 nr = lds_read(at), fr = lds_read(at + 16u);
                // Bounds3::IntersectP with the planes already ordered; equals box_hit for finite reciprocals
                const float tnear = fmaxf(fmaxf((nr.x - o.x) * inv.x, (nr.y - o.y) * inv.y), (nr.z - o.z) * inv.z);
                const float tfar = fminf(fminf((fr.x - o.x) * inv.x, (fr.y - o.y) * inv.y), (fr.z - o.z) * inv.z);
                const bool hit = tnear <= tfar;
                if constexpr (kMask32) {
                    pending |= hit ? __float_as_uint(fr.w) : 0u;  // staged as a leaf bit
                } else {
                    const int tri = __float_as_int(fr.w);
                    if (hit && tri >= 0) pending |= (mask_t)1 << tri;
                }
                at = hit ? at + 32u : __float_as_uint(nr.w);
            }
            if constexpr (kMask32) undecided = pending & table[16u + 2u * oct];
            else undecided = pending & ((mask_t)table[16u + 2u * oct] | ((mask_t)table[17u + 2u * oct] << 32));
        } else {  // a direction component is 0 (or denormal): the reference's NaN-aware sequence on the whole tree
            uint32_t at = nbase + 8u * copy_bytes;
            const uint32_t stop = at + copy_bytes;
            while (at < stop) {
                COUNT_BOX(1);
                const float4 lo = lds_read(at), hi = lds_read(at + 16u);
                const bool hit = box_hit(lo, hi, o, inv, nx, ny, nz);
                if constexpr (kMask32) {
                    pending |= hit ? __float_as_uint(hi.w) : 0u;
                } else {
                    const int tri = __float_as_int(hi.w);
                    if (hit && tri >= 0) pending |= (mask_t)1 << tri;
                }
                at = hit ? at + 32u : __float_as_uint(lo.w);
            }
            undecided = pending;
        }
        PROF_T1(kShadow ? kProfBoxS : kProfBoxC);
        // cheap filter first: drop back-facing triangles, so that the expensive rounds below are
        // only as many as the busiest lane needs.  Most triangles are decided by the octant alone.
        if constexpr (kMask32) {
            // word 3 of the normal's float4 = mask of the triangles with the same normal (set at
            // upload): one test settles them all (the two triangles of a rectangle)
            for (uint32_t m = undecided; m;) {
                const int tri = __ffs((int)m) - 1;
                const float4 nr = T[4 * tri + 3];
                const uint32_t group = __float_as_uint(nr.w);
                COUNT_TRI(dot(d, xyz(nr)) > 0 ? 1 : 0);  // a triangle rejected here never reaches the loop below
                if (dot(d, xyz(nr)) > 0) pending &= ~group;
                m &= ~group;
            }
        } else {
            for (mask_t m = undecided; m;) {
                const int tri = __ffsll((long long)m) - 1;
                const mask_t rest = m & (m - 1);
                COUNT_TRI(dot(d, xyz(T[4 * tri + 3])) > 0 ? 1 : 0);
                if (dot(d, xyz(T[4 * tri + 3])) > 0) pending ^= m ^ rest;  // m ^ rest = the bit just examined
                m = rest;
            }
        }
        Nearest nb = nearest_none();
        const float limit_f = (float)limit;
        // any-hit: the first accepted triangle whose t is too close to the limit for the float
        // quotient waits here; its division runs once, after the loop (a second one is settled at once)
        int wait_tri = -1;
        float wait_tn = 0.f, wait_det = 1.f;
        while (pending) {
            PROF_ITER(kShadow ? kProfTriS : kProfTriC);
            COUNT_TRI(1);
            const int tri = kMask32 ? __ffs((int)pending) - 1 : __ffsll((long long)pending) - 1;
            pending &= pending - 1;
            const TriNum n = tri_numerators(T, tri, o, d);
            float ta;
            if (tri_classify(n, ta)) {
                if constexpr (kShadow) {
                    int occ = tri_occludes_fast(ta, limit_f);
                    if (occ < 0) {
                        if (wait_tri < 0) {
                            wait_tri = tri;
                            wait_tn = n.tn;
                            wait_det = n.det;
                            occ = 0;
                        } else {
                            occ = tri_exact_t(n.tn, n.det) < limit ? 1 : 0;
                        }
                    }
                    if (occ > 0) {
                        nb.tri = tri;
                        nb.tn = n.tn;
                        nb.det = n.det;
                        pending = 0;
                    }
                } else {
                    nearest_update<false>(nb, tri, n, ta);  // ascending index = the reference's order: strict <
                }
            }
        }
        if constexpr (kShadow) {
            // the waiting triangle precedes whatever ended the loop: it is the reference's first occluder if it passes
            if (wait_tri >= 0 && tri_exact_t(wait_tn, wait_det) < limit) {
                nb.tri = wait_tri;
                nb.tn = wait_tn;
                nb.det = wait_det;
            }
        }
        best = nb.tri;
        best_t = nearest_t(nb);
        PROF_T1(kShadow ? kProfTriS : kProfTriC);  // filter + tests (includes the box time, subtracted by the reader)
    } else if constexpr (kScene == kScenePair || kScene == kScenePairLds || kScene == kScenePairCull) {
        // Depth-first over two-child records.  A record is visited iff its node's box was hit,
        // and both children's boxes are tested there: the same tests as the threaded walk makes
        // (it tests a node iff its parent was hit).  Leaves reached are queued and tested in a
        // loop of their own; the visit order differs from the reference's, so the closest-hit
        // update spells out its tie rule: among equal t the triangle visited first by the
        // reference (= lowest index, triangles are stored in its visit order) wins.
        // kCull (see kScenePairCull): a child also has to lie in front of the origin and not
        // beyond `cull_t`; of two interior children the nearer is walked first and the other
        // waits on the stack with its entry distance (10 bits, rounded down), to be dropped
        // when a closer hit has been found by then.
        constexpr bool kCull = (kScene == kScenePairCull);
        constexpr uint32_t kQueue = tri_queue_entries(kScene);
        const float kInf = __builtin_inff();
        const bool finite = fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf;
        const bool cull = kCull && finite;
        const float behind = -S.cull_slack;
        float cull_t = kShadow ? (float)limit * 1.000001f + S.cull_slack : kInf;
        const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.f);
        const float4 rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.f);
        int cur;
        if (cull) {
            float tn, tf;
            cur = (box_range_finite(rlo, rhi, o, inv, tn, tf) && tf >= behind && tn <= cull_t) ? 0 : -1;
        } else {
            // (a ray with a zero direction component walks the reference's own tree: DevScene::ref_root)
            cur = finite ? (box_hit_finite(rlo, rhi, o, inv) ? 0 : -1) : (box_hit(rlo, rhi, o, inv, nx, ny, nz) ? (int)S.ref_root : -1);
        }
        uint32_t cnt = 0, sp = 0;
        COUNT_BOX(1);  // the root's box
        for (;;) {
            while (cur >= 0 && cnt + 2u <= kQueue) {
                COUNT_BOX(2);
                const float4 q0 = N[4 * cur], q1 = N[4 * cur + 1], q2 = N[4 * cur + 2], q3 = N[4 * cur + 3];
                bool hl, hr;
                float tnl = 0.f, tnr = 0.f;
                if (cull) {
                    float tfl, tfr;
                    hl = box_range_finite(q0, q1, o, inv, tnl, tfl) && tfl >= behind && tnl <= cull_t;
                    hr = box_range_finite(q2, q3, o, inv, tnr, tfr) && tfr >= behind && tnr <= cull_t;
                } else if (finite) {
                    hl = box_hit_finite(q0, q1, o, inv);
                    hr = box_hit_finite(q2, q3, o, inv);
                } else {
                    hl = box_hit(q0, q1, o, inv, nx, ny, nz);
                    hr = box_hit(q2, q3, o, inv, nx, ny, nz);
                }
                const int rl = __float_as_int(q0.w), rr = __float_as_int(q1.w);
                if (hl && rl < 0) {
                    tq.q[__umul24(cnt, tq.stride)] = (uint32_t)~rl;
                    ++cnt;
                }
                if (hr && rr < 0) {
                    tq.q[__umul24(cnt, tq.stride)] = (uint32_t)~rr;
                    ++cnt;
                }
                const bool gl = hl && rl >= 0, gr = hr && rr >= 0;
                if (gl && gr) {
                    if constexpr (kCull) {
                        const bool right_first = tnr < tnl;
                        const float tfar_child = right_first ? tnl : tnr;
                        const uint32_t qd = (uint32_t)fminf(fmaxf(tfar_child, 0.f) * S.cull_qscale, 1023.f);
                        tq.stk[__umul24(sp, tq.stride)] = (uint32_t)(right_first ? rl : rr) | (qd << 22);
                        cur = right_first ? rr : rl;
                    } else {
                        tq.stk[__umul24(sp, tq.stride)] = (uint32_t)rr;
                        cur = rl;
                    }
                    ++sp;
                } else if (gl) {
                    cur = rl;
                } else if (gr) {
                    cur = rr;
                } else {
                    cur = -1;
                    while (sp) {
                        --sp;
                        uint32_t e = tq.stk[__umul24(sp, tq.stride)];
                        if constexpr (kCull) {
                            if ((float)(e >> 22) * S.cull_qinv > cull_t) continue;
                            e &= 0x3FFFFFu;
                        }
                        cur = (int)e;
                        break;
                    }
                }
            }
            if (cnt == 0) break;
            for (uint32_t k = 0; k < cnt; ++k) {
                const int tri = (int)tq.q[__umul24(k, tq.stride)];
                double t;
                COUNT_TRI(1);
                const float4 r0 = T[4 * tri], r1 = T[4 * tri + 1], r2 = T[4 * tri + 2], r3 = T[4 * tri + 3];  // one fetch of the whole record
                if (tri_hit_record(r0, r1, r2, r3, o, d, t)) {
                    if constexpr (kShadow) {
                        if (t < limit) {
                            best = tri;
                            best_t = t;
                            cur = -1;
                            break;
                        }
                    } else {
                        if (t < best_t || (t == best_t && tri < best)) {
                            best = tri;
                            best_t = t;
                            if constexpr (kCull) cull_t = (float)t * 1.000001f + S.cull_slack;
                        }
                    }
                }
            }
            cnt = 0;
            if (cur < 0) break;
        }
    } else {
        // Threaded walk with queued triangle tests (kept for trees whose depth exceeds the stack
        // budget, and for comparison): the box walk queues the triangles of the leaves it reaches
        // (up to kTriQueue per lane) and the tests run in a loop of their own; among equal
        // distances the lowest record index wins, which is the reference's first visited.
        // A lane whose queue is full waits for the flush.
        const float kInf = __builtin_inff();
        const bool finite = fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf;
        uint32_t cnt = 0;
        int end = (int)S.num_nodes;
        if constexpr (kScene == kSceneGlobal) {
            if (!finite) {  // a ray with a zero direction component walks the reference's own tree
                N = S.ref_nodes;
                end = (int)S.num_ref_nodes;
            }
        }
        for (;;) {
            while (node < end && cnt < kTriQueue) {
                COUNT_BOX(1);
                const float4 lo = N[2 * node], hi = N[2 * node + 1];
                const bool hit = finite ? box_hit_finite(lo, hi, o, inv) : box_hit(lo, hi, o, inv, nx, ny, nz);
                const int tri = __float_as_int(hi.w);
                if (hit && tri >= 0) {
                    tq.q[__umul24(cnt, tq.stride)] = (uint32_t)tri;
                    ++cnt;
                }
                node = hit ? node + 1 : __float_as_int(lo.w);
            }
            if (cnt == 0) break;
            for (uint32_t k = 0; k < cnt; ++k) {
                const int tri = (int)tq.q[__umul24(k, tq.stride)];
                double t;
                COUNT_TRI(1);
                const float4 r0 = T[4 * tri], r1 = T[4 * tri + 1], r2 = T[4 * tri + 2], r3 = T[4 * tri + 3];  // one fetch of the whole record
                if (tri_hit_record(r0, r1, r2, r3, o, d, t)) {
                    if constexpr (kShadow) {
                        if (t < limit) {
                            best = tri;
                            best_t = t;
                            node = end;
                            break;
                        }
                    } else {
                        // visit order is record order in the reference's tree; another tree over the same records
                        // (the host's surface-area build) meets them in another order, so the tie goes by index
                        if (t < best_t || (t == best_t && tri < best)) {
                            best = tri;
                            best_t = t;
                        }
                    }
                }
            }
            cnt = 0;
            if (node >= end) break;
        }
    }
    hit_tri = best;
    hit_t = best_t;
}

// ---- resumable traversal of the two-child records ---------------------------
// The depth-first walk of trace<kScenePair...> cut into pieces a wave scheduler can interleave
// between lanes: walk_begin (root test), walk_step (one record: two box tests), walk_test_one
// (one queued triangle).  Same box tests, same triangles tested, same closest-hit rule (lowest
// t, then lowest leaf-order index) as trace<>; only the order of the triangle tests differs,
// which no closest-hit result and no occlusion answer depends on.  Closest and any-hit rays
// share the code (`shadow` is per lane), so one wave can advance both kinds at once.
struct Walk {
    f3 o, d, inv;
    int cur;            // record to visit next, -1 = none left
    // Top of the lane's stack and end of its triangle queue as LDS byte addresses (the lane's entries lie
    // 4 * blockDim.x bytes apart, from tq.stk / tq.q on): a push is a write and one addition, where an entry count
    // would want a multiplication and a shift as well.  Empty: sp == address of tq.stk, qe == address of tq.q.
    uint32_t sp, qe;
    int best;           // closest: best triangle so far; any-hit: the occluder found
    double best_t;
    float tmax;         // any-hit: Ray::shadowDistance (a float's value, Scene.hpp:203)
    float cull_t;       // kScenePairCull only
    bool finite, shadow;
    // closest-hit walks of scenes without spheres: normal and material index of `best`, kept from its record when the
    // test accepted it, so that the shading need not fetch the record again
    f3 best_n;
    uint32_t best_m;
};
// (a step may queue two triangles)
template <int kScene>
HD bool walk_queue_has_room(const Walk& W, TriQueue tq) { return W.qe - lds_address(tq.q) <= (tri_queue_entries(kScene) - 2u) * 4u * tq.stride; }
HD bool walk_queue_empty(const Walk& W, TriQueue tq) { return W.qe == lds_address(tq.q); }
template <int kScene>
HD bool walk_can_step(const Walk& W, TriQueue tq) { return W.cur >= 0 && walk_queue_has_room<kScene>(W, tq); }
HD bool walk_done(const Walk& W, TriQueue tq) { return W.cur < 0 && walk_queue_empty(W, tq); }

template <int kScene>
HD void walk_begin(const DevScene& S, TriQueue tq, Walk& W, f3 o, f3 d, bool shadow, float tmax COUNT_ARG) {
    constexpr bool kCull = (kScene == kScenePairCull);
    W.o = o;
    W.d = d;
    W.inv = mk(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));  // == float(1. / double(d)), Ray.hpp:21
    const float kInf = __builtin_inff();
    W.finite = fabsf(W.inv.x) < kInf && fabsf(W.inv.y) < kInf && fabsf(W.inv.z) < kInf;
    W.shadow = shadow;
    W.tmax = tmax;
    W.best = -1;
    W.best_t = (double)kInfinity;
    W.sp = lds_address(tq.stk);
    W.qe = lds_address(tq.q);
    W.cull_t = shadow ? (float)((double)tmax - (double)kEpsilon4) * 1.000001f + S.cull_slack : kInf;
    const float4 rlo = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.f);
    const float4 rhi = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.f);
    COUNT_BOX(1);
    bool hit;
    if (kCull && W.finite) {
        float tn, tf;
        hit = box_range_finite(rlo, rhi, o, W.inv, tn, tf) && tf >= -S.cull_slack && tn <= W.cull_t;
    } else {
        hit = W.finite ? box_hit_finite(rlo, rhi, o, W.inv) : box_hit(rlo, rhi, o, W.inv, d.x < 0, d.y < 0, d.z < 0);
    }
    W.cur = hit ? (W.finite ? 0 : (int)S.ref_root) : -1;  // a ray with a zero direction component walks the reference's own tree
}

// visit record W.cur (requires walk_can_step): test both children's boxes, queue the leaves that
// are hit, descend into / push the interior children that are hit, else pop
template <int kScene>
HD void walk_step_record(const DevScene& S, const float4 q0, const float4 q1, const float4 q2, const float4 q3, TriQueue tq, Walk& W COUNT_ARG);
template <int kScene>
HD void walk_step(const DevScene& S, const float4* __restrict__ N, TriQueue tq, Walk& W COUNT_ARG) {
    const float4 q0 = N[4 * W.cur], q1 = N[4 * W.cur + 1], q2 = N[4 * W.cur + 2], q3 = N[4 * W.cur + 3];
    walk_step_record<kScene>(S, q0, q1, q2, q3, tq, W COUNT_PASS);
}
// Wave-level half of a cooperative step: every lane of the wave calls it (lanes that do not step pass record -1).
HD int quad_lane_value(int v, int k) {
    switch (k) {  // DPP quad_perm broadcast of quad-lane k; a lane that does not exist reads as -1
    case 0: return __builtin_amdgcn_update_dpp(-1, v, 0x00, 0xF, 0xF, false);
    case 1: return __builtin_amdgcn_update_dpp(-1, v, 0x55, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(-1, v, 0xAA, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(-1, v, 0xFF, 0xF, 0xF, false);
    }
}
HD void coop_fetch(const float4* __restrict__ N, TriQueue tq, int record) {
    const uint32_t piece = threadIdx.x & 3u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = quad_lane_value(record, k);
        if (r >= 0) {
            const float4* src = N + 4 * (size_t)r + piece;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(tq.stage + (uint32_t)k * kCoopPieceBytes), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// Lane-level half: the record this lane asked for, read back from the staging area
template <int kScene>
HD void walk_step_staged(const DevScene& S, TriQueue tq, Walk& W COUNT_ARG) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t at = tq.stage + (lane & 3u) * kCoopPieceBytes + (lane & ~3u) * 16u;
    const float4 q0 = lds_read(at), q1 = lds_read(at + 16u), q2 = lds_read(at + 32u), q3 = lds_read(at + 48u);
    walk_step_record<kScene>(S, q0, q1, q2, q3, tq, W COUNT_PASS);
}
template <int kScene>
HD void walk_step_record(const DevScene& S, const float4 q0, const float4 q1, const float4 q2, const float4 q3, TriQueue tq, Walk& W COUNT_ARG) {
    constexpr bool kCull = (kScene == kScenePairCull);
    const bool cull = kCull && W.finite;
    const float behind = -S.cull_slack;
    const f3 o = W.o, inv = W.inv;
    COUNT_BOX(2);
    bool hl, hr;
    float tnl = 0.f, tnr = 0.f;
    if (cull) {
        float tfl, tfr;
        hl = box_range_finite(q0, q1, o, inv, tnl, tfl) && tfl >= behind && tnl <= W.cull_t;
        hr = box_range_finite(q2, q3, o, inv, tnr, tfr) && tfr >= behind && tnr <= W.cull_t;
    } else {
        // The min/max form for every lane, in one block with its products: were it the branch of an `if (W.finite)`, the
        // products would be formed ahead of the branch and the compiler, which then cannot see where the operands of
        // the min/max come from, would put six canonicalising `v_max x, x, x` in front of them in every step.  The rare
        // lane with a zero direction component replaces the answer by the reference's NaN-aware sequence.
        hl = box_hit_finite(q0, q1, o, inv);
        hr = box_hit_finite(q2, q3, o, inv);
        if (!W.finite) {
            const bool nx = W.d.x < 0, ny = W.d.y < 0, nz = W.d.z < 0;
            hl = box_hit(q0, q1, o, inv, nx, ny, nz);
            hr = box_hit(q2, q3, o, inv, nx, ny, nz);
        }
    }
    const int rl = __float_as_int(q0.w), rr = __float_as_int(q1.w);
    const uint32_t pitch = 4u * tq.stride;
    if (hl && rl < 0) {
        lds_write_word(W.qe, (uint32_t)~rl);
        W.qe += pitch;
    }
    if (hr && rr < 0) {
        lds_write_word(W.qe, (uint32_t)~rr);
        W.qe += pitch;
    }
    const bool gl = hl && rl >= 0, gr = hr && rr >= 0;
    if (gl && gr) {
        if constexpr (kCull) {
            const bool right_first = tnr < tnl;
            const float tfar_child = right_first ? tnl : tnr;
            const uint32_t qd = (uint32_t)fminf(fmaxf(tfar_child, 0.f) * S.cull_qscale, 1023.f);
            lds_write_word(W.sp, (uint32_t)(right_first ? rl : rr) | (qd << 22));
            W.cur = right_first ? rr : rl;
        } else {
            lds_write_word(W.sp, (uint32_t)rr);
            W.cur = rl;
        }
        W.sp += pitch;
    } else if (gl) {
        W.cur = rl;
    } else if (gr) {
        W.cur = rr;
    } else {
        W.cur = -1;
        while (W.sp != lds_address(tq.stk)) {
            W.sp -= pitch;
            uint32_t e = lds_read_word(W.sp);
            if constexpr (kCull) {
                if ((float)(e >> 22) * S.cull_qinv > W.cull_t) continue;
                e &= 0x3FFFFFu;
            }
            W.cur = (int)e;
            break;
        }
    }
}

// test the triangle queued last (requires a queue that is not empty)
template <int kScene, bool kSpheres = true>
HD void walk_test_one(const DevScene& S, const float4* __restrict__ T, TriQueue tq, Walk& W COUNT_ARG) {
    constexpr bool kCull = (kScene == kScenePairCull);
    W.qe -= 4u * tq.stride;
    const int tri = (int)lds_read_word(W.qe);
    double t;
    COUNT_TRI(1);
    const float4 r0 = T[4 * tri], r1 = T[4 * tri + 1], r2 = T[4 * tri + 2], r3 = T[4 * tri + 3];  // one fetch of the whole record
    if (tri_hit_record<kSpheres>(r0, r1, r2, r3, W.o, W.d, t)) {
        if (W.shadow) {
            if (t < (double)W.tmax - (double)kEpsilon4) {  // BVH.hpp:121-127: occluded, the query ends
                W.best = tri;
                W.best_t = t;
                W.cur = -1;
                W.sp = lds_address(tq.stk);
                W.qe = lds_address(tq.q);
            }
        } else if (t < W.best_t || (t == W.best_t && tri < W.best)) {
            W.best = tri;
            W.best_t = t;
            if constexpr (!kSpheres) {
                W.best_n = xyz(r3);
                W.best_m = __float_as_uint(r1.w);
            }
            if constexpr (kCull) W.cull_t = (float)t * 1.000001f + S.cull_slack;
        }
    }
}

// ---- resumable walk of the marking variants ---------------------------------
// The box walk over the ray's octant copy (one node per step), the back-face filter when the walk
// ends, then one triangle test per round in ascending index (so the strict `<` of the closest-hit
// update keeps the reference's tie order).  Same tests as trace<kSceneTiny32/64>.
template <bool kMask32>
struct MarkWalk {
    using mask_t = typename std::conditional<kMask32, uint32_t, unsigned long long>::type;
    f3 o, d, inv;
    uint32_t at, stop;  // LDS byte addresses of the next node and of the end of the copy
    mask_t pending;     // marked triangles; after the filter: triangles still to be tested
    int best;
    double best_t;
    float tmax;
    bool finite, shadow, filtered;
};

// One interface over the resumable walks, so that the wave scheduler is written once:
//   begin      start a closest-hit or any-hit query
//   can_step   the lane has a box step to make          step      make it
//   wants_test the lane cannot step, triangles wait      test_one  test one of them
//   may_test   the lane has a triangle it could test now (a superset of wants_test)
//   settle     bookkeeping after a round of steps (the marking variants' back-face filter)
//   done       the query has its answer in best / best_t
template <int kScene>
struct Traversal;

template <int kScene>
struct PairTraversal {
    using State = Walk;
    static constexpr bool kKeepsHitData = true;  // (for test_one<false>) State::best_n / best_m describe State::best
    HD static void init(State& W, TriQueue tq) {
        W.cur = -1;
        W.qe = lds_address(tq.q);
    }
    HD static const float4* records(const DevScene& S, const float4* lds) { return kScene == kScenePairLds ? lds : S.pairs; }
    HD static const float4* triangles(const DevScene& S, const float4* lds) { return kScene == kScenePairLds ? lds + 4u * S.num_pairs : S.tris; }
    HD static void begin(const DevScene& S, const float4* lds, TriQueue tq, State& W, f3 o, f3 d, bool shadow, float tmax COUNT_ARG) {
        walk_begin<kScene>(S, tq, W, o, d, shadow, tmax COUNT_PASS);
    }
    HD static bool can_step(const State& W, TriQueue tq) { return walk_can_step<kScene>(W, tq); }
    HD static bool wants_test(const State& W, TriQueue tq) { return !walk_can_step<kScene>(W, tq) && !walk_queue_empty(W, tq); }
    // the same two as lane masks (every lane of the wave calls them)
    HD static unsigned long long can_step_mask(const State& W, TriQueue tq) { return lane_mask(W.cur >= 0) & lane_mask(walk_queue_has_room<kScene>(W, tq)); }
    HD static unsigned long long wants_test_mask(const State& W, TriQueue tq) { return ~can_step_mask(W, tq) & lane_mask(!walk_queue_empty(W, tq)); }
    HD static bool may_test(const State& W, TriQueue tq) { return !walk_queue_empty(W, tq); }
    HD static bool done(const State& W, TriQueue tq) { return walk_done(W, tq); }
    // A round of steps: `fetch` is called by EVERY lane of the wave (stepping = this lane takes part), then `step` by the
    // lanes that take part.  Only the cooperative record fetch does anything in `fetch`.
    HD static void fetch(const DevScene& S, const float4* lds, TriQueue tq, const State& W, bool stepping) {
        if constexpr (coop_variant(kScene)) coop_fetch(S.pairs, tq, stepping ? W.cur : -1);
    }
    HD static void step(const DevScene& S, const float4* lds, TriQueue tq, State& W COUNT_ARG) {
        if constexpr (coop_variant(kScene)) walk_step_staged<kScene>(S, tq, W COUNT_PASS);
        else walk_step<kScene>(S, records(S, lds), tq, W COUNT_PASS);
    }
    template <bool kSpheres = true>
    HD static void test_one(const DevScene& S, const float4* lds, TriQueue tq, State& W COUNT_ARG) {
        walk_test_one<kScene, kSpheres>(S, triangles(S, lds), tq, W COUNT_PASS);
    }
    HD static void settle(const DevScene&, const float4*, State& COUNT_ARG) {}
};
template <> struct Traversal<kScenePair> : PairTraversal<kScenePair> {};
template <> struct Traversal<kScenePairLds> : PairTraversal<kScenePairLds> {};
template <> struct Traversal<kScenePairCull> : PairTraversal<kScenePairCull> {};

template <int kScene>
struct MarkTraversal {
    static constexpr bool kMask32 = (kScene == kSceneTiny32);
    static constexpr bool kKeepsHitData = false;
    using State = MarkWalk<kMask32>;
    using mask_t = typename State::mask_t;
    HD static void init(State& W, TriQueue) {
        W.at = W.stop = 0u;
        W.pending = 0;
        W.filtered = true;
    }
    HD static const float4* triangles(const DevScene& S, const float4* lds) { return lds + 18u * S.num_nodes; }
    HD static void begin(const DevScene& S, const float4* lds, TriQueue, State& W, f3 o, f3 d, bool shadow, float tmax COUNT_ARG) {
        W.o = o;
        W.d = d;
        W.inv = mk(rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z));  // == float(1. / double(d)), Ray.hpp:21
        const float kInf = __builtin_inff();
        W.finite = fabsf(W.inv.x) < kInf && fabsf(W.inv.y) < kInf && fabsf(W.inv.z) < kInf;
        W.shadow = shadow;
        W.tmax = tmax;
        W.best = -1;
        W.best_t = (double)kInfinity;
        W.pending = 0;
        W.filtered = false;
        const uint32_t nbase = lds_address(lds), copy_bytes = S.num_nodes * 32u;
        const uint32_t* table = reinterpret_cast<const uint32_t*>(triangles(S, lds) + 4u * S.num_tris);
        if (W.finite) {
            const uint32_t oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
            W.at = nbase + oct * copy_bytes;
            W.stop = W.at + table[oct] * 32u;
        } else {  // a direction component is 0 (or denormal): the whole tree, the reference's NaN-aware test
            W.at = nbase + 8u * copy_bytes;
            W.stop = W.at + copy_bytes;
        }
    }
    HD static void fetch(const DevScene&, const float4*, TriQueue, const State&, bool) {}
    HD static bool can_step(const State& W, TriQueue) { return W.at < W.stop; }
    HD static bool wants_test(const State& W, TriQueue) { return W.at >= W.stop && W.pending != 0; }
    HD static unsigned long long can_step_mask(const State& W, TriQueue) { return lane_mask(W.at < W.stop); }
    HD static unsigned long long wants_test_mask(const State& W, TriQueue) { return lane_mask(W.at >= W.stop) & lane_mask(W.pending != 0); }
    HD static bool may_test(const State& W, TriQueue) { return W.at >= W.stop && W.filtered && W.pending != 0; }
    HD static bool done(const State& W, TriQueue) { return W.at >= W.stop && W.filtered && W.pending == 0; }
    HD static void step(const DevScene& S, const float4* lds, TriQueue, State& W COUNT_ARG) {
        COUNT_BOX(1);
        const float4 nr = lds_read(W.at), fr = lds_read(W.at + 16u);
        bool hit;
        if (W.finite) {
            const float tnear = fmaxf(fmaxf((nr.x - W.o.x) * W.inv.x, (nr.y - W.o.y) * W.inv.y), (nr.z - W.o.z) * W.inv.z);
            const float tfar = fminf(fminf((fr.x - W.o.x) * W.inv.x, (fr.y - W.o.y) * W.inv.y), (fr.z - W.o.z) * W.inv.z);
            hit = tnear <= tfar;
        } else {
            hit = box_hit(nr, fr, W.o, W.inv, W.d.x < 0, W.d.y < 0, W.d.z < 0);
        }
        if constexpr (kMask32) {
            W.pending |= hit ? __float_as_uint(fr.w) : 0u;
        } else {
            const int tri = __float_as_int(fr.w);
            if (hit && tri >= 0) W.pending |= (mask_t)1 << tri;
        }
        W.at = hit ? W.at + 32u : __float_as_uint(nr.w);
    }
    // the walk has ended: drop the marked triangles that face away from the ray (Triangle.hpp:168-169);
    // the octant decides most of them, the rest are tested here
    HD static void settle(const DevScene& S, const float4* lds, State& W COUNT_ARG) {
        if (W.filtered || W.at < W.stop) return;
        W.filtered = true;
        const float4* T = triangles(S, lds);
        const uint32_t* table = reinterpret_cast<const uint32_t*>(T + 4u * S.num_tris);
        mask_t undecided = W.pending;
        if (W.finite) {
            const uint32_t oct = (W.d.x < 0 ? 1u : 0u) | (W.d.y < 0 ? 2u : 0u) | (W.d.z < 0 ? 4u : 0u);
            if constexpr (kMask32) undecided &= table[16u + 2u * oct];
            else undecided &= ((mask_t)table[16u + 2u * oct] | ((mask_t)table[17u + 2u * oct] << 32));
        }
        if constexpr (kMask32) {
            for (uint32_t m = undecided; m;) {
                const int tri = __ffs((int)m) - 1;
                const float4 nr = T[4 * tri + 3];
                const uint32_t group = __float_as_uint(nr.w);  // the triangles with this normal: one test settles them all
                COUNT_TRI(dot(W.d, xyz(nr)) > 0 ? 1 : 0);
                if (dot(W.d, xyz(nr)) > 0) W.pending &= ~group;
                m &= ~group;
            }
        } else {
            for (mask_t m = undecided; m;) {
                const int tri = __ffsll((long long)m) - 1;
                const mask_t rest = m & (m - 1);
                COUNT_TRI(dot(W.d, xyz(T[4 * tri + 3])) > 0 ? 1 : 0);
                if (dot(W.d, xyz(T[4 * tri + 3])) > 0) W.pending ^= m ^ rest;
                m = rest;
            }
        }
    }
    template <bool kSpheres = true>
    HD static void test_one(const DevScene& S, const float4* lds, TriQueue, State& W COUNT_ARG) {
        const float4* T = triangles(S, lds);
        const int tri = kMask32 ? __ffs((int)W.pending) - 1 : __ffsll((long long)W.pending) - 1;
        W.pending &= W.pending - 1;
        double t;
        COUNT_TRI(1);
        if (tri_hit<true>(T, tri, W.o, W.d, t)) {
            if (W.shadow) {
                if (t < (double)W.tmax - (double)kEpsilon4) {
                    W.best = tri;
                    W.best_t = t;
                    W.pending = 0;
                }
            } else if (t < W.best_t) {
                W.best = tri;
                W.best_t = t;
            }
        }
    }
};
template <> struct Traversal<kSceneTiny32> : MarkTraversal<kSceneTiny32> {};
template <> struct Traversal<kSceneTiny64> : MarkTraversal<kSceneTiny64> {};

// -------------------------------------------------------------- shading -----
// common/MathUtils.hpp:124-222
HD f3 local_to_world(f3 local_dir, f3 normal) {
    // The reference branches on |n.x| > |n.y| and builds the tangent from (n.x, n.z) or (n.y, n.z).  A wave holds
    // lanes of both kinds, so the two sides are written as one: the same products, sum, square root and reciprocal
    // on the selected component, then the zero goes where the branch would have put it.
    const bool use_x = fabsf(normal.x) > fabsf(normal.y);
    const float a = use_x ? normal.x : normal.y;
    const float inv_len = rcp_rn(sqrt_rn(a * a + normal.z * normal.z));
    const float tz = normal.z * inv_len, ta = -a * inv_len;
    const f3 t = mk(use_x ? tz : 0.0f, use_x ? 0.0f : tz, ta);
    const f3 b = cross(t, normal);
    return local_dir.x * b + local_dir.y * t + local_dir.z * normal;
}
HD f3 sampleHemisphereCosine(f3 n, float rx, float ry) {
    float r = sqrt_rn(rx);
    float theta = ry * kPi * 2.0f;
    float sn, cs;
    portable_sincos(theta, sn, cs);
    float dx = cs * r, dy = sn * r;
    float z = sqrt_rn(1.f - (dx * dx + dy * dy));
    return local_to_world(mk(dx, dy, z), n);
}
HD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
HD float lerp(float x, float y, float t) { return x * (1.0f - t) + y * t; }
HD f3 lerp(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
HD float satDot(f3 a, f3 b) { return gmax(dot(a, b), 0.f); }
HD float absDot(f3 a, f3 b) { return fabsf(dot(a, b)); }
HD float powerHeuristic(float f, float g) {
    if (isinf(f)) return 1.0f;
    float f2 = f * f;
    return f2 / (f2 + g * g);
}
HD float luminance(f3 c) { return dot(c, mk(.2126f, .7152f, .0722f)); }

HD f3 reflect(f3 ray_in_dir, f3 normal) { return ray_in_dir - dot(ray_in_dir, normal) * normal * 2.f; }  // :6-8

// Dielectric helpers of Material::Type::Glass (Material.hpp:11-46).
HD float clampf(float lo, float hi, float v) { return gmax(lo, gmin(hi, v)); }  // MathUtils.hpp:82-84
HD f3 refract(f3 ray_in_dir, f3 normal, float ior) {
    float cos_i = clampf(-1.0f, 1.0f, dot(ray_in_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    f3 correct_normal = normal;
    if (cos_i < 0.0f) {
        cos_i = -cos_i;
    } else {
        eta_i = ior;
        eta_t = 1.0f;
        correct_normal = -normal;
    }
    const float eta = eta_i / eta_t;
    const float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
    if (k < 0.0f) return mk(0.0f);
    return normalize(eta * ray_in_dir + (eta * cos_i - sqrt_rn(k)) * correct_normal);
}
HD float fresnel(f3 observation_dir, f3 normal, float ior) {
    float cos_i = clampf(-1.0f, 1.0f, dot(observation_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    if (cos_i > 0.0f) {
        eta_i = ior;
        eta_t = 1.0f;
    }
    const float sin_t = eta_i / eta_t * sqrt_rn(gmax(0.0f, 1.0f - cos_i * cos_i));
    if (sin_t >= 1.0f) return 1.0f;  // total internal reflection
    const float cos_t = sqrt_rn(gmax(0.0f, 1.0f - sin_t * sin_t));
    cos_i = fabsf(cos_i);
    const float rs = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    const float rp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rs * rs + rp * rp) / 2;
}

// Material::pdf and Material::bsdf of Glass (Material.hpp:277-309, :348-381) with
// Microfacet::outward_micro_surface_normal (:82-98) and refract_jacobian (:104-108).
HD void glass_eval(float rsq, float rsq_minus_1, float roughness, float ior, f3 src, f3 out, f3 normal,
                                        f3& bsdf, float& pdf) {
    const float nds = dot(normal, src), ndo = dot(normal, out);
    const float check = nds * ndo;
    if (check == 0.0f) {
        pdf = 0.0f;
        bsdf = mk(0.0f);
        return;
    }
    const bool same_side = check > 0.0f, outward = ndo > 0.0f;
    f3 mm;
    if (same_side) {
        mm = normalize(src + out);
        if (!outward) mm = -mm;
    } else {
        mm = outward ? -normalize(out + src * ior) : -normalize(ior * out + src);
    }
    const float F = fresnel(-out, mm, ior);
    const float ndm = dot(normal, mm);
    const float mds = dot(mm, src), mdo = dot(mm, out);
    const float andm = fabsf(ndm);
    // Microfacet::distribution (:48-53); |ndm|^2 == ndm^2, so D of pdf and bsdf are one value
    float denominator = (ndm * ndm) * rsq_minus_1 + 1.0f;
    denominator = kPi * denominator * denominator;
    const float D = rsq / denominator;
    const float pm = D * andm;                                                      // pdf_micro_surface (:76-80)
    const float G = 2.0f / lerp(fabsf(2 * mds * mdo), fabsf(mds + mdo), roughness);  // geometry (:59-61)
    if (same_side) {
        const float jac = mdo == 0.0f ? 0.0f : rcp_rn(4 * fabsf(mdo));  // reflect_jacobian (:100-102)
        pdf = pm * F * jac;
        bsdf = mk(D * F * G / 4.0f);
        return;
    }
    const float ior_in = nds < 0.0f ? ior : 1.0f, ior_out = ndo < 0.0f ? ior : 1.0f;
    float den = ior_in * mds + ior_out * mdo;
    den *= den;
    const float jac = den == 0.0f ? 0.0f : (ior_out * ior_out * fabsf(mdo)) / den;
    pdf = pm * (1.0f - F) * jac;
    bsdf = mk(jac * fabsf(mds) * D * (1.0f - F) * G);
}

// Material::pdf and Material::bsdf (Material.hpp:257-312, :315-384), evaluated together
// because the integrator always needs both.  kExt = false compiles the Glass branch out:
// kernels for scenes without a glass material pay nothing for it in registers.
template <bool kExt>
HD void mat_eval(const DevMaterial& m, f3 src, f3 out, f3 normal, f3& bsdf, float& pdf) {
    if (m.type == 0) {
        const bool front = dot(out, normal) > 0.0f;
        pdf = front ? 0.5f * 1.f / kPi : 0.0f;
        bsdf = front ? mk(m.albedo_over_pi[0], m.albedo_over_pi[1], m.albedo_over_pi[2]) : mk(0.0f);
        return;
    }
    if constexpr (kExt) {
        if (m.type == 2) {
            glass_eval(m.rsq, m.rsq_minus_1, m.roughness, m.ior, src, out, normal, bsdf, pdf);
            return;
        }
    }
    const float nds = dot(normal, src), ndo = dot(normal, out);
    const float check = nds * ndo;
    if (check <= 0.0f) {
        pdf = 0.0f;
        bsdf = mk(0.0f);
        return;
    }
    const f3 mm = normalize(src + out);
    const float ndm = dot(normal, mm);
    const float mdo = dot(mm, out);
    const float andm = fabsf(ndm);
    // Microfacet::distribution (:48-53); |ndm|^2 == ndm^2, so D of pdf and bsdf are one value
    float denominator = (ndm * ndm) * m.rsq_minus_1 + 1.0f;
    denominator = kPi * denominator * denominator;
    const float D = m.rsq / denominator;
    const float jac = mdo == 0.0f ? 0.0f : rcp_rn(4 * fabsf(mdo));  // reflect_jacobian (:100-102)
    pdf = (D * andm) * jac;
    const float G = two_over(lerp(fabsf(2 * nds * ndo), fabsf(nds + ndo), m.roughness));  // geometry (:59-61)
    const f3 f0 = mk(m.f0[0], m.f0[1], m.f0[2]);
    const f3 F = f0 + (mk(1.0f) - f0) * pow5(1.0f - mdo);
    const f3 kd = (mk(1.0f) - F) * m.one_minus_metallic;
    // "* 1.f" of PiInv is exact.  A fully metallic surface has kd = +0: +0 / Pi is +0, and a wave whose
    // microfacet lanes are all of that kind skips the three IEEE divisions
    const f3 kda = kd * mk(m.albedo[0], m.albedo[1], m.albedo[2]);
    f3 diffuse = mk(0.0f);
    if ((__float_as_uint(kda.x) | __float_as_uint(kda.y) | __float_as_uint(kda.z)) != 0u) diffuse = kda / kPi;
    const f3 specular = D * F * G / 4.0f;
    bsdf = diffuse + specular;
}
// The choice between reflection and transmission of Material::sample (Material.hpp:200-215)
HD f3 glass_direction(f3 observation_dir, f3 micro, float ior, float r2) {
    const float f = fresnel(observation_dir, micro, ior);
    if (r2 < f) return reflect(observation_dir, micro);
    return refract(observation_dir, micro, ior);
}
// Material::sample (Material.hpp:186-219); r0/r1/r2 are draws kSlotBsdf0/1/2.
// A wave usually holds lanes on Lambertian and on microfacet surfaces and then runs both
// branches, so what the two have in common is computed once, outside them: both turn r1 into an
// azimuth -- sampleHemisphereCosine's `ry * Pi * 2.0f` and the microfacet `2 * Pi * r1` are the same
// float (a factor of 2 is exact) -- and both end in local_to_world about the normal.  Per lane the
// operations and their order are those of the reference.
template <bool kExt>
HD f3 mat_sample(const DevMaterial& m, float r0, float r1, float r2, f3 ray_out_dir, f3 normal) {
    float sp, cp;
    portable_sincos(r1 * kPi * 2.0f, sp, cp);
    f3 local;
    if (m.type == 0) {  // sampleHemisphereCosine (MathUtils.hpp:187-197)
        const float r = sqrt_rn(r0);
        const float dx = cp * r, dy = sp * r;
        local = mk(dx, dy, sqrt_rn(1.f - (dx * dx + dy * dy)));
    } else {
        const float theta = portable_acos(sqrt_rn((1.0f - r0) / (m.rsq_minus_1 * r0 + 1.0f)));
        float st, ct;
        portable_sincos(theta, st, ct);
        local = mk(st * cp, st * sp, ct);
    }
    const f3 w = local_to_world(local, normal);  // the sampled direction (Lambert) or the micro-normal
    if (m.type == 0) return w;
    if constexpr (kExt) {
        if (m.type == 2) return glass_direction(-ray_out_dir, w, m.ior, r2);
    }
    return reflect(-ray_out_dir, w);
}

// Material and shading normal of a hit.  kExt: the table of regularised materials serves a path
// that has left a non-specular surface behind when the integrator regularises (Scene.hpp:186-189);
// a sphere's normal is normalize(coords - center) (Sphere.hpp:62).
// Where the shading reads the hit triangle's record and the materials: the copies in LDS where a kernel has them
// (triangle records: the variants that stage their scene; materials: the kernels built without kExt, whose scenes have
// at most kMaxLdsMaterials materials), the device arrays otherwise.  Same bits either way.
struct ShadeSrc {
    const float4* T;
    const DevMaterial* M;
    const float4* L;  // DevScene::fast_light's copy in LDS, or null
    // M[m] with a 24-bit multiply for the offset (the 32-bit one is a quarter-rate instruction)
    HD const DevMaterial& material(uint32_t m) const {
        return *reinterpret_cast<const DevMaterial*>(reinterpret_cast<const char*>(M) + __umul24(m, (uint32_t)sizeof(DevMaterial)));
    }
};
template <int kScene, bool kExt>
HD ShadeSrc shade_source(const DevScene& S, const float4* lds) {
    ShadeSrc src;
#ifndef HPT_SHADE_T_GLOBAL
#define HPT_SHADE_T_GLOBAL 0
#endif
#ifndef HPT_SHADE_M_GLOBAL
#define HPT_SHADE_M_GLOBAL 0
#endif
    constexpr bool kTiny = kScene == kSceneTiny32 || kScene == kSceneTiny64;
    if constexpr (kScene == kScenePairLds) src.T = lds + 4u * S.num_pairs;
    else if constexpr (kScene == kSceneLds) src.T = lds + 2u * S.num_nodes;
    else if constexpr (kTiny && !HPT_SHADE_T_GLOBAL) src.T = lds + 18u * S.num_nodes;
    else src.T = S.tris;
    if constexpr (kTiny && HPT_SHADE_M_GLOBAL) src.M = S.mats;
    else if constexpr (kExt) src.M = S.mat_lds_float4s ? reinterpret_cast<const DevMaterial*>(lds + S.mat_lds_off) : S.mats;
    else src.M = reinterpret_cast<const DevMaterial*>(lds + S.mat_lds_off);
    src.L = S.fast_light ? lds + S.light_lds_off : nullptr;
    return src;
}
// Scene::sampleLight (common/Scene.hpp:99-120) with BVHAccel::Sample / getSample (common/BVH.hpp:147-184) and
// Triangle::Sample (common/Triangle.hpp:75-82): the sampled point, its normal, the triangle's MIS light pdf, the
// sampling pdf (0 = no light chosen) and the light's index, from the four draws of one Philox block.
struct LightPoint {
    f3 x, n;
    float lightPdf, lightSamplePdf;
    uint32_t light;
};
HD LightPoint sample_light(const DevScene& S, const ShadeSrc& src, const u4 ra) {
    LightPoint lp;
    lp.x = lp.n = mk(0.f);
    lp.lightPdf = lp.lightSamplePdf = 0.0f;
    lp.light = 0u;
    const float pick = u01(ra.x) * S.sum_light_area;
    if (src.L) {  // (wave-uniform) one light of two triangles, staged in LDS
        const float4 areas = src.L[8];  // mesh area, root area, 1 / root area, left leaf's area
        const float emit_area_sum = 0.f + areas.x;
        if (pick <= emit_area_sum) {
            const float q = sqrt_rn(u01(ra.y)) * areas.y;
            const float4* tri = src.L + (q < areas.w ? 0 : 4);  // the root's children are leaves: left if q < its area, else right
            const float sx = sqrt_rn(u01(ra.z)), sy = u01(ra.w);
            const float4 a0 = tri[0], b1 = tri[1], b2 = tri[2], a3 = tri[3];
            lp.x = xyz(a0) * (1.0f - sx) + xyz(b1) * (sx * (1.0f - sy)) + xyz(b2) * (sx * sy);
            lp.n = xyz(a3);
            lp.lightPdf = a0.w;
            lp.lightSamplePdf = areas.z;
        }
        return lp;
    }
    float emit_area_sum = 0;
    for (uint32_t k = 0; k < S.num_lights; ++k) {
        const DevLight& L = S.lights[k];
        emit_area_sum += L.area;
        if (pick <= emit_area_sum) {
            float q = sqrt_rn(u01(ra.y)) * L.root_area;
            uint32_t node = L.root_node;
            int tri;
            while ((tri = __float_as_int(S.nodes[2 * node + 1].w)) < 0) {
                const uint32_t left = node + 1;
                const float la = S.node_area[left];
                if (q < la) node = left;
                else {
                    q -= la;
                    node = (uint32_t)__float_as_int(S.nodes[2 * left].w);
                }
            }
            const float sx = sqrt_rn(u01(ra.z)), sy = u01(ra.w);
            const float4 a0 = S.tris[4 * tri], a3 = S.tris[4 * tri + 3];
            const float4 b1 = S.tri_v12[2 * tri], b2 = S.tri_v12[2 * tri + 1];
            lp.x = xyz(a0) * (1.0f - sx) + xyz(b1) * (sx * (1.0f - sy)) + xyz(b2) * (sx * sy);
            lp.n = xyz(a3);
            lp.lightPdf = a0.w;
            lp.lightSamplePdf = L.inv_root_area;
            lp.light = k;
            break;
        }
    }
    return lp;
}
HD f3 light_emission(const DevScene& S, const ShadeSrc& src, uint32_t light) {
    if (src.L) return xyz(src.L[9]);
    const DevLight& L = S.lights[light];
    return mk(L.emission[0], L.emission[1], L.emission[2]);
}
template <bool kExt>
HD const DevMaterial& hit_material(const DevScene& S, const ShadeSrc& src, int hit_tri, bool diffused) {
    uint32_t m = (uint32_t)__float_as_int(src.T[4 * hit_tri + 1].w);
    if constexpr (kExt) {
        if ((S.integrator & kIntegratorRegularize) && diffused) m += S.num_mats;
    }
    return src.material(m);
}
template <bool kExt>
HD f3 hit_normal(const DevScene& S, const ShadeSrc& src, int hit_tri, f3 coords) {
    const float4 n4 = src.T[4 * hit_tri + 3];
    if constexpr (kExt) {
        if (S.has_spheres && (__float_as_uint(n4.w) & kSphereBit)) return normalize(coords - xyz(src.T[4 * hit_tri]));
    }
    return xyz(n4);
}
// Per-path state carried between bounces (PathSegment, common/Ray.hpp:36-52,
// minus the RNG state, which the keyed stream makes unnecessary).
struct Path {
    f3 o, d;          // current ray
    f3 thr, rad;      // throughput, accumulated radiance
    float pdf;        // bsdfSamplePdf of the previous bounce
    int depth;        // bounces done so far (== loop index of Scene::castRay)
    bool specular;    // specularBounce
    bool diffused;    // anyNonSpecularBounces (read by kernels built with kExt only)
    uint32_t pixel, sample;
};

// the tail of a bounce that went on (Scene.hpp:243-264): anyNonSpecularBounces, and Russian
// roulette AS WRITTEN there -- its `break` leaves only the function's while(true) wrapper, so a
// "killed" bounce merely goes without the 1 / (1 - q) compensation; the path is not ended
template <bool kExt>
HD void bounce_tail(const DevScene& S, Path& p, float roulette_draw) {
    if constexpr (kExt) {
        p.diffused = p.diffused || !p.specular;
        if (S.integrator & kIntegratorRoulette) {
            if (p.depth > 1 && !(roulette_draw < S.roulette_q)) p.thr = p.thr / (1.f - S.roulette_q);
        }
    }
}


// Renderer.cpp:32-39 / CudaPathTrace.cu:39-59: jittered primary ray (one jitter
// value for both axes), throughput 1, radiance 0.
HD f3 camera_direction(const DevScene& S, uint32_t pixel, uint32_t sample) {
    const uint32_t i = pixel % (uint32_t)S.width, j = pixel / (uint32_t)S.width;
    const u4 r = philox4x32_10(kCameraDepth, 0u, S.seed, 0u, pixel, sample);
    const float bias = u01(r.x);
    const float x = (2 * ((float)i + bias) / (float)S.width - 1) * S.aspect * S.cam_scale;
    const float y = (1 - 2 * ((float)j + bias) / (float)S.height) * S.cam_scale;
    return normalize(mk(-x, y, 1));
}
// the state of a path that leaves the camera along d
HD void start_camera_path(const DevScene& S, uint32_t pixel, uint32_t sample, f3 d, Path& p) {
    p.o = mk(S.cam_x, S.cam_y, S.cam_z);
    p.d = d;
    p.thr = mk(1.0f);
    p.rad = mk(0.0f);
    p.pdf = 0.0f;
    p.depth = 0;
    p.specular = false;
    p.diffused = false;
    p.pixel = pixel;
    p.sample = sample;
}
HD void generate_camera_path(const DevScene& S, uint32_t pixel, uint32_t sample, Path& p) {
    start_camera_path(S, pixel, sample, camera_direction(S, pixel, sample), p);
}

// Scene::TracePath (common/Scene.hpp:122-272) for a path whose closest hit is
// (hit_tri, hit_t); hit_tri < 0 = miss.  Returns false when the path ends on
// this bounce for the reason "missed the scene" (remainingBounces = 0, :156-161);
// the caller ends it after max_depth bounces otherwise.  Counts shadow queries.
template <int kScene, bool kExt, uint32_t kShadowInc = 1u>
HD bool shade_bounce(const DevScene& S, const float4* __restrict__ lds, TriQueue tq, Path& p, int hit_tri, double hit_t,
                     uint32_t& shadow_queries PROF_ARG COUNT_ARG) {
    bool alive = true;
    const ShadeSrc src = shade_source<kScene, kExt>(S, lds);
#ifdef HPT_PROFILE
    unsigned long long prof_s0 = __builtin_readcyclecounter();
    PROF_ITER(kProfShadeA);
#endif
    do {
        if (hit_tri < 0) {
            alive = false;
            break;
        }
        // (word 3 of a device triangle record = its MIS light pdf, in the LDS copies too)
        const f3 coords = p.o + p.d * (float)hit_t;  // Ray::operator(), Ray.hpp:27
        {
            const DevMaterial& material = hit_material<kExt>(S, src, hit_tri, p.diffused);
            if (material.flags & kMatEmitting) {  // :164-184
                const f3 radiance = mk(material.emission[0], material.emission[1], material.emission[2]);
                if (p.depth == 0 || p.specular) {
                    p.rad = p.rad + p.thr * radiance;
                } else {
                    const float weight = powerHeuristic(p.pdf, src.T[4 * hit_tri].w);
                    p.rad = p.rad + p.thr * radiance * weight;
                }
            }
        }

        if (!p.specular) {  // :192-228
            // Scene::sampleLight (:99-120)
            const u4 ra = philox4x32_10((uint32_t)p.depth, 0u, S.seed, 0u, p.pixel, p.sample);
            const LightPoint lp = sample_light(S, src, ra);
            const float lightSamplePdf = lp.lightSamplePdf, lightPdf = lp.lightPdf;
            const f3 lx = lp.x, ln = lp.n;
            uint32_t light = lp.light;
            if (lightSamplePdf > 0.0f) {
                const f3 px = lx - coords;
                const f3 wi = normalize(px);
                const f3 so = coords + wi * kEpsilon5;
                // everything that depends on the light sample is reduced to scalars here, and
                // the hit's normal / material are re-read after the traversal, so that little
                // stays live in registers across it
                const float cos_theta_prime = satDot(ln, -wi);
                const float r2 = dot(lx - coords, lx - coords);
                int occ_tri;
                double occ_t;
                shadow_queries += kShadowInc;
#ifdef HPT_PROFILE
                prof_add(prof, kProfShadeA, __builtin_readcyclecounter() - prof_s0);
#endif
                trace<kScene, true>(S, lds, tq, so, wi, (double)length(px), occ_tri, occ_t PROF_PASS COUNT_PASS);
#ifdef HPT_PROFILE
                prof_s0 = __builtin_readcyclecounter();
                PROF_ITER(kProfShadeB);
#endif
                if (occ_tri < 0) {
                    asm volatile("" : "+v"(hit_tri), "+v"(light));  // forget what was derived from them
                    const f3 normal = hit_normal<kExt>(S, src, hit_tri, coords);
                    const DevMaterial& material = hit_material<kExt>(S, src, hit_tri, p.diffused);
                    const f3 lemit = light_emission(S, src, light);
                    const f3 wo = -p.d;
                    f3 bsdf;
                    float bsdfPdf;
                    mat_eval<kExt>(material, wi, wo, normal, bsdf, bsdfPdf);
                    const float cos_theta = satDot(normal, wi);
                    const float weight = powerHeuristic(lightPdf, bsdfPdf);
                    p.rad = p.rad + p.thr * lemit * bsdf * cos_theta * cos_theta_prime / r2 / lightSamplePdf * weight;
                }
            }
        }

        // :230-250
        asm volatile("" : "+v"(hit_tri));
        const f3 normal = hit_normal<kExt>(S, src, hit_tri, coords);
        const DevMaterial& material = hit_material<kExt>(S, src, hit_tri, p.diffused);
        const u4 rb = philox4x32_10((uint32_t)p.depth, 1u, S.seed, 0u, p.pixel, p.sample);
        const f3 wo = -p.d;
        const f3 wi = mat_sample<kExt>(material, u01(rb.x), u01(rb.y), u01(rb.z), wo, normal);
        f3 bsdf;
        float pdf;
        mat_eval<kExt>(material, wi, wo, normal, bsdf, pdf);
        p.pdf = pdf;
        if (pdf < kEpsilon5) break;  // the path keeps its ray and re-traces it (:238-240)
        p.specular = (material.flags & kMatSpecular) != 0;
        const float cos_theta = absDot(normal, wi);
        p.thr = p.thr * (bsdf * cos_theta / pdf);
        p.o = coords + wi * kEpsilon5;
        p.d = wi;
        bounce_tail<kExt>(S, p, u01(rb.w));
    } while (false);

#ifdef HPT_PROFILE
    prof_add(prof, kProfShadeB, __builtin_readcyclecounter() - prof_s0);
#endif
    p.depth++;  // :268-271
    if (isnan(p.rad.x) || isnan(p.rad.y) || isnan(p.rad.z) || isinf(p.rad.x) || isinf(p.rad.y) || isinf(p.rad.z))
        p.rad = mk(0.0f);
    return alive;
}

// Scene::TracePath cut in two at its shadow query, for the wave scheduler: shade_a is
// :156-203 (miss, emission, light sample, the shadow ray), shade_b is :205-271 (the direct-light
// term once the occlusion answer is known, BSDF sample, depth++ and the NaN reset).  The
// expressions are those of shade_bounce, operation for operation.
struct LightSample {
    float lightPdf, lightSamplePdf, cos_theta_prime, r2;
    uint32_t light;
    // the hit's shading normal and its material's index (table offset included), read once in shade_a: shade_b starts
    // without a fetch of its own
    f3 normal;
    uint32_t material;
};
enum { kShadeMiss = 0, kShadeShadowRay = 1, kShadeNoShadowRay = 2 };
HD void finish_bounce(Path& p) {  // :268-271
    p.depth++;
    if (isnan(p.rad.x) || isnan(p.rad.y) || isnan(p.rad.z) || isinf(p.rad.x) || isinf(p.rad.y) || isinf(p.rad.z))
        p.rad = mk(0.0f);
}
// kHave: the walk kept the hit's normal and material index and the caller has put them into ls.normal / ls.material;
// else they are fetched here
template <bool kExt, bool kHave = false>
HD int shade_a(const DevScene& S, const ShadeSrc& src, Path& p, int hit_tri, double hit_t, f3& so, f3& wi, float& tmax, LightSample& ls) {
    if (hit_tri < 0) {
        finish_bounce(p);
        return kShadeMiss;
    }
    const f3 coords = p.o + p.d * (float)hit_t;  // Ray::operator(), Ray.hpp:27
    {
        uint32_t m;
        if constexpr (kHave && !kExt) {
            m = ls.material;
        } else {
            // one fetch: material index and normal
            m = (uint32_t)__float_as_int(src.T[4 * hit_tri + 1].w);
            ls.normal = hit_normal<kExt>(S, src, hit_tri, coords);
        }
        if constexpr (kExt) {
            if ((S.integrator & kIntegratorRegularize) && p.diffused) m += S.num_mats;
        }
        ls.material = m;
        const DevMaterial& material = src.material(m);
        if (material.flags & kMatEmitting) {  // :164-184
            const f3 radiance = mk(material.emission[0], material.emission[1], material.emission[2]);
            if (p.depth == 0 || p.specular) {
                p.rad = p.rad + p.thr * radiance;
            } else {
                const float light_pdf_of_hit = src.T[4 * hit_tri].w;  // (few paths come here: the light was hit)
                const float weight = powerHeuristic(p.pdf, light_pdf_of_hit);
                p.rad = p.rad + p.thr * radiance * weight;
            }
        }
    }
    if (p.specular) return kShadeNoShadowRay;
    // Scene::sampleLight (:99-120)
    const u4 ra = philox4x32_10((uint32_t)p.depth, 0u, S.seed, 0u, p.pixel, p.sample);
    const LightPoint lp = sample_light(S, src, ra);
    const float lightSamplePdf = lp.lightSamplePdf, lightPdf = lp.lightPdf;
    const f3 lx = lp.x, ln = lp.n;
    const uint32_t light = lp.light;
    if (!(lightSamplePdf > 0.0f)) return kShadeNoShadowRay;
    const f3 px = lx - coords;
    wi = normalize(px);
    so = coords + wi * kEpsilon5;
    ls.cos_theta_prime = satDot(ln, -wi);
    ls.r2 = dot(lx - coords, lx - coords);
    ls.lightPdf = lightPdf;
    ls.lightSamplePdf = lightSamplePdf;
    ls.light = light;
    tmax = length(px);
    return kShadeShadowRay;
}
// lit: a shadow ray towards the light sample `ls` along `wi` was traced and found nothing
template <bool kExt>
HD void shade_b(const DevScene& S, const ShadeSrc& src, Path& p, int hit_tri, double hit_t, bool lit, f3 wi_light, const LightSample& ls) {
    const f3 coords = p.o + p.d * (float)hit_t;
    const f3 normal = ls.normal;
    const DevMaterial& material = src.material(ls.material);
    const f3 wo = -p.d;
    if (lit) {  // :208-227
        const f3 lemit = light_emission(S, src, ls.light);
        f3 bsdf;
        float bsdfPdf;
        mat_eval<kExt>(material, wi_light, wo, normal, bsdf, bsdfPdf);
        const float cos_theta = satDot(normal, wi_light);
        const float weight = powerHeuristic(ls.lightPdf, bsdfPdf);
        p.rad = p.rad + p.thr * lemit * bsdf * cos_theta * ls.cos_theta_prime / ls.r2 / ls.lightSamplePdf * weight;
    }
    // :230-250
    const u4 rb = philox4x32_10((uint32_t)p.depth, 1u, S.seed, 0u, p.pixel, p.sample);
    const f3 wi = mat_sample<kExt>(material, u01(rb.x), u01(rb.y), u01(rb.z), wo, normal);
    f3 bsdf;
    float pdf;
    mat_eval<kExt>(material, wi, wo, normal, bsdf, pdf);
    p.pdf = pdf;
    if (!(pdf < kEpsilon5)) {  // else the path keeps its ray and re-traces it (:238-240)
        p.specular = (material.flags & kMatSpecular) != 0;
        const float cos_theta = absDot(normal, wi);
        p.thr = p.thr * (bsdf * cos_theta / pdf);
        p.o = coords + wi * kEpsilon5;
        p.d = wi;
        bounce_tail<kExt>(S, p, u01(rb.w));
    }
    finish_bounce(p);
}

#undef HD
}  // namespace hpt

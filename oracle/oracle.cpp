// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference path
// tracer's hot path; see oracle.h for the usage rules and the parity status.
//
// Every function cites the reference file:line it restates.  The code keeps
// the reference's data model on purpose (array-of-structs triangles, a
// two-level pointer BVH with nextIfHit/nextIfMiss links, a fat hit record):
// it is written to be *obviously the same algorithm*, not to be fast, and it
// shares no code with the product under cudapathtracer_amd/.
//
// Arithmetic contract: plain IEEE-754 binary32/binary64 operations in exactly
// the reference's (glm 0.9.9.0) order; compiled without FMA contraction.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include <omp.h>

namespace orc {

// ----------------------------------------------------------------------------
// glm::vec3 arithmetic in glm's operation order
// (external/glm/detail/func_geometric.inl:54-96, func_exponential.inl:128-133)
// ----------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
static inline V3 mk(float a, float b, float c) { return V3{a, b, c}; }
static inline V3 mk(float a) { return V3{a, a, a}; }
static inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
static inline float dot(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
static inline V3 cross(V3 x, V3 y) {
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float length(V3 v) { return std::sqrt(dot(v, v)); }
static inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
static inline float gmin(float x, float y) { return x < y ? x : y; }  // glm::min (detail/func_common.inl:15-19)
static inline float gmax(float x, float y) { return x > y ? x : y; }  // glm::max (detail/func_common.inl:23-28)
static inline V3 gmin(V3 a, V3 b) { return mk(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline V3 gmax(V3 a, V3 b) { return mk(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }

// common/MathUtils.hpp:64-74
static const float kPi = 3.1415926535897932384626422832795028841971f;
static const float kEpsilon5 = 1e-5f;
static const float kEpsilon4 = (5 * 1e-4f);
static const float kInfinity = 1e30f;  // kDoubleInfinity is a *float* literal

// ----------------------------------------------------------------------------
// Contract of the new core (NOT from the reference): portable sin/cos/acos,
// Philox4x32-10 keyed uniforms, fixed-point accumulation.  The HIP kernels
// implement the same definitions operation for operation, which is what makes
// GPU-vs-oracle images comparable bit for bit.
// ----------------------------------------------------------------------------
static inline void portable_sincos(float x, float* s_out, float* c_out) {
    // quadrant reduction with a 3-term pi/2 (Cody-Waite), then the classic
    // single-precision minimax polynomials on [-pi/4, pi/4]
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
    switch (k & 3) {
    case 0: *s_out = sp; *c_out = cp; break;
    case 1: *s_out = cp; *c_out = -sp; break;
    case 2: *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp; break;
    }
}
static inline float portable_asin_poly(float a, float z) {
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
static inline float portable_acos(float x) {
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = std::sqrt(z);
        return 2.0f * portable_asin_poly(s, z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = std::sqrt(z);
        return kPi - 2.0f * portable_asin_poly(s, z);
    }
    return 1.5707963267948966f - portable_asin_poly(x, x * x);
}

static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float philox_uniform(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot, uint32_t seed) {
    uint32_t ctr[4] = {depth, slot >> 2, seed, 0u};
    uint32_t key[2] = {pixel, sample};
    uint32_t out[4];
    philox4x32_10(ctr, key, out);
    return (float)(out[slot & 3] >> 8) * (1.0f / 16777216.0f);
}
static const uint32_t kCameraDepth = 0xFFFFFFFFu;  // counter word of the pixel-jitter draw

static inline int64_t radiance_to_fixed(float v) {
    double x = (double)v;
    if (!(x == x)) return 0;
    if (x > 1048576.0) x = 1048576.0;
    if (x < -1048576.0) x = -1048576.0;
    return (int64_t)std::rint(x * 16777216.0);
}
static inline float fixed_to_radiance(int64_t acc, int spp_total) {
    return (float)((double)acc / (16777216.0 * (double)spp_total));
}

// ----------------------------------------------------------------------------
// RNG (common/MathUtils.hpp:22-62) -- the reference stream (mt19937 seeded by
// utilhash) or the new core's keyed Philox stream behind one interface.  The
// shading code asks for draw (depth, slot) *in the reference's program order*,
// so the MT stream, which ignores the arguments, sees the same call sequence
// as the reference binary.
// ----------------------------------------------------------------------------
static inline unsigned int utilhash(unsigned int a) {  // MathUtils.hpp:53-61
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}

enum Slot { kSlotLightPick = 0, kSlotTriPick = 1, kSlotTriX = 2, kSlotTriY = 3, kSlotBsdf0 = 4, kSlotBsdf1 = 5, kSlotBsdf2 = 6, kSlotRoulette = 7 };

struct Rng {
    int mode = ORC_RNG_MT;
    std::mt19937 mt;
    const float* fixed = nullptr;  // explicit stream for the unit-level entry points
    int fixed_pos = 0;
    uint32_t pixel = 0, sample = 0, seed = 0;
    Rng() {}
    Rng(int mode_, uint32_t seed_, int iter, int index) : mode(mode_), pixel((uint32_t)index), sample((uint32_t)iter), seed(seed_) {
        if (mode == ORC_RNG_MT) {
            // MathUtils.hpp:25-28, dim = 0
            unsigned int h = utilhash((1u << 31) | (0u << 22) | (unsigned int)iter) ^ utilhash((unsigned int)index);
            mt = std::mt19937(h);
        }
    }
    float draw(uint32_t depth, uint32_t slot) {
        if (fixed) return fixed[fixed_pos++];
        if (mode == ORC_RNG_MT) return std::uniform_real_distribution<float>(0.f, 1.f)(mt);
        return philox_uniform(pixel, sample, depth, slot, seed);
    }
};

// ----------------------------------------------------------------------------
struct Trig {
    bool portable = false;
    float sin(float x) const { if (portable) { float s, c; portable_sincos(x, &s, &c); return s; } return std::sin(x); }
    float cos(float x) const { if (portable) { float s, c; portable_sincos(x, &s, &c); return c; } return std::cos(x); }
    float acos(float x) const { return portable ? portable_acos(x) : std::acos(x); }
};

// ----------------------------------------------------------------------------
// common/MathUtils.hpp:124-222
// ----------------------------------------------------------------------------
static inline V3 local_to_world(V3 local_dir, V3 normal) {  // :124-136
    V3 t;
    if (std::fabs(normal.x) > std::fabs(normal.y)) {
        const float inv_len = 1.0f / std::sqrt(normal.x * normal.x + normal.z * normal.z);
        t = mk(normal.z * inv_len, 0.0f, -normal.x * inv_len);
    } else {
        const float inv_len = 1.0f / std::sqrt(normal.y * normal.y + normal.z * normal.z);
        t = mk(0.0f, normal.z * inv_len, -normal.y * inv_len);
    }
    const V3 b = cross(t, normal);
    return local_dir.x * b + local_dir.y * t + local_dir.z * normal;
}
static inline V3 polar_to_cartesian(const Trig& tg, float theta, float phi) {  // :138-143
    return mk(tg.sin(theta) * tg.cos(phi), tg.sin(theta) * tg.sin(phi), tg.cos(theta));
}
static inline V3 sampleHemisphereCosine(const Trig& tg, V3 n, float rx, float ry) {  // :152-162
    float r = std::sqrt(rx);
    float theta = ry * kPi * 2.0f;
    float dx = tg.cos(theta) * r, dy = tg.sin(theta) * r;
    float z = std::sqrt(1.f - (dx * dx + dy * dy));
    return local_to_world(mk(dx, dy, z), n);
}
static inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }             // :174-177
static inline float lerp(float x, float y, float t) { return x * (1.0f - t) + y * t; }   // :179-181
static inline V3 lerp(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }            // :182-184
static inline float satDot(V3 a, V3 b) { return gmax(dot(a, b), 0.f); }                  // :189-191
static inline float absDot(V3 a, V3 b) { return std::fabs(dot(a, b)); }                  // :193-195
static inline float square(float x) { return x * x; }                                    // :196-198
static inline float powerHeuristic(float f, float g) {                                   // :200-211
    if (std::isinf(f)) return 1.0f;
    float f2 = f * f;
    return f2 / (f2 + g * g);
}
static inline float luminance(V3 c) { return dot(c, mk(.2126f, .7152f, .0722f)); }       // :218-222

// ----------------------------------------------------------------------------
// common/Material.hpp
// ----------------------------------------------------------------------------
static inline float clampf(float lo, float hi, float v) { return gmax(lo, gmin(hi, v)); }  // MathUtils.hpp:82-84
static inline V3 reflect(V3 ray_in_dir, V3 normal) {                                        // Material.hpp:6-8
    return ray_in_dir - dot(ray_in_dir, normal) * normal * 2.f;
}
static inline V3 refract(V3 ray_in_dir, V3 normal, float ior) {  // Material.hpp:11-26
    float cos_i = clampf(-1.0f, 1.0f, dot(ray_in_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    V3 correct_normal = normal;
    if (cos_i < 0.0f) cos_i = -cos_i;
    else { std::swap(eta_i, eta_t); correct_normal = -normal; }
    const float eta = eta_i / eta_t;
    const float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
    return k < 0.0f ? mk(0.0f) : normalize(eta * ray_in_dir + (eta * cos_i - std::sqrt(k)) * correct_normal);
}
static inline float fresnel(V3 observation_dir, V3 normal, float ior) {  // Material.hpp:27-46
    float cos_i = clampf(-1.0f, 1.0f, dot(observation_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    if (cos_i > 0.0f) std::swap(eta_i, eta_t);
    const float sin_t = eta_i / eta_t * std::sqrt(gmax(0.0f, 1.0f - cos_i * cos_i));
    if (sin_t >= 1.0f) return 1.0f;
    const float cos_t = std::sqrt(gmax(0.0f, 1.0f - sin_t * sin_t));
    cos_i = std::fabs(cos_i);
    const float rs = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    const float rp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rs * rs + rp * rp) / 2;
}
namespace Microfacet {
static inline float distribution(float ndm, float roughness_sq) {  // :48-53
    const float ndm_sq = ndm * ndm;
    float denominator = ndm_sq * (roughness_sq - 1.0f) + 1.0f;
    denominator = kPi * denominator * denominator;
    return roughness_sq / denominator;
}
static inline V3 fresnel_schlick(float mdo, V3 f0) { return f0 + (mk(1.0f) - f0) * pow5(1.0f - mdo); }  // :55-57
static inline float geometry(float nl, float nv, float roughness) {                                     // :59-61
    return 2.0f / lerp(std::fabs(2 * nl * nv), std::fabs(nl + nv), roughness);
}
static inline V3 sample_micro_surface(const Trig& tg, float r0, float r1, V3 normal, float roughness_sq) {  // :66-74
    const float theta = tg.acos(std::sqrt((1.0f - r0) / ((roughness_sq - 1.0f) * r0 + 1.0f)));
    const float phi = 2 * kPi * r1;
    return local_to_world(polar_to_cartesian(tg, theta, phi), normal);
}
static inline float pdf_micro_surface(float ndm, float roughness_sq) {  // :76-80
    const float a = std::fabs(ndm);
    return distribution(a, roughness_sq) * a;
}
static inline V3 outward_micro_surface_normal(V3 src, V3 out, bool same_side, bool outward, float ior) {  // :82-98
    if (same_side) return outward ? normalize(src + out) : -normalize(src + out);
    return outward ? -normalize(out + src * ior) : -normalize(ior * out + src);
}
static inline float reflect_jacobian(float mdo) { return mdo == 0.0f ? 0.0f : 1.0f / (4 * std::fabs(mdo)); }  // :100-102
static inline float refract_jacobian(float mds, float ior_in, float mdo, float ior_out) {                  // :104-108
    float denominator = ior_in * mds + ior_out * mdo;
    denominator *= denominator;
    return denominator == 0.0f ? 0.0f : (ior_out * ior_out * std::fabs(mdo)) / denominator;
}
}  // namespace Microfacet

struct Material {  // Material.hpp:130-384
    enum Type { Lambertian, MetallicWorkflow, Glass };
    int type = Lambertian;
    V3 _emission = mk(0.f);
    V3 _albedo = mk(.9f);
    float _metallic = 0.f;
    float _roughness = 1.f;
    float _ior = 0.f;

    bool emitting() const { return dot(_emission, _emission) > 0.0f; }                        // :144-146
    float alpha() const { return std::sqrt(_roughness); }                                     // :151-153
    bool hasSpecular() const { return (type == MetallicWorkflow && alpha() < 1e-3f) || type == Glass; }  // :155-162
    void regularize() {                                                                        // :164-171
        float alpha_x = alpha();
        if (alpha_x < 0.3f) {
            alpha_x = gmin(gmax(2 * alpha_x, 0.1f), 0.3f);  // glm::clamp = min(max(x, lo), hi)
            _roughness = alpha_x * alpha_x;
        }
    }

    // :186-219.  The two Lambert draws are *arguments of one call* in the
    // reference (:189); g++ evaluates them right to left, so the stream's first
    // float is ry.  Metal/Glass draw r0 then r1 in separate statements (:67-68).
    V3 sample(const Trig& tg, Rng& rng, uint32_t depth, V3 ray_out_dir, V3 normal) const {
        switch (type) {
        case Lambertian: {
            float ry = rng.draw(depth, kSlotBsdf1);
            float rx = rng.draw(depth, kSlotBsdf0);
            return sampleHemisphereCosine(tg, normal, rx, ry);
        }
        case MetallicWorkflow: {
            float r0 = rng.draw(depth, kSlotBsdf0);
            float r1 = rng.draw(depth, kSlotBsdf1);
            const V3 m = Microfacet::sample_micro_surface(tg, r0, r1, normal, square(_roughness));
            return reflect(-ray_out_dir, m);
        }
        case Glass: {
            float r0 = rng.draw(depth, kSlotBsdf0);
            float r1 = rng.draw(depth, kSlotBsdf1);
            const V3 m = Microfacet::sample_micro_surface(tg, r0, r1, normal, square(_roughness));
            const V3 observation_dir = -ray_out_dir;
            const float f = fresnel(observation_dir, m, _ior);
            if (rng.draw(depth, kSlotBsdf2) < f) return reflect(observation_dir, m);
            return refract(observation_dir, m, _ior);
        }
        default: return mk(0.0f);
        }
    }

    float pdf(V3 src, V3 out, V3 normal) const {  // :257-312
        switch (type) {
        case Lambertian: return dot(out, normal) > 0.0f ? 0.5f * 1.f / kPi : 0.0f;
        case MetallicWorkflow: {
            const float check = dot(normal, src) * dot(normal, out);
            if (check <= 0.0f) return 0.0f;
            const V3 m = normalize(src + out);
            const float ndm = dot(normal, m);
            const float mdo = dot(m, out);
            return Microfacet::pdf_micro_surface(ndm, square(_roughness)) * Microfacet::reflect_jacobian(mdo);
        }
        case Glass: {
            const float nds = dot(normal, src), ndo = dot(normal, out);
            const V3 observation_dir = -out;
            const float check = nds * ndo;
            if (check == 0.0f) return 0.0f;
            const bool same_side = check > 0.0f, outward = ndo > 0.0f;
            const V3 m = Microfacet::outward_micro_surface_normal(src, out, same_side, outward, _ior);
            const float f = fresnel(observation_dir, m, _ior);
            const float ndm = dot(normal, m);
            const float pm = Microfacet::pdf_micro_surface(ndm, square(_roughness));
            const float mds = dot(m, src), mdo = dot(m, out);
            if (same_side) return pm * f * Microfacet::reflect_jacobian(mdo);
            const float ior_in = nds < 0.0f ? _ior : 1.0f, ior_out = ndo < 0.0f ? _ior : 1.0f;
            return pm * (1.0f - f) * Microfacet::refract_jacobian(mds, ior_in, mdo, ior_out);
        }
        }
        return 0.0f;
    }

    V3 bsdf(V3 src, V3 out, V3 normal) const {  // :315-384
        switch (type) {
        case Lambertian: return dot(out, normal) > 0.0f ? _albedo * 1.f / kPi : mk(0.0f);
        case MetallicWorkflow: {
            const float check = dot(normal, src) * dot(normal, out);
            if (check <= 0.0f) return mk(0.0f);
            const V3 m = normalize(src + out);
            const float ndm = dot(normal, m);
            const float mdo = dot(m, out);
            const float D = Microfacet::distribution(ndm, square(_roughness));
            const float G = Microfacet::geometry(dot(normal, src), dot(normal, out), _roughness);
            const V3 f0 = lerp(mk(0.04f), _albedo, _metallic);
            const V3 F = Microfacet::fresnel_schlick(mdo, f0);
            const V3 kd = (mk(1.0f) - F) * (1.0f - _metallic);
            const V3 diffuse = kd * _albedo * 1.f / kPi;
            const V3 specular = D * F * G / 4.0f;
            return diffuse + specular;
        }
        case Glass: {
            const float nds = dot(normal, src), ndo = dot(normal, out);
            const float check = nds * ndo;
            if (check == 0.0f) return mk(0.0f);
            const V3 observation_dir = -out;
            const bool same_side = check > 0.0f, outward = ndo > 0.0f;
            const V3 m = Microfacet::outward_micro_surface_normal(src, out, same_side, outward, _ior);
            const float ndm = dot(normal, m);
            const float mds = dot(m, src), mdo = dot(m, out);
            const float D = Microfacet::distribution(ndm, square(_roughness));
            const float G = Microfacet::geometry(mds, mdo, _roughness);
            const float F = fresnel(observation_dir, m, _ior);
            if (same_side) return mk(D * F * G / 4.0f);
            const float ior_in = nds < 0.0f ? _ior : 1.0f, ior_out = ndo < 0.0f ? _ior : 1.0f;
            float r = Microfacet::refract_jacobian(mds, ior_in, mdo, ior_out) * std::fabs(mds) * D * (1.0f - F) * G;
            return mk(r);
        }
        }
        return mk(0.0f);
    }
};

static Material from_c(const orc_material* m) {
    Material r;
    r.type = m->type;
    r._emission = mk(m->emission[0], m->emission[1], m->emission[2]);
    r._albedo = mk(m->albedo[0], m->albedo[1], m->albedo[2]);
    r._metallic = m->metallic;
    r._roughness = m->roughness;
    r._ior = m->ior;
    return r;
}

// ----------------------------------------------------------------------------
// common/Ray.hpp:9-34, common/Bounds3.hpp, common/Intersection.hpp:14-32
// ----------------------------------------------------------------------------
struct Ray {
    V3 origin, direction, direction_inv;
    bool shadowTest = false;
    double shadowDistance = kInfinity;
    Ray() {}
    Ray(V3 o, V3 d, bool st = false, double sd = kInfinity) : origin(o), direction(d), shadowTest(st), shadowDistance(sd) {
        direction_inv = mk((float)(1. / d.x), (float)(1. / d.y), (float)(1. / d.z));  // Ray.hpp:21
    }
    V3 at(double t) const { return origin + direction * (float)t; }  // Ray.hpp:27
};

struct Bounds3 {
    V3 pMin, pMax;
    Bounds3() {  // Bounds3.hpp:16-22
        pMax = mk(-kInfinity);
        pMin = mk(kInfinity);
    }
    Bounds3(V3 p1, V3 p2) {  // :24-28
        pMin = mk(std::fmin(p1.x, p2.x), std::fmin(p1.y, p2.y), std::fmin(p1.z, p2.z));
        pMax = mk(std::fmax(p1.x, p2.x), std::fmax(p1.y, p2.y), std::fmax(p1.z, p2.z));
    }
    int maxExtent() const {  // :31-40
        V3 d = pMax - pMin;
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    V3 Centroid() const { return pMin * 0.5f + pMax * 0.5f; }  // :48
    // :93-118 -- float arithmetic whose results are merely *stored* in double
    bool IntersectP(const Ray& ray, V3 invDir, const int* dirIsNeg) const {
        double tMin = ((dirIsNeg[0] ? pMax.x : pMin.x) - ray.origin.x) * invDir.x;
        double tMax = (((1 - dirIsNeg[0]) ? pMax.x : pMin.x) - ray.origin.x) * invDir.x;
        double tyMin = ((dirIsNeg[1] ? pMax.y : pMin.y) - ray.origin.y) * invDir.y;
        double tyMax = (((1 - dirIsNeg[1]) ? pMax.y : pMin.y) - ray.origin.y) * invDir.y;
        if ((tMin > tyMax) || (tyMin > tMax)) return false;
        if (tyMin > tMin) tMin = tyMin;
        if (tyMax < tMax) tMax = tyMax;
        double tzMin = ((dirIsNeg[2] ? pMax.z : pMin.z) - ray.origin.z) * invDir.z;
        double tzMax = (((1 - dirIsNeg[2]) ? pMax.z : pMin.z) - ray.origin.z) * invDir.z;
        if ((tMin > tzMax) || (tzMin > tMax)) return false;
        return true;
    }
};
static inline Bounds3 Union(const Bounds3& a, const Bounds3& b) {  // :120-126
    Bounds3 r;
    r.pMin = gmin(a.pMin, b.pMin);
    r.pMax = gmax(a.pMax, b.pMax);
    return r;
}
static inline Bounds3 Union(const Bounds3& a, V3 p) {  // :128-134
    Bounds3 r;
    r.pMin = gmin(a.pMin, p);
    r.pMax = gmax(a.pMax, p);
    return r;
}

struct Intersection {
    bool happened = false;
    V3 coords = mk(0.f), normal = mk(0.f), emit = mk(0.f);
    double distance = kInfinity;
    float triangleArea = 0.f;
    const Material* m = nullptr;  // the reference copies the 40-byte Material by value
    int prim = -1;                // id tracking (not in the reference's struct)
};

// ----------------------------------------------------------------------------
// common/Triangle.hpp:42-85, :162-198; common/Triangle.cpp:5-61
// ----------------------------------------------------------------------------
struct Counters {
    uint64_t box = 0, tri = 0;
};

struct Triangle {
    V3 v0, v1, v2, e1, e2, normal;
    float area = 0.f;
    const Material* material = nullptr;
    int prim = -1;
    Triangle() {}
    Triangle(V3 a, V3 b, V3 c, const Material* m) : v0(a), v1(b), v2(c), material(m) {  // Triangle.hpp:52-59
        e1 = v1 - v0;
        e2 = v2 - v0;
        normal = normalize(cross(e1, e2));
        area = length(cross(e1, e2)) * 0.5f;
    }
    Bounds3 getBounds() const { return Union(Bounds3(v0, v1), v2); }  // :162
    Intersection getIntersection(const Ray& ray) const {               // :164-198
        Intersection inter;
        if (dot(ray.direction, normal) > 0) return inter;
        double u, v, t_tmp = 0;
        V3 pvec = cross(ray.direction, e2);
        double det = dot(e1, pvec);
        if (std::fabs(det) < kEpsilon5) return inter;
        double det_inv = 1. / det;
        V3 tvec = ray.origin - v0;
        u = dot(tvec, pvec) * det_inv;
        if (u < 0 || u > 1) return inter;
        V3 qvec = cross(tvec, e1);
        v = dot(ray.direction, qvec) * det_inv;
        if (v < 0 || u + v > 1) return inter;
        t_tmp = dot(e2, qvec) * det_inv;
        if (t_tmp < 0) return inter;
        inter.happened = true;
        inter.coords = ray.at(t_tmp);
        inter.normal = normal;
        inter.distance = t_tmp;
        inter.triangleArea = area;
        inter.m = material;
        inter.prim = prim;
        return inter;
    }
};

struct Mesh {
    std::vector<Triangle> triangles;
    Bounds3 bounding_box;
    float area = 0.f;
    Material material;
};

// common/MathUtils.hpp:86-100.  The unqualified `sqrt(discr)` there resolves to the C library's
// sqrt(double); sums and products with it and with the double literal -0.5 are evaluated in double
// and narrowed on assignment (checked against the compiled reference, 20 006 coefficient triples).
static inline bool solveQuadratic(const float& a, const float& b, const float& c, float& x0, float& x1) {
    float discr = b * b - 4 * a * c;
    if (discr < 0)
        return false;
    else if (discr == 0)
        x0 = x1 = -0.5 * b / a;
    else {
        float q = (b > 0) ? -0.5 * (b + std::sqrt((double)discr)) : -0.5 * (b - std::sqrt((double)discr));
        x0 = q / a;
        x1 = c / q;
    }
    if (x0 > x1) std::swap(x0, x1);
    return true;
}

// common/Sphere.hpp:13-101 -- the whole file is commented out in the reference and no scene can
// hold a sphere (Scene::Add takes MeshTriangle*), so this restates the commented text:
// getIntersection (:49-69) and getBounds (:81-84).  Parity unpinned except for solveQuadratic,
// which is live code.  A sphere is an object of the top-level BVH like a mesh; spheres do not emit.
struct Sphere {
    V3 center;
    float radius = 0.f, radius2 = 0.f, area = 0.f;
    Material material;
    int prim = -1;
    Sphere(V3 c, float r, const Material& mt) : center(c), radius(r), radius2(r * r), area(4 * kPi * r * r), material(mt) {}
    Bounds3 getBounds() const {
        return Bounds3(mk(center.x - radius, center.y - radius, center.z - radius), mk(center.x + radius, center.y + radius, center.z + radius));
    }
    Intersection getIntersection(const Ray& ray) const {
        Intersection result;
        V3 L = ray.origin - center;
        float a = dot(ray.direction, ray.direction);
        float b = 2 * dot(ray.direction, L);
        float c = dot(L, L) - radius2;
        float t0, t1;
        if (!solveQuadratic(a, b, c, t0, t1)) return result;
        if (t0 < 0) t0 = t1;
        if (t0 < 0) return result;
        result.happened = true;
        result.coords = ray.origin + ray.direction * t0;
        result.normal = normalize(result.coords - center);
        result.m = &material;
        result.distance = t0;
        result.triangleArea = area;
        result.prim = prim;
        return result;
    }
};

// ----------------------------------------------------------------------------
// common/BVH.hpp, common/BVH.cpp
// ----------------------------------------------------------------------------
struct Bvh;
struct Node {
    Bounds3 bounds;
    Node* left = nullptr;
    Node* right = nullptr;
    int mesh = -1;           // top level leaf: mesh index
    Bvh* meshBvhRoot = nullptr;
    const Triangle* triangle = nullptr;
    const Sphere* sphere = nullptr;  // top level leaf: a sphere
    float area = 0.f;
    Node* nextIfHit = nullptr;
    Node* nextIfMiss = nullptr;
    ~Node();
};

// a BVH is built over "objects": either all meshes of a scene (top level) or
// the triangles of one mesh (nested level); Obj is that tagged reference
struct Obj {
    int mesh = -1;                       // >= 0: a MeshTriangle
    const Triangle* triangle = nullptr;  // else: a Triangle
    const Sphere* sphere = nullptr;      // or a Sphere
};

struct Scene;
struct Bvh {
    Node* root = nullptr;
    Bvh(const Scene& scene, std::vector<Obj>& p, std::vector<Bvh*>& meshBvhMap);
    ~Bvh() { delete root; }
    Node* recursiveBuild(const Scene& scene, std::vector<Obj>& objects, std::vector<Bvh*>& meshBvhMap);
    Intersection getIntersection(const Ray& ray, Counters& cnt) const;
};
Node::~Node() {
    delete left;
    delete right;
    delete meshBvhRoot;
}

struct Scene {
    int width = 1280, height = 960;
    double fov = 40;
    V3 camPos = mk(0.f);
    std::vector<Mesh*> meshes;
    std::vector<Sphere*> spheres;
    std::vector<Obj> objects;  // meshes and spheres in Scene::Add order
    Bvh* bvh = nullptr;
    std::vector<Bvh*> mesh_bvhs;
    std::vector<int> mesh_base;
    float sumLightArea = 0.f, sumLightPower = 0.f;
    int num_lights = 0;
    // the integrator's switches, compile-time `false` in the reference (Scene.hpp:141-142)
    bool pathRegularization = false, enableRR = false;
    float RussianRoulette = 0.8f;  // Scene.hpp:30
    ~Scene() {
        delete bvh;
        for (Mesh* m : meshes) delete m;
        for (Sphere* sp : spheres) delete sp;
    }
    void Add(Sphere* sphere) {
        Obj o;
        o.sphere = sphere;
        objects.push_back(o);
        spheres.push_back(sphere);
    }
    void Add(Mesh* mesh) {  // Scene.hpp:47-56
        Obj mo;
        mo.mesh = (int)meshes.size();
        objects.push_back(mo);
        meshes.push_back(mesh);
        if (mesh->material.emitting()) {
            sumLightArea += mesh->area;
            sumLightPower += mesh->area * luminance(mesh->material._emission) * 2.f * 3.14159265358979323846264338327950288f;
            ++num_lights;
        }
    }
    void buildBVH() {  // Scene.cpp:9-23
        // primitive ids count through the objects in Add order (a sphere is one primitive)
        mesh_base.assign(meshes.size(), 0);
        int base = 0;
        for (const Obj& o : objects) {
            if (o.sphere) {
                const_cast<Sphere*>(o.sphere)->prim = base++;
                continue;
            }
            Mesh* m = meshes[o.mesh];
            mesh_base[o.mesh] = base;
            for (size_t k = 0; k < m->triangles.size(); ++k) {
                m->triangles[k].prim = base + (int)k;
                m->triangles[k].material = &m->material;
            }
            base += (int)m->triangles.size();
        }
        delete bvh;
        mesh_bvhs.assign(meshes.size(), nullptr);
        std::vector<Obj> top(objects);
        bvh = new Bvh(*this, top, mesh_bvhs);
    }
    Bounds3 boundsOf(const Obj& o) const { return o.sphere ? o.sphere->getBounds() : o.mesh >= 0 ? meshes[o.mesh]->bounding_box : o.triangle->getBounds(); }
    float areaOf(const Obj& o) const { return o.sphere ? o.sphere->area : o.mesh >= 0 ? meshes[o.mesh]->area : o.triangle->area; }
};

static void preorder(Node* n, std::vector<Node*>& out) {  // BVH.cpp:8-15
    if (!n) return;
    out.push_back(n);
    preorder(n->left, out);
    preorder(n->right, out);
}
static void SetNextIfMiss(Node* node, Node* val) {  // BVH.cpp:22-37
    if (!node->left && !node->right) return;
    node->left->nextIfMiss = node->right;
    node->right->nextIfMiss = val;
    SetNextIfMiss(node->left, node->left->nextIfMiss);
    SetNextIfMiss(node->right, node->right->nextIfMiss);
}

Bvh::Bvh(const Scene& scene, std::vector<Obj>& p, std::vector<Bvh*>& meshBvhMap) {  // BVH.cpp:39-77
    if (p.empty()) return;
    root = recursiveBuild(scene, p, meshBvhMap);
    std::vector<Node*> nodes;
    preorder(root, nodes);
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i]->nextIfHit = (i + 1 == nodes.size()) ? nullptr : nodes[i + 1];
    SetNextIfMiss(root, nullptr);
}

Node* Bvh::recursiveBuild(const Scene& scene, std::vector<Obj>& objects, std::vector<Bvh*>& meshBvhMap) {  // BVH.cpp:81-168
    Node* node = new Node();
    if (objects.size() == 1) {
        node->bounds = scene.boundsOf(objects[0]);
        if (objects[0].sphere) {
            node->sphere = objects[0].sphere;
        } else if (objects[0].mesh >= 0) {
            node->mesh = objects[0].mesh;
            Mesh* mesh = scene.meshes[objects[0].mesh];
            std::vector<Obj> triangles(mesh->triangles.size());
            for (size_t i = 0; i < mesh->triangles.size(); ++i) triangles[i].triangle = &mesh->triangles[i];
            node->meshBvhRoot = new Bvh(scene, triangles, meshBvhMap);
            meshBvhMap[objects[0].mesh] = node->meshBvhRoot;
        } else {
            node->triangle = objects[0].triangle;
        }
        node->area = scene.areaOf(objects[0]);
        return node;
    } else if (objects.size() == 2) {
        std::vector<Obj> l{objects[0]}, r{objects[1]};
        node->left = recursiveBuild(scene, l, meshBvhMap);
        node->right = recursiveBuild(scene, r, meshBvhMap);
        node->bounds = Union(node->left->bounds, node->right->bounds);
        node->area = node->left->area + node->right->area;
        return node;
    } else {
        Bounds3 centroidBounds;
        for (size_t i = 0; i < objects.size(); ++i) centroidBounds = Union(centroidBounds, scene.boundsOf(objects[i]).Centroid());
        int dim = centroidBounds.maxExtent();
        const Scene* sc = &scene;
        switch (dim) {
        case 0: std::sort(objects.begin(), objects.end(), [sc](const Obj& f1, const Obj& f2) { return sc->boundsOf(f1).Centroid().x < sc->boundsOf(f2).Centroid().x; }); break;
        case 1: std::sort(objects.begin(), objects.end(), [sc](const Obj& f1, const Obj& f2) { return sc->boundsOf(f1).Centroid().y < sc->boundsOf(f2).Centroid().y; }); break;
        case 2: std::sort(objects.begin(), objects.end(), [sc](const Obj& f1, const Obj& f2) { return sc->boundsOf(f1).Centroid().z < sc->boundsOf(f2).Centroid().z; }); break;
        }
        auto mid = objects.begin() + (objects.size() / 2);
        std::vector<Obj> l(objects.begin(), mid), r(mid, objects.end());
        node->left = recursiveBuild(scene, l, meshBvhMap);
        node->right = recursiveBuild(scene, r, meshBvhMap);
        node->bounds = Union(node->left->bounds, node->right->bounds);
        node->area = node->left->area + node->right->area;
    }
    return node;
}

Intersection Bvh::getIntersection(const Ray& ray, Counters& cnt) const {  // BVH.hpp:82-145
    Intersection isect;
    if (!root) return isect;
    const int dirIsNeg[3] = {ray.direction.x < 0, ray.direction.y < 0, ray.direction.z < 0};
    Node* current = root;
    while (current) {
        ++cnt.box;
        bool boundsHit = current->bounds.IntersectP(ray, ray.direction_inv, dirIsNeg);
        if (boundsHit) {
            if (!current->left && !current->right) {
                Intersection hit;
                if (current->mesh >= 0) hit = current->meshBvhRoot->getIntersection(ray, cnt);
                else if (current->sphere) { ++cnt.tri; hit = current->sphere->getIntersection(ray); }
                else { ++cnt.tri; hit = current->triangle->getIntersection(ray); }
                if (hit.happened) {
                    if (ray.shadowTest) {
                        if (hit.distance < ray.shadowDistance - kEpsilon4) return hit;
                    } else {
                        if (hit.distance < isect.distance) isect = hit;
                    }
                }
            }
            current = current->nextIfHit;
        } else {
            current = current->nextIfMiss;
        }
    }
    return isect;
}

// BVH.hpp:147-184 + Triangle.hpp:75-82
static void bvhSample(const Bvh* bvh, Rng& rng, uint32_t depth, Intersection& pos, float& pdf) {
    float p = std::sqrt(rng.draw(depth, kSlotTriPick)) * bvh->root->area;
    Node* current = bvh->root;
    while (current) {
        if (!current->left && !current->right) {
            const Triangle* t = current->triangle;
            float x = std::sqrt(rng.draw(depth, kSlotTriX)), y = rng.draw(depth, kSlotTriY);
            pos.coords = t->v0 * (1.0f - x) + t->v1 * (x * (1.0f - y)) + t->v2 * (x * y);
            pos.normal = t->normal;
            pos.triangleArea = t->area;
            break;
        }
        if (p < current->left->area) current = current->left;
        else { p -= current->left->area; current = current->right; }
    }
    pdf = 1.0f / bvh->root->area;
}

// Scene.hpp:99-120
static void sampleLight(const Scene& sc, Rng& rng, uint32_t depth, Intersection& pos, float& pdf) {
    float emit_area_sum = sc.sumLightArea;
    float p = rng.draw(depth, kSlotLightPick) * emit_area_sum;
    emit_area_sum = 0;
    for (size_t k = 0; k < sc.meshes.size(); ++k) {
        Mesh* mesh = sc.meshes[k];
        if (mesh->material.emitting()) {
            emit_area_sum += mesh->area;
            if (p <= emit_area_sum) {
                bvhSample(sc.mesh_bvhs[k], rng, depth, pos, pdf);
                pos.emit = mesh->material._emission;
                break;
            }
        }
    }
}

// Ray.hpp:36-52
struct PathSegment {
    Rng rng;
    Ray ray;
    V3 throughput = mk(1.f), radiance = mk(0.f);
    int remainingBounces = 0;
    bool specularBounce = false;
    bool anyNonSpecularBounces = false;
    float bsdfSamplePdf = 0.0f;
    int depth = 0;
};

struct Tally {
    orc_stats s;
    Counters cnt;
    Tally() { std::memset(&s, 0, sizeof s); }
};

static Intersection sceneIntersect(const Scene& sc, const Ray& ray, Tally& tl) {  // Scene.hpp:95-97
    if (ray.shadowTest) ++tl.s.shadow_queries; else ++tl.s.closest_queries;
    return sc.bvh->getIntersection(ray, tl.cnt);
}

// Scene.hpp:122-272 (pathRegularization and enableRR are compile-time false there; switches of Scene here)
static void TracePath(const Scene& sc, const Trig& tg, PathSegment& path, Intersection& intersec, Tally& tl) {
    V3& accRadiance = path.radiance;
    V3& throughput = path.throughput;
    Ray& ray = path.ray;
    const uint32_t d = (uint32_t)path.depth;
    do {
        if (!intersec.happened) {  // :156-161
            path.remainingBounces = 0;
            break;
        }
        Material material = *intersec.m;  // the reference's Intersection holds the Material by value
        if (sc.pathRegularization && path.anyNonSpecularBounces) material.regularize();  // :186-189
        if (material.emitting()) {  // :164-184
            if (path.depth == 0 || path.specularBounce) {
                accRadiance = accRadiance + throughput * material._emission;
            } else {
                V3 radiance = material._emission;
                float lightPdf = luminance(radiance) * 2.f * 3.14159265358979323846264338327950288f * intersec.triangleArea / sc.sumLightPower;
                float weight = powerHeuristic(path.bsdfSamplePdf, lightPdf);
                accRadiance = accRadiance + throughput * radiance * weight;
            }
        }
        if (!path.specularBounce) {  // :192-228
            Intersection lightSample;
            float lightSamplePdf = 0.0f;
            sampleLight(sc, path.rng, d, lightSample, lightSamplePdf);
            if (lightSamplePdf > 0.0f) {
                V3 p = intersec.coords;
                V3 x = lightSample.coords;
                V3 wo = -ray.direction;
                V3 px = x - p;
                V3 wi = normalize(px);
                Ray shadowRay(p + wi * kEpsilon5, wi, true, length(px));
                Intersection shadowIntersec = sceneIntersect(sc, shadowRay, tl);
                if (!shadowIntersec.happened) {
                    V3 radiance = lightSample.emit;
                    V3 bsdf = material.bsdf(wi, wo, intersec.normal);
                    float cos_theta = satDot(intersec.normal, wi);
                    float cos_theta_prime = satDot(lightSample.normal, -wi);
                    float r2 = dot(x - p, x - p);
                    float bsdfPdf = material.pdf(wi, wo, intersec.normal);
                    float lightPdf = luminance(radiance) * 2.f * 3.14159265358979323846264338327950288f * lightSample.triangleArea / sc.sumLightPower;
                    float weight = powerHeuristic(lightPdf, bsdfPdf);
                    accRadiance = accRadiance + throughput * radiance * bsdf * cos_theta * cos_theta_prime / r2 / lightSamplePdf * weight;
                }
            }
        }
        // :230-250
        V3 wo = -ray.direction;
        V3 wi = material.sample(tg, path.rng, d, wo, intersec.normal);
        V3 p = intersec.coords;
        V3 bsdf = material.bsdf(wi, wo, intersec.normal);
        path.bsdfSamplePdf = material.pdf(wi, wo, intersec.normal);
        if (path.bsdfSamplePdf < kEpsilon5) {
            ++tl.s.stalled_bounces;
            break;
        }
        path.specularBounce = material.hasSpecular();
        path.anyNonSpecularBounces |= !path.specularBounce;
        float cos_theta = absDot(intersec.normal, wi);
        throughput = throughput * (bsdf * cos_theta / path.bsdfSamplePdf);
        ray = Ray(p + wi * kEpsilon5, wi);
        // :252-264, as written: the `break` only leaves the reference's while(true) wrapper, it does
        // not end the path -- a "killed" bounce merely goes without the 1 / (1 - q) compensation
        if (sc.enableRR) {
            float rrProb = path.rng.draw(d, kSlotRoulette);
            if (path.depth > 1) {
                float q = 1 - sc.RussianRoulette;
                if (rrProb < q) break;
                throughput = throughput / (1.f - q);
            }
        }
    } while (false);

    path.depth++;  // :268-271
    if (std::isnan(accRadiance.x) || std::isnan(accRadiance.y) || std::isnan(accRadiance.z) || std::isinf(accRadiance.x) ||
        std::isinf(accRadiance.y) || std::isinf(accRadiance.z)) {
        accRadiance = mk(0.0f);
        ++tl.s.nan_resets;
    }
}

// Scene.hpp:274-285
static V3 castRay(const Scene& sc, const Trig& tg, const Rng& rng, const Ray& eyeRay, int maxDepth, Tally& tl) {
    PathSegment path;
    path.rng = rng;
    path.ray = eyeRay;
    path.remainingBounces = maxDepth;
    for (int depth = 0; depth < maxDepth; depth++) {
        if (path.remainingBounces <= 0) break;
        Intersection intersec = sceneIntersect(sc, path.ray, tl);
        path.remainingBounces -= 1;
        TracePath(sc, tg, path, intersec, tl);
    }
    ++tl.s.paths;
    return path.radiance;
}

// Renderer.cpp:11,20: deg2rad takes a const float&, multiplies in float,
// divides by the double 180.0, returns float; tan() is the double overload.
static float cameraScale(double fov) {
    const float half_deg = fov * 0.5;
    const float half_rad = half_deg * kPi / 180.0;
    return tan(half_rad);
}

// Renderer.cpp:32-40
static V3 pixelSample(const Scene& sc, const Trig& tg, int rng_mode, uint32_t seed, int maxDepth, float scale, int k, int m, Tally& tl) {
    uint32_t i = (uint32_t)m % (uint32_t)sc.width, j = (uint32_t)m / (uint32_t)sc.width;
    float imageAspectRatio = sc.width / (float)sc.height;
    Rng rng(rng_mode, seed, k, m);
    float bias = rng.draw(kCameraDepth, 0);
    float x = (2 * (i + bias) / (float)sc.width - 1) * imageAspectRatio * scale;
    float y = (1 - 2 * (j + bias) / (float)sc.height) * scale;
    V3 dir = normalize(mk(-x, y, 1));
    return castRay(sc, tg, rng, Ray(sc.camPos, dir), maxDepth, tl);
}

static void mergeStats(orc_stats* dst, const Tally& tl) {
    if (!dst) return;
#pragma omp critical(orc_stats_merge)
    {
        dst->closest_queries += tl.s.closest_queries;
        dst->shadow_queries += tl.s.shadow_queries;
        dst->box_tests += tl.cnt.box;
        dst->tri_tests += tl.cnt.tri;
        dst->paths += tl.s.paths;
        dst->stalled_bounces += tl.s.stalled_bounces;
        dst->nan_resets += tl.s.nan_resets;
    }
}

// minimal Wavefront reader: `v x y z` and triangular `f a b c` (a, a/b, a//c,
// a/b/c forms), one mesh per file, three un-indexed vertices per face -- the
// subset of the vendored loader's behaviour MeshTriangle relies on
// (common/OBJ_Loader.hpp:431-640, common/Triangle.cpp:28-53)
static bool loadObj(const char* path, std::vector<V3>& face_vertices) {
    FILE* fp = std::fopen(path, "r");
    if (!fp) return false;
    std::vector<V3> verts;
    char line[1024];
    bool ok = true;
    while (std::fgets(line, sizeof line, fp)) {
        char* s = line;
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* e = s + 1;
            float c[3];
            for (int k = 0; k < 3; ++k) c[k] = std::strtof(e, &e);
            verts.push_back(mk(c[0], c[1], c[2]));
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            char* e = s + 1;
            long idx[4];
            int cnt = 0;
            while (cnt < 4) {
                while (*e == ' ' || *e == '\t') ++e;
                if (*e == '\0' || *e == '\n' || *e == '\r') break;
                idx[cnt++] = std::strtol(e, &e, 10);
                while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') ++e;
            }
            if (cnt != 3) { ok = false; break; }
            for (int k = 0; k < 3; ++k) {
                long id = idx[k] < 0 ? (long)verts.size() + idx[k] : idx[k] - 1;
                if (id < 0 || id >= (long)verts.size()) { ok = false; break; }
                face_vertices.push_back(verts[id]);
            }
            if (!ok) break;
        }
    }
    std::fclose(fp);
    return ok;
}

// common/Triangle.cpp:5-61
static Mesh* makeMesh(const std::vector<V3>& fv, const Material& mt) {
    Mesh* mesh = new Mesh();
    mesh->material = mt;
    mesh->area = 0;
    const float inf = std::numeric_limits<float>::infinity();
    V3 min_vert = mk(inf), max_vert = mk(-inf);
    for (size_t i = 0; i + 2 < fv.size(); i += 3) {
        for (int j = 0; j < 3; ++j) {
            V3 vert = fv[i + j];
            min_vert = mk(std::min(min_vert.x, vert.x), std::min(min_vert.y, vert.y), std::min(min_vert.z, vert.z));
            max_vert = mk(std::max(max_vert.x, vert.x), std::max(max_vert.y, vert.y), std::max(max_vert.z, vert.z));
        }
        mesh->triangles.push_back(Triangle(fv[i], fv[i + 1], fv[i + 2], nullptr));
        mesh->area += mesh->triangles.back().area;
    }
    mesh->bounding_box = Bounds3(min_vert, max_vert);
    return mesh;
}

}  // namespace orc

struct orc_scene {
    orc::Scene sc;
};

using namespace orc;

extern "C" {

orc_scene* orc_scene_create(int width, int height, const float cam_pos[3], double fov_deg) {
    orc_scene* s = new orc_scene();
    s->sc.width = width;
    s->sc.height = height;
    s->sc.camPos = mk(cam_pos[0], cam_pos[1], cam_pos[2]);
    s->sc.fov = fov_deg;
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_scene_add_obj(orc_scene* s, const char* obj_path, const orc_material* m) {
    std::vector<V3> fv;
    if (!loadObj(obj_path, fv)) return -1;
    s->sc.Add(makeMesh(fv, from_c(m)));
    return (int)s->sc.meshes.size() - 1;
}
int orc_scene_add_triangles(orc_scene* s, const float* verts9, int num_tris, const orc_material* m) {
    std::vector<V3> fv;
    for (int i = 0; i < num_tris * 3; ++i) fv.push_back(mk(verts9[3 * i], verts9[3 * i + 1], verts9[3 * i + 2]));
    s->sc.Add(makeMesh(fv, from_c(m)));
    return (int)s->sc.meshes.size() - 1;
}
int orc_scene_add_sphere(orc_scene* s, const float center[3], float radius, const orc_material* m) {
    s->sc.Add(new Sphere(mk(center[0], center[1], center[2]), radius, from_c(m)));
    return (int)s->sc.spheres.size() - 1;
}
void orc_scene_set_integrator(orc_scene* s, int russian_roulette, int path_regularization) {
    s->sc.enableRR = russian_roulette != 0;
    s->sc.pathRegularization = path_regularization != 0;
}
float orc_material_regularize(float roughness) {
    Material m;
    m._roughness = roughness;
    m.regularize();
    return m._roughness;
}
int orc_solve_quadratic(float a, float b, float c, float* x0, float* x1) { return solveQuadratic(a, b, c, *x0, *x1) ? 1 : 0; }
int orc_scene_build(orc_scene* s) {
    s->sc.buildBVH();
    return 0;
}
void orc_scene_set_resolution(orc_scene* s, int width, int height) {
    s->sc.width = width;
    s->sc.height = height;
}

orc_scene* orc_scene_create_cornell(const char* models_dir, int width, int height) {  // Scene.cpp:25-73
    const float cam[3] = {278, 273, -800};
    orc_scene* s = orc_scene_create(width > 0 ? width : 512, height > 0 ? height : 512, cam, 40);
    Material red, green, white, light, metal, glass;
    red._albedo = mk(0.63f, 0.065f, 0.05f);
    green._albedo = mk(0.14f, 0.45f, 0.091f);
    white._albedo = mk(0.725f, 0.71f, 0.68f);
    light._emission = (8.0f * mk(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) + 15.6f * mk(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                       18.4f * mk(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));
    metal.type = Material::MetallicWorkflow;
    metal._albedo = mk((float)0.7, (float)0.5, (float)0.2);
    metal._metallic = 1.f;
    metal._roughness = 0.005f;
    glass.type = Material::Lambertian;
    glass._roughness = 0.2f;
    glass._ior = 1.5f;
    const char* files[6] = {"floor", "shortbox", "tallbox", "left", "right", "light"};
    const Material* mats[6] = {&white, &glass, &metal, &red, &green, &light};
    for (int i = 0; i < 6; ++i) {
        std::string path = std::string(models_dir) + "/cornellbox/" + files[i] + ".obj";
        std::vector<V3> fv;
        if (!loadObj(path.c_str(), fv)) {
            delete s;
            return nullptr;
        }
        s->sc.Add(makeMesh(fv, *mats[i]));
    }
    s->sc.buildBVH();
    return s;
}

void orc_scene_info(const orc_scene* s, int* num_meshes, int* num_tris, int* width, int* height, float* sum_light_area,
                    float* sum_light_power, int* num_lights, float* cam_scale) {
    *num_meshes = (int)s->sc.meshes.size();
    int t = 0;
    for (Mesh* m : s->sc.meshes) t += (int)m->triangles.size();
    *num_tris = t;
    *width = s->sc.width;
    *height = s->sc.height;
    *sum_light_area = s->sc.sumLightArea;
    *sum_light_power = s->sc.sumLightPower;
    *num_lights = s->sc.num_lights;
    *cam_scale = cameraScale(s->sc.fov);
}

void orc_mesh_info(const orc_scene* s, int mesh, int* num_tris, float* area, float* bbox6, float* bvh_root_area) {
    const Mesh* m = s->sc.meshes[mesh];
    *num_tris = (int)m->triangles.size();
    *area = m->area;
    bbox6[0] = m->bounding_box.pMin.x; bbox6[1] = m->bounding_box.pMin.y; bbox6[2] = m->bounding_box.pMin.z;
    bbox6[3] = m->bounding_box.pMax.x; bbox6[4] = m->bounding_box.pMax.y; bbox6[5] = m->bounding_box.pMax.z;
    *bvh_root_area = s->sc.mesh_bvhs[mesh]->root->area;
}

void orc_mesh_triangles(const orc_scene* s, int mesh, float* out13) {
    const Mesh* m = s->sc.meshes[mesh];
    for (size_t i = 0; i < m->triangles.size(); ++i) {
        const Triangle& t = m->triangles[i];
        float* o = out13 + 13 * i;
        o[0] = t.v0.x; o[1] = t.v0.y; o[2] = t.v0.z;
        o[3] = t.e1.x; o[4] = t.e1.y; o[5] = t.e1.z;
        o[6] = t.e2.x; o[7] = t.e2.y; o[8] = t.e2.z;
        o[9] = t.normal.x; o[10] = t.normal.y; o[11] = t.normal.z;
        o[12] = t.area;
    }
}

int orc_bvh_dump(const orc_scene* s, int mesh, float* bounds6, int* payload, float* area, int* miss_rank) {
    const Bvh* bvh = mesh < 0 ? s->sc.bvh : s->sc.mesh_bvhs[mesh];
    std::vector<Node*> nodes;
    preorder(bvh->root, nodes);
    if (!bounds6) return (int)nodes.size();
    std::unordered_map<const Node*, int> rank_of;
    for (size_t i = 0; i < nodes.size(); ++i) rank_of[nodes[i]] = (int)i;
    for (size_t i = 0; i < nodes.size(); ++i) {
        Node* n = nodes[i];
        bounds6[6 * i + 0] = n->bounds.pMin.x; bounds6[6 * i + 1] = n->bounds.pMin.y; bounds6[6 * i + 2] = n->bounds.pMin.z;
        bounds6[6 * i + 3] = n->bounds.pMax.x; bounds6[6 * i + 4] = n->bounds.pMax.y; bounds6[6 * i + 5] = n->bounds.pMax.z;
        bool leaf = !n->left && !n->right;
        if (!leaf) payload[i] = -1;
        else if (n->mesh >= 0 || n->sphere) {  // top level: the object's position in Scene::Add order
            payload[i] = -1;
            for (size_t k = 0; k < s->sc.objects.size(); ++k)
                if ((n->sphere && s->sc.objects[k].sphere == n->sphere) || (!n->sphere && s->sc.objects[k].mesh == n->mesh && !s->sc.objects[k].sphere)) payload[i] = (int)k;
        }
        else payload[i] = (int)(n->triangle - s->sc.meshes[mesh]->triangles.data());
        area[i] = n->area;
        miss_rank[i] = n->nextIfMiss ? rank_of.at(n->nextIfMiss) : (int)nodes.size();
    }
    return (int)nodes.size();
}

void orc_intersect_batch(const orc_scene* s, int64_t n, const float* o, const float* d, const uint8_t* shadow,
                         const double* tmax, int32_t* prim, double* t, int32_t* counts, int threads) {
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t r = 0; r < n; ++r) {
        V3 ro = mk(o[3 * r], o[3 * r + 1], o[3 * r + 2]);
        V3 rd = mk(d[3 * r], d[3 * r + 1], d[3 * r + 2]);
        bool sh = shadow && shadow[r];
        Ray ray = sh ? Ray(ro, rd, true, tmax[r]) : Ray(ro, rd);
        Counters cnt;
        Intersection hit = s->sc.bvh->getIntersection(ray, cnt);
        prim[r] = hit.happened ? hit.prim : -1;
        t[r] = hit.distance;
        if (counts) {
            counts[2 * r] = (int32_t)cnt.box;
            counts[2 * r + 1] = (int32_t)cnt.tri;
        }
    }
}

void orc_render(const orc_scene* s, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int sample_begin,
                int sample_count, int spp_total, int threads, float* rgb_out, int64_t* accum_out, orc_stats* stats) {
    const Scene& sc = s->sc;
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    const float scale = cameraScale(sc.fov);
    if (threads <= 0) threads = omp_get_max_threads();
    if (stats) std::memset(stats, 0, sizeof *stats);
    const int P = sc.width * sc.height;
    for (int k = sample_begin; k < sample_begin + sample_count; ++k) {  // Renderer.cpp:28
#pragma omp parallel num_threads(threads)
        {
            Tally tl;
#pragma omp for schedule(dynamic, 256)
            for (int m = 0; m < P; ++m) {
                V3 L = pixelSample(sc, tg, rng_mode, seed, max_depth, scale, k, m, tl);
                if (rgb_out) {
                    V3 q = L / (float)spp_total;  // Renderer.cpp:40
                    rgb_out[3 * m] += q.x; rgb_out[3 * m + 1] += q.y; rgb_out[3 * m + 2] += q.z;
                }
                if (accum_out) {
                    accum_out[3 * m] += radiance_to_fixed(L.x);
                    accum_out[3 * m + 1] += radiance_to_fixed(L.y);
                    accum_out[3 * m + 2] += radiance_to_fixed(L.z);
                }
            }
            mergeStats(stats, tl);
        }
    }
}

void orc_cast_paths(const orc_scene* s, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int64_t n,
                    const int32_t* sample_k, const int32_t* pixel_m, float* rgb_out, int threads, orc_stats* stats) {
    const Scene& sc = s->sc;
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    const float scale = cameraScale(sc.fov);
    if (threads <= 0) threads = omp_get_max_threads();
    if (stats) std::memset(stats, 0, sizeof *stats);
#pragma omp parallel num_threads(threads)
    {
        Tally tl;
#pragma omp for schedule(dynamic, 256)
        for (int64_t q = 0; q < n; ++q) {
            V3 L = pixelSample(sc, tg, rng_mode, seed, max_depth, scale, sample_k[q], pixel_m[q], tl);
            rgb_out[3 * q] = L.x; rgb_out[3 * q + 1] = L.y; rgb_out[3 * q + 2] = L.z;
        }
        mergeStats(stats, tl);
    }
}

// Scene::castRay (common/Scene.hpp:274-285) for caller-supplied eye rays.  The random stream of ray q is
// Rng(rng_mode, seed, sample_k[q], pixel_m[q]) from its beginning: with the keyed Philox stream that is the stream of
// the camera path of that pixel and sample (draws are addressed by bounce and slot, the camera's draw has a bounce
// number of its own), so a ray that equals that path's camera ray returns that path's radiance.
void orc_cast_rays(const orc_scene* s, int rng_mode, int trig_mode, uint32_t seed, int max_depth, int64_t n, const float* o,
                   const float* d, const int32_t* sample_k, const int32_t* pixel_m, float* rgb_out, int threads, orc_stats* stats) {
    const Scene& sc = s->sc;
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    if (threads <= 0) threads = omp_get_max_threads();
    if (stats) std::memset(stats, 0, sizeof *stats);
#pragma omp parallel num_threads(threads)
    {
        Tally tl;
#pragma omp for schedule(dynamic, 256)
        for (int64_t q = 0; q < n; ++q) {
            Rng rng(rng_mode, seed, sample_k[q], pixel_m[q]);
            V3 L = castRay(sc, tg, rng, Ray(mk(o[3 * q], o[3 * q + 1], o[3 * q + 2]), mk(d[3 * q], d[3 * q + 1], d[3 * q + 2])), max_depth, tl);
            rgb_out[3 * q] = L.x; rgb_out[3 * q + 1] = L.y; rgb_out[3 * q + 2] = L.z;
        }
        mergeStats(stats, tl);
    }
}

// The rays that camera paths trace at bounce `bounce` (0 = the camera rays), as Scene::castRay (common/Scene.hpp:274-285)
// hands them to Scene::intersect: an incoherent ray batch with the directions and origins a render really produces.
// alive_out[q] = 0 when path q ended before that bounce (its o/d are left untouched).
void orc_capture_bounce_rays(const orc_scene* s, int rng_mode, int trig_mode, uint32_t seed, int bounce, int64_t n,
                             const int32_t* sample_k, const int32_t* pixel_m, float* o_out, float* d_out, uint8_t* alive_out, int threads) {
    const Scene& sc = s->sc;
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    const float scale = cameraScale(sc.fov);
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        Tally tl;
#pragma omp for schedule(dynamic, 256)
        for (int64_t q = 0; q < n; ++q) {
            // pixelSample (Renderer.cpp:32-40), then castRay's loop cut off at `bounce`
            const int k = sample_k[q], m = pixel_m[q];
            uint32_t i = (uint32_t)m % (uint32_t)sc.width, j = (uint32_t)m / (uint32_t)sc.width;
            float imageAspectRatio = sc.width / (float)sc.height;
            Rng rng(rng_mode, seed, k, m);
            float bias = rng.draw(kCameraDepth, 0);
            float x = (2 * (i + bias) / (float)sc.width - 1) * imageAspectRatio * scale;
            float y = (1 - 2 * (j + bias) / (float)sc.height) * scale;
            PathSegment path;
            path.rng = rng;
            path.ray = Ray(sc.camPos, normalize(mk(-x, y, 1)));
            path.remainingBounces = bounce + 1;
            for (int depth = 0; depth < bounce; depth++) {
                if (path.remainingBounces <= 0) break;
                Intersection intersec = sceneIntersect(sc, path.ray, tl);
                path.remainingBounces -= 1;
                TracePath(sc, tg, path, intersec, tl);
            }
            const bool alive = path.remainingBounces > 0;
            alive_out[q] = alive ? 1 : 0;
            if (alive) {
                o_out[3 * q] = path.ray.origin.x; o_out[3 * q + 1] = path.ray.origin.y; o_out[3 * q + 2] = path.ray.origin.z;
                d_out[3 * q] = path.ray.direction.x; d_out[3 * q + 1] = path.ray.direction.y; d_out[3 * q + 2] = path.ray.direction.z;
            }
        }
    }
}

void orc_rng_floats(int rng_mode, uint32_t seed, int iter, int index, int count, float* out) {
    Rng rng(rng_mode, seed, iter, index);
    for (int i = 0; i < count; ++i) out[i] = rng.draw(0, (uint32_t)i);
}

void orc_material_eval(const orc_material* cm, int64_t n, const float* wi, const float* wo, const float* nrm,
                       float* bsdf_out, float* pdf_out) {
    Material m = from_c(cm);
    for (int64_t q = 0; q < n; ++q) {
        V3 a = mk(wi[3 * q], wi[3 * q + 1], wi[3 * q + 2]);
        V3 b = mk(wo[3 * q], wo[3 * q + 1], wo[3 * q + 2]);
        V3 c = mk(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]);
        V3 f = m.bsdf(a, b, c);
        bsdf_out[3 * q] = f.x; bsdf_out[3 * q + 1] = f.y; bsdf_out[3 * q + 2] = f.z;
        pdf_out[q] = m.pdf(a, b, c);
    }
}

void orc_material_sample(const orc_material* cm, int trig_mode, int64_t n, const float* u3, const float* wo,
                         const float* nrm, float* wi_out) {
    Material m = from_c(cm);
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    for (int64_t q = 0; q < n; ++q) {
        Rng rng;
        rng.fixed = u3 + 3 * q;
        V3 b = mk(wo[3 * q], wo[3 * q + 1], wo[3 * q + 2]);
        V3 c = mk(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]);
        V3 w = m.sample(tg, rng, 0, b, c);
        wi_out[3 * q] = w.x; wi_out[3 * q + 1] = w.y; wi_out[3 * q + 2] = w.z;
    }
}

void orc_sample_hemisphere_cosine(int trig_mode, int64_t n, const float* nrm, const float* rx, const float* ry,
                                  float* out) {
    Trig tg;
    tg.portable = trig_mode == ORC_TRIG_PORTABLE;
    for (int64_t q = 0; q < n; ++q) {
        V3 w = sampleHemisphereCosine(tg, mk(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]), rx[q], ry[q]);
        out[3 * q] = w.x; out[3 * q + 1] = w.y; out[3 * q + 2] = w.z;
    }
}

void orc_sample_light(const orc_scene* s, int64_t n, const float* u4, float* out11) {
    for (int64_t q = 0; q < n; ++q) {
        Rng rng;
        rng.fixed = u4 + 4 * q;
        Intersection pos;
        float pdf = 0.f;
        sampleLight(s->sc, rng, 0, pos, pdf);
        float* o = out11 + 11 * q;
        o[0] = pos.coords.x; o[1] = pos.coords.y; o[2] = pos.coords.z;
        o[3] = pos.normal.x; o[4] = pos.normal.y; o[5] = pos.normal.z;
        o[6] = pos.emit.x; o[7] = pos.emit.y; o[8] = pos.emit.z;
        o[9] = pos.triangleArea;
        o[10] = pdf;
    }
}

float orc_power_heuristic(float f, float g) { return powerHeuristic(f, g); }

// Utility.cpp:16: (unsigned char)(255 * glm::pow(clamp(0, 1, v), 0.6f))
int orc_tonemap(float v) { return (unsigned char)(255 * std::pow(clampf(0, 1, v), 0.6f)); }

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
float orc_philox_uniform(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot, uint32_t seed) {
    return philox_uniform(pixel, sample, depth, slot, seed);
}
int64_t orc_radiance_to_fixed(float v) { return radiance_to_fixed(v); }
float orc_fixed_to_radiance(int64_t acc, int spp_total) { return fixed_to_radiance(acc, spp_total); }
void orc_portable_trig(int64_t n, const float* x, float* sin_out, float* cos_out, float* acos_out) {
    for (int64_t i = 0; i < n; ++i) {
        if (sin_out || cos_out) {
            float sv, cv;
            portable_sincos(x[i], &sv, &cv);
            if (sin_out) sin_out[i] = sv;
            if (cos_out) cos_out[i] = cv;
        }
        if (acos_out) acos_out[i] = portable_acos(x[i]);
    }
}

}  // extern "C"

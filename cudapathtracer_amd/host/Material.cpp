// Material::sample / pdf / bsdf for host callers (common/Material.hpp:186-384 with the helpers at
// :11-108 and common/MathUtils.hpp:124-197), in the reference's float operations and order, with the
// C library's transcendental functions like its CPU build.  Written around one evaluation that
// returns pdf and bsdf together, as the device code does (csrc/hpt_device.h: mat_eval, glass_eval).
#include "Material.hpp"

#include <cmath>

namespace {

constexpr float kPi = 3.1415926535897932384626422832795028841971f;

inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline float lerp(float x, float y, float t) { return x * (1.0f - t) + y * t; }  // MathUtils.hpp:182-184
inline float clampf(float lo, float hi, float v) { return vmax(lo, vmin(hi, v)); }  // MathUtils.hpp:82-84
inline float pow5(float x) {
    const float x2 = x * x;
    return x2 * x2 * x;
}

// Math::local_to_world (MathUtils.hpp:124-136)
Vec3 local_to_world(Vec3 local_dir, Vec3 normal) {
    Vec3 t;
    if (std::fabs(normal.x) > std::fabs(normal.y)) {
        const float inv_len = 1.0f / std::sqrt(normal.x * normal.x + normal.z * normal.z);
        t = Vec3(normal.z * inv_len, 0.0f, -normal.x * inv_len);
    } else {
        const float inv_len = 1.0f / std::sqrt(normal.y * normal.y + normal.z * normal.z);
        t = Vec3(0.0f, normal.z * inv_len, -normal.y * inv_len);
    }
    const Vec3 b = cross(t, normal);
    return local_dir.x * b + local_dir.y * t + local_dir.z * normal;
}

Vec3 reflect(Vec3 ray_in_dir, Vec3 normal) { return ray_in_dir - dot(ray_in_dir, normal) * normal * 2.f; }  // Material.hpp:6-8

// Material.hpp:11-26
Vec3 refract(Vec3 ray_in_dir, Vec3 normal, float ior) {
    float cos_i = clampf(-1.0f, 1.0f, dot(ray_in_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    Vec3 correct_normal = normal;
    if (cos_i < 0.0f) {
        cos_i = -cos_i;
    } else {
        eta_i = ior;
        eta_t = 1.0f;
        correct_normal = -normal;
    }
    const float eta = eta_i / eta_t;
    const float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
    if (k < 0.0f) return Vec3(0.0f);
    return normalize(eta * ray_in_dir + (eta * cos_i - std::sqrt(k)) * correct_normal);
}

// Material.hpp:28-46
float fresnel(Vec3 observation_dir, Vec3 normal, float ior) {
    float cos_i = clampf(-1.0f, 1.0f, dot(observation_dir, normal));
    float eta_i = 1.0f, eta_t = ior;
    if (cos_i > 0.0f) {
        eta_i = ior;
        eta_t = 1.0f;
    }
    const float sin_t = eta_i / eta_t * std::sqrt(vmax(0.0f, 1.0f - cos_i * cos_i));
    if (sin_t >= 1.0f) return 1.0f;
    const float cos_t = std::sqrt(vmax(0.0f, 1.0f - sin_t * sin_t));
    cos_i = std::fabs(cos_i);
    const float rs = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    const float rp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rs * rs + rp * rp) / 2;
}

// Microfacet::distribution (:48-53)
float distribution(float ndm, float rsq) {
    float denominator = (ndm * ndm) * (rsq - 1.0f) + 1.0f;
    denominator = kPi * denominator * denominator;
    return rsq / denominator;
}
// Microfacet::geometry (:59-61)
float geometry(float a, float b, float roughness) { return 2.0f / lerp(std::fabs(2 * a * b), std::fabs(a + b), roughness); }

struct Eval {
    Vec3 bsdf = Vec3(0.0f);
    float pdf = 0.0f;
};

Eval evaluate(const Material& m, Vec3 src, Vec3 out, Vec3 normal) {
    Eval e;
    if (m.type == Material::Lambertian) {  // :259-262, :317-319
        if (dot(out, normal) > 0.0f) {
            e.pdf = 0.5f * 1.f / kPi;
            e.bsdf = (m._albedo * 1.f) / kPi;
        }
        return e;
    }
    const float rsq = m._roughness * m._roughness;
    const float nds = dot(normal, src), ndo = dot(normal, out);
    const float check = nds * ndo;
    if (m.type == Material::MetallicWorkflow) {  // :263-276, :320-347
        if (check <= 0.0f) return e;
        const Vec3 mm = normalize(src + out);
        const float ndm = dot(normal, mm), mdo = dot(mm, out);
        const float D = distribution(ndm, rsq);
        const float jacobian = mdo == 0.0f ? 0.0f : 1.0f / (4 * std::fabs(mdo));  // reflect_jacobian (:100-102)
        e.pdf = (D * std::fabs(ndm)) * jacobian;                                    // pdf_micro_surface (:76-80) * jacobian
        const float G = geometry(nds, ndo, m._roughness);
        const Vec3 f0(lerp(0.04f, m._albedo.x, m._metallic), lerp(0.04f, m._albedo.y, m._metallic), lerp(0.04f, m._albedo.z, m._metallic));
        const Vec3 F = f0 + (Vec3(1.0f) - f0) * pow5(1.0f - mdo);  // fresnel_schlick (:55-57)
        const Vec3 kd = (Vec3(1.0f) - F) * (1.0f - m._metallic);
        const Vec3 diffuse = (kd * m._albedo * 1.f) / kPi;
        const Vec3 specular = D * F * G / 4.0f;
        e.bsdf = diffuse + specular;
        return e;
    }
    // Glass (:277-309, :348-381), outward_micro_surface_normal (:82-98), refract_jacobian (:104-108)
    if (check == 0.0f) return e;
    const bool same_side = check > 0.0f, outward = ndo > 0.0f;
    Vec3 mm;
    if (same_side) {
        mm = normalize(src + out);
        if (!outward) mm = -mm;
    } else {
        mm = outward ? -normalize(out + src * m._ior) : -normalize(m._ior * out + src);
    }
    const float F = fresnel(-out, mm, m._ior);
    const float ndm = dot(normal, mm), mds = dot(mm, src), mdo = dot(mm, out);
    const float D = distribution(ndm, rsq);
    const float pm = D * std::fabs(ndm);
    const float G = geometry(mds, mdo, m._roughness);
    if (same_side) {
        const float jacobian = mdo == 0.0f ? 0.0f : 1.0f / (4 * std::fabs(mdo));
        e.pdf = pm * F * jacobian;
        e.bsdf = Vec3(D * F * G / 4.0f);
        return e;
    }
    const float ior_in = nds < 0.0f ? m._ior : 1.0f, ior_out = ndo < 0.0f ? m._ior : 1.0f;
    float den = ior_in * mds + ior_out * mdo;
    den *= den;
    const float jacobian = den == 0.0f ? 0.0f : (ior_out * ior_out * std::fabs(mdo)) / den;
    e.pdf = pm * (1.0f - F) * jacobian;
    e.bsdf = Vec3(jacobian * std::fabs(mds) * D * (1.0f - F) * G);
    return e;
}

}  // namespace

unsigned int RNG::utilhash(unsigned int a) {  // MathUtils.hpp:53-61
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}
RNG::RNG(int iter, int index, int dim) {
    const unsigned int h = utilhash((1u << 31) | ((unsigned int)dim << 22) | (unsigned int)iter) ^ utilhash((unsigned int)index);
    sampler = std::mt19937(h);
}

void Material::regularize() {
    float alpha_x = alpha();
    if (alpha_x < 0.3f) {
        alpha_x = vmin(vmax(2 * alpha_x, 0.1f), 0.3f);
        _roughness = alpha_x * alpha_x;
    }
}

Vec3 Material::sample(float r0, float r1, float r2, const Vec3& ray_out_dir, const Vec3& normal) const {
    if (type == Lambertian) {  // Math::sampleHemisphereCosine(normal, r0, r1) (MathUtils.hpp:152-162)
        const float r = std::sqrt(r0);
        const float theta = r1 * kPi * 2.0f;
        const float dx = std::cos(theta) * r, dy = std::sin(theta) * r;
        const float z = std::sqrt(1.f - (dx * dx + dy * dy));
        return local_to_world(Vec3(dx, dy, z), normal);
    }
    // Microfacet::sample_micro_surface (:66-74)
    const float rsq = _roughness * _roughness;
    const float theta = std::acos(std::sqrt((1.0f - r0) / ((rsq - 1.0f) * r0 + 1.0f)));
    const float phi = 2 * kPi * r1;
    const Vec3 micro = local_to_world(Vec3(std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)), normal);
    const Vec3 observation_dir = -ray_out_dir;
    if (type == Glass) {
        const float f = fresnel(observation_dir, micro, _ior);
        return r2 < f ? reflect(observation_dir, micro) : refract(observation_dir, micro, _ior);
    }
    return reflect(observation_dir, micro);
}

// The two Lambertian draws are arguments of one call in the reference (Material.hpp:189) and g++
// evaluates them right to left: the stream's first float is the second argument.  The microfacet
// types draw r0, r1 (and Glass a third) in separate statements (:67-68, :207).
Vec3 Material::sample(RNG& rng, const Vec3& ray_out_dir, const Vec3& normal) const {
    if (type == Lambertian) {
        const float second = rng.sample1D();
        const float first = rng.sample1D();
        return sample(first, second, 0.f, ray_out_dir, normal);
    }
    const float r0 = rng.sample1D();
    const float r1 = rng.sample1D();
    const float r2 = type == Glass ? rng.sample1D() : 0.f;
    return sample(r0, r1, r2, ray_out_dir, normal);
}

float Material::pdf(const Vec3& ray_source_dir, const Vec3& ray_out_dir, const Vec3& normal) const {
    return evaluate(*this, ray_source_dir, ray_out_dir, normal).pdf;
}
Vec3 Material::bsdf(const Vec3& ray_source_dir, const Vec3& ray_out_dir, const Vec3& normal) const {
    return evaluate(*this, ray_source_dir, ray_out_dir, normal).bsdf;
}

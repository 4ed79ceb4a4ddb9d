// Host-side material: the data members and predicates of the reference's Material
// (common/Material.hpp:130-162) and its sample / pdf / bsdf (:186-384) for host callers, with the C
// library's sin / cos / acos like the reference's CPU build.  The renderer does not call them: the
// bounce loop evaluates materials on the device (csrc/hpt_device.h).
#pragma once
#include <cstdint>
#include <random>

#include "Vector.hpp"

// The reference's host random stream (common/MathUtils.hpp:22-62): mt19937 seeded by a hash of
// (iteration, pixel index, dimension), uniform floats in [0, 1).
struct RNG {
    std::mt19937 sampler;
    RNG() {}
    RNG(int iter, int index, int dim = 0);
    float get_random_float() { return std::uniform_real_distribution<float>(0.f, 1.f)(sampler); }
    float sample1D() { return get_random_float(); }
    static unsigned int utilhash(unsigned int a);
};

struct Material {
    enum Type {
        Lambertian,
        MetallicWorkflow,
        Glass,
    };

    int type = Type::Lambertian;
    Vec3 _emission = Vec3(0.f);  // for light source
    Vec3 _albedo = Vec3(.9f);    // for diffuse & metallic
    float _metallic = 0.f;       // for metallic
    float _roughness = 1.f;      // for metallic & frosted glass
    float _ior = 0.f;            // for frosted glass

    bool emitting() const { return dot(_emission, _emission) > 0.0f; }
    Vec3 emission() const { return _emission; }
    float alpha() const { return std::sqrt(_roughness); }
    bool effectivelySmooth() const { return alpha() < 1e-3f; }
    bool hasSpecular() const { return (type == Type::MetallicWorkflow && effectivelySmooth()) || type == Type::Glass; }
    // Material::regularize (common/Material.hpp:164-171)
    void regularize();

    // Directions point away from the surface.  sample: the next direction for a path that arrived along
    // -ray_out_dir; the (r0, r1, r2) form takes the three uniform draws explicitly (r2: Glass only).
    Vec3 sample(RNG& rng, const Vec3& ray_out_dir, const Vec3& normal) const;
    Vec3 sample(float r0, float r1, float r2, const Vec3& ray_out_dir, const Vec3& normal) const;
    float pdf(const Vec3& ray_source_dir, const Vec3& ray_out_dir, const Vec3& normal) const;
    Vec3 bsdf(const Vec3& ray_source_dir, const Vec3& ray_out_dir, const Vec3& normal) const;
};

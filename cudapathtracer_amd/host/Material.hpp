// Host-side material description: the data members and predicates of the
// reference's Material (common/Material.hpp:130-162).  sample/pdf/bsdf are
// evaluated on the device only (csrc/hpt_device.h).
#pragma once
#include "Vector.hpp"

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
};

// Renderer: the base interface of the reference (common/Renderer.hpp:16-24):
// a sample count, the float RGB framebuffer (row-major, linear, already divided
// by spp) and a virtual Render(scene).
//
// The reference's base class also *is* its OpenMP CPU renderer
// (common/Renderer.cpp:16-51).  This library contains no CPU rendering path on
// purpose: the only renderer is HipRenderer, and it fails loudly when the GPU
// library is unavailable.  (A CPU restatement of the reference renderer exists
// under oracle/ strictly as test infrastructure.)
#pragma once
#include <vector>

#include "Scene.hpp"

class Renderer {
public:
    int spp = 32;
    std::vector<Vec3> framebuffer;
    virtual void Render(const Scene& scene) = 0;
    virtual ~Renderer() {}
};

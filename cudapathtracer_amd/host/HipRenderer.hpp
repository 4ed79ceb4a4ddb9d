// HipRenderer: drop-in for the reference's CudaRenderer
// (gpu_path_tracer/CudaRenderer.hpp:9-33) -- same public members, same call
// order (SetMode, num_threads, PrepareRender, Render, FinishRender, PathTrace)
// -- implemented on the C ABI of include/hpt.h.  Non-zero ABI status codes
// become std::runtime_error carrying hpt_last_error().
#pragma once
#include "../../include/hpt.h"
#include "Renderer.hpp"

enum HipRenderMode { SingleKernel = 1, Streamed = 2 };

class HipRenderer : public Renderer {
private:
    HipRenderMode mode = HipRenderMode::SingleKernel;
    hpt_ctx* ctx = nullptr;
    int num_pixels = 0;
    unsigned next_sample = 0;
    bool framebuffer_stale = false;  // PathTrace has added samples that `framebuffer` does not show yet

public:
    int num_threads = 256;
    int device = 0;  // the reference always renders on device 0
    // HPT_TRAVERSAL_REFERENCE (0, bit-exact, default) or HPT_TRAVERSAL_CULLED (1, opt-in); no
    // counterpart in the reference
    int traversal = 0;
    ~HipRenderer() override;
    void SetMode(HipRenderMode mode);
    void PrepareRender(const Scene& scene);
    void Render(const Scene& scene) override;
    void FinishRender(const Scene& scene);

    // one more sample per pixel (sample index = iter), accumulated on the device;
    // call it many times to reach the targeted spp.  Queues device work only.
    void PathTrace(const Scene& scene, int iter);
    // download accumulated / spp into `framebuffer` if PathTrace has run since the last download
    // (FinishRender does it too); no counterpart in the reference, whose PathTrace callers have
    // no way to see the picture before Render's own copy
    void RefreshFramebuffer();

    // ray counters and kernel time since PrepareRender
    hpt_stats Stats();
    double last_render_seconds = 0.0;
};

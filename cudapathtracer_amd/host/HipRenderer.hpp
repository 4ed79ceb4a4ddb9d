// HipRenderer: drop-in for the reference's CudaRenderer
// (gpu_path_tracer/CudaRenderer.hpp:9-33) -- same public members, same call
// order (SetMode, num_threads, PrepareRender, Render, FinishRender, PathTrace)
// -- implemented on the C ABI of include/hpt.h.  Non-zero ABI status codes
// become std::runtime_error carrying hpt_last_error().
#pragma once
#include "../../include/hpt.h"
#include <vector>

#include "Renderer.hpp"

enum HipRenderMode { SingleKernel = 1, Streamed = 2 };

class HipRenderer : public Renderer {
private:
    HipRenderMode mode = HipRenderMode::SingleKernel;
    hpt_ctx* ctx = nullptr;             // the context whose buffer holds the picture (devices[0], or `device`)
    std::vector<hpt_ctx*> extra_ctx;    // one more context per further entry of `devices`
    int num_pixels = 0;
    unsigned next_sample = 0;
    bool reduce_pending = false;     // PathTrace has put samples into the other GPUs' buffers
    bool framebuffer_stale = false;  // PathTrace has added samples that `framebuffer` does not show yet

public:
    int num_threads = 256;
    int device = 0;  // the reference always renders on device 0
    // More than one entry: samples-per-pixel sharded over these GPUs (contiguous sample ranges, one
    // host thread per GPU), the accumulation buffers summed onto devices[0] with one RCCL reduce
    // (hpt_accum_reduce); the picture is bit-identical to the single-GPU one.  An index may repeat
    // (several contexts on one GPU: a rehearsal of the flow on a one-GPU machine).  Empty = {device}.
    // No counterpart in the reference.
    std::vector<int> devices;
    // HPT_TRAVERSAL_REFERENCE (0, bit-exact, default) or HPT_TRAVERSAL_CULLED (1, opt-in); no
    // counterpart in the reference
    int traversal = 0;
    // HPT_INTEGRATOR_RUSSIAN_ROULETTE | HPT_INTEGRATOR_REGULARIZE: the steps of Scene::TracePath the
    // reference switches off at compile time (common/Scene.hpp:141-142); 0 = the reference's behaviour
    unsigned integrator = 0;
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
    double last_reduce_ms = 0.0;  // host time of the last cross-context reduce (several `devices` only)
};

#include "HipRenderer.hpp"

#include <chrono>
#include <iostream>
#include <stdexcept>
#include <string>

namespace {
void check(int rc, const char* what) {
    if (rc != HPT_OK) throw std::runtime_error(std::string(what) + ": " + hpt_last_error());
}
}  // namespace

HipRenderer::~HipRenderer() {
    if (ctx) hpt_context_destroy(ctx);
}

void HipRenderer::SetMode(HipRenderMode m) { mode = m; }

// scene upload + device buffers (CudaRenderer::PrepareRender, gpu_path_tracer/CudaRenderer.cu:19-35)
void HipRenderer::PrepareRender(const Scene& scene) {
    num_pixels = scene.width * scene.height;
    if (ctx) {
        hpt_context_destroy(ctx);
        ctx = nullptr;
    }
    check(hpt_context_create(device, &ctx), "hpt_context_create");
    check(hpt_set_traversal(ctx, traversal), "hpt_set_traversal");
    const hpt_scene_desc desc = scene.describe();
    check(hpt_scene_upload(ctx, &desc), "hpt_scene_upload");
    framebuffer.assign(num_pixels, Vec3(0.f));
    framebuffer_stale = false;
    next_sample = 0;
}

// CudaRenderer::FinishRender (gpu_path_tracer/CudaRenderer.cu:37-49)
void HipRenderer::FinishRender(const Scene&) {
    if (ctx) {
        RefreshFramebuffer();  // samples added by PathTrace since the last download
        check(hpt_context_destroy(ctx), "hpt_context_destroy");
    }
    ctx = nullptr;
}

// accumulators -> framebuffer (accumulated / spp), only when PathTrace has added samples since
// the last download
void HipRenderer::RefreshFramebuffer() {
    if (!ctx || !framebuffer_stale) return;
    check(hpt_resolve(ctx, nullptr, (uint32_t)(this->spp > 0 ? this->spp : 1), &framebuffer[0].x, nullptr), "hpt_resolve");
    framebuffer_stale = false;
}

// clear, trace spp samples per pixel, download (CudaRenderer::Render,
// gpu_path_tracer/CudaRenderer.cu:51-74, including its timing printout).  The
// spp samples go to the device as one request instead of spp launches.
void HipRenderer::Render(const Scene& scene) {
    if (!ctx) throw std::runtime_error("HipRenderer::Render: PrepareRender has not been called");
    auto start = std::chrono::system_clock::now();
    {
        check(hpt_accum_clear(ctx, nullptr, nullptr), "hpt_accum_clear");
        if (this->spp > 0) {
            check(hpt_render(ctx, (int)mode, num_threads, scene.maxDepth, 0u, (uint32_t)this->spp, nullptr, nullptr), "hpt_render");
            next_sample = (unsigned)this->spp;
            static_assert(sizeof(Vec3) == 3 * sizeof(float), "Vec3 must be three packed floats");
            check(hpt_resolve(ctx, nullptr, (uint32_t)this->spp, &framebuffer[0].x, nullptr), "hpt_resolve");
            framebuffer_stale = false;
        }
    }
    auto stop = std::chrono::system_clock::now();
    last_render_seconds = std::chrono::duration_cast<std::chrono::microseconds>(stop - start).count() * 1e-6;
    std::cout << "Render complete: \n";
    std::cout << "Time taken: " << std::chrono::duration_cast<std::chrono::hours>(stop - start).count() << " hours\n";
    std::cout << "          : " << std::chrono::duration_cast<std::chrono::minutes>(stop - start).count() << " minutes\n";
    std::cout << "          : " << std::chrono::duration_cast<std::chrono::milliseconds>(stop - start).count() / 1000.0f << " seconds\n";
}

// CudaRenderer::PathTrace (gpu_path_tracer/CudaRenderer.cu:76-94): one sample
// per pixel with sample index `iter`, added to the device accumulators.  Like
// the reference's, it only queues device work: nothing is copied to the host.
// `framebuffer` is brought up to date (accumulated / spp, what the reference's
// running `+= L / spp` holds after the same calls) by RefreshFramebuffer() or
// FinishRender().
void HipRenderer::PathTrace(const Scene& scene, int iter) {
    if (!ctx) throw std::runtime_error("HipRenderer::PathTrace: PrepareRender has not been called");
    check(hpt_render(ctx, (int)mode, num_threads, scene.maxDepth, (uint32_t)iter, 1u, nullptr, nullptr), "hpt_render");
    framebuffer_stale = true;
}

hpt_stats HipRenderer::Stats() {
    hpt_stats s{};
    if (ctx) check(hpt_stats_get(ctx, &s), "hpt_stats_get");
    return s;
}

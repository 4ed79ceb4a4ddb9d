#include "HipRenderer.hpp"

#include <chrono>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>

namespace {
void check(int rc, const char* what) {
    if (rc != HPT_OK) throw std::runtime_error(std::string(what) + ": " + hpt_last_error());
}
}  // namespace

HipRenderer::~HipRenderer() {
    for (hpt_ctx* c : extra_ctx) hpt_context_destroy(c);
    if (ctx) hpt_context_destroy(ctx);
}

void HipRenderer::SetMode(HipRenderMode m) { mode = m; }

// scene upload + device buffers (CudaRenderer::PrepareRender, gpu_path_tracer/CudaRenderer.cu:19-35)
void HipRenderer::PrepareRender(const Scene& scene) {
    num_pixels = scene.width * scene.height;
    for (hpt_ctx* c : extra_ctx) hpt_context_destroy(c);
    extra_ctx.clear();
    if (ctx) {
        hpt_context_destroy(ctx);
        ctx = nullptr;
    }
    const hpt_scene_desc desc = scene.describe();
    const std::vector<int> devs = devices.empty() ? std::vector<int>{device} : devices;
    for (size_t k = 0; k < devs.size(); ++k) {  // the scene is replicated on every GPU
        hpt_ctx* c = nullptr;
        check(hpt_context_create(devs[k], &c), "hpt_context_create");
        if (k == 0) ctx = c;
        else extra_ctx.push_back(c);
        check(hpt_set_traversal(c, traversal), "hpt_set_traversal");
        check(hpt_set_integrator(c, integrator), "hpt_set_integrator");
        check(hpt_set_bvh_build(c, scene.flat().gpu_build ? HPT_BVH_BUILD_GPU : HPT_BVH_BUILD_HOST), "hpt_set_bvh_build");
        check(hpt_scene_upload(c, &desc), "hpt_scene_upload");
    }
    if (!extra_ctx.empty()) {
        // the communicators of the reduce are made here, once, so that Render's clock holds a reduce and not a bootstrap
        std::vector<hpt_ctx*> all{ctx};
        all.insert(all.end(), extra_ctx.begin(), extra_ctx.end());
        check(hpt_accum_reduce_prepare(all.data(), (int)all.size()), "hpt_accum_reduce_prepare");
    }
    framebuffer.assign(num_pixels, Vec3(0.f));
    framebuffer_stale = false;
    reduce_pending = false;
    next_sample = 0;
}

// CudaRenderer::FinishRender (gpu_path_tracer/CudaRenderer.cu:37-49)
void HipRenderer::FinishRender(const Scene&) {
    if (ctx) {
        RefreshFramebuffer();  // samples added by PathTrace since the last download
        for (hpt_ctx* c : extra_ctx) check(hpt_context_destroy(c), "hpt_context_destroy");
        extra_ctx.clear();
        check(hpt_context_destroy(ctx), "hpt_context_destroy");
    }
    ctx = nullptr;
}

// accumulators -> framebuffer (accumulated / spp), only when PathTrace has added samples since
// the last download
void HipRenderer::RefreshFramebuffer() {
    if (!ctx || !framebuffer_stale) return;
    if (reduce_pending) {
        std::vector<hpt_ctx*> all{ctx};
        all.insert(all.end(), extra_ctx.begin(), extra_ctx.end());
        check(hpt_accum_reduce(all.data(), (int)all.size(), 0), "hpt_accum_reduce");
        for (hpt_ctx* c : extra_ctx) check(hpt_accum_clear(c, nullptr, nullptr), "hpt_accum_clear");  // their samples now live in ctx
        reduce_pending = false;
        check(hpt_accum_reduce_info(nullptr, &last_reduce_ms), "hpt_accum_reduce_info");
    }
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
        for (hpt_ctx* c : extra_ctx) check(hpt_accum_clear(c, nullptr, nullptr), "hpt_accum_clear");
        if (this->spp > 0) {
            if (extra_ctx.empty()) {
                check(hpt_render(ctx, (int)mode, num_threads, scene.maxDepth, 0u, (uint32_t)this->spp, nullptr, nullptr), "hpt_render");
            } else {
                // contiguous sample ranges, the first spp % G contexts take one more; one host thread per context
                std::vector<hpt_ctx*> all{ctx};
                all.insert(all.end(), extra_ctx.begin(), extra_ctx.end());
                const unsigned G = (unsigned)all.size(), base = (unsigned)this->spp / G, extra = (unsigned)this->spp % G;
                std::vector<std::string> errors(G);
                std::vector<std::thread> workers;
                for (unsigned g = 0; g < G; ++g) {
                    const unsigned begin = g * base + (g < extra ? g : extra), count = base + (g < extra ? 1u : 0u);
                    workers.emplace_back([&, g, begin, count]() {
                        if (count == 0) return;
                        if (hpt_render(all[g], (int)mode, num_threads, scene.maxDepth, begin, count, nullptr, nullptr) != HPT_OK)
                            errors[g] = hpt_last_error();  // thread-local message, read on the thread that failed
                    });
                }
                for (std::thread& t : workers) t.join();
                for (const std::string& e : errors)
                    if (!e.empty()) throw std::runtime_error("hpt_render: " + e);
                check(hpt_accum_reduce(all.data(), (int)G, 0), "hpt_accum_reduce");
                // the other contexts' samples now live in ctx: a later PathTrace + reduce must not count them again
                for (hpt_ctx* c : extra_ctx) check(hpt_accum_clear(c, nullptr, nullptr), "hpt_accum_clear");
                reduce_pending = false;
                check(hpt_accum_reduce_info(nullptr, &last_reduce_ms), "hpt_accum_reduce_info");
            }
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
    // several GPUs: sample `iter` goes to context iter mod G (any assignment gives the same sums)
    hpt_ctx* target = extra_ctx.empty() || iter % (int)(extra_ctx.size() + 1) == 0 ? ctx : extra_ctx[(size_t)(iter % (int)(extra_ctx.size() + 1)) - 1];
    check(hpt_render(target, (int)mode, num_threads, scene.maxDepth, (uint32_t)iter, 1u, nullptr, nullptr), "hpt_render");
    framebuffer_stale = true;
    reduce_pending = !extra_ctx.empty();
}

hpt_stats HipRenderer::Stats() {
    hpt_stats s{};
    if (ctx) check(hpt_stats_get(ctx, &s), "hpt_stats_get");
    for (hpt_ctx* c : extra_ctx) {  // counters add up over the GPUs; the kernel time is the longest GPU's
        hpt_stats t{};
        check(hpt_stats_get(c, &t), "hpt_stats_get");
        s.closest_queries += t.closest_queries;
        s.shadow_queries += t.shadow_queries;
        s.paths += t.paths;
        s.launches += t.launches;
        s.box_tests += t.box_tests;
        s.tri_tests += t.tri_tests;
        if (t.kernel_ms > s.kernel_ms) s.kernel_ms = t.kernel_ms;
    }
    return s;
}

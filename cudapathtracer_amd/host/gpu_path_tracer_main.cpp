// gpu_path_tracer <spp> <maxDepth> <threads> <mode>
//
// Same positional arguments, defaults, console output and output file as the
// reference CLI (gpu_path_tracer/main.cpp:10-58): spp 32, maxDepth 10, threads
// per workgroup 256, mode 1 = single kernel / 2 = streamed (wavefront with
// compaction).  Must be run from a directory that contains models/ and out/gpu/
// (or set HPT_MODELS_DIR), like the reference.
//
// Optional environment knobs that default to the reference's behaviour:
//   HPT_WIDTH / HPT_HEIGHT   resolution (the built-in scene is 512x512)
//   HPT_DEVICE               GPU index (default 0)
//   HPT_GPUS=N               shard the samples over GPUs 0..N-1 (one RCCL reduce sums the accumulators)
//   HPT_DEVICES=0,1,3        the same over the GPUs listed (an index may repeat: rehearsal on one GPU)
//   HPT_BVH=gpu | sah | reference  traversal records built on the GPU / surface-area tree / the reference's tree alone
//                                  (default "auto": surface-area tree for scenes of more than 64 primitives; same pictures)
//   HPT_INTEGRATOR=rr,regularize   Russian roulette / path regularisation as the reference writes them (off there)
//   HPT_SCENE                bunny | glass | heightfield:<n> | file:<scene file> instead of the reference's Cornell box
#include <cstdlib>
#include <iostream>
#include <string>

#include "HipRenderer.hpp"
#include "Scene.hpp"
#include "Utility.hpp"

int main(int argc, char** argv) {
    int spp = 32;
    int maxDepth = 10;
    int num_threads = 256;
    HipRenderMode mode = HipRenderMode::SingleKernel;
    if (argc > 1) spp = atoi(argv[1]);
    if (argc > 2) maxDepth = atoi(argv[2]);
    if (argc > 3) num_threads = atoi(argv[3]);
    if (argc > 4) mode = (HipRenderMode)atoi(argv[4]);

    try {
        // HPT_SCENE=bunny / glass / heightfield:<n> / file:<path> renders another scene than the reference's only one
        const char* which = getenv("HPT_SCENE");
        if (const char* b = getenv("HPT_BVH"))  // "gpu": traversal records built on the device at upload; "sah": surface-area splits on the host; anything else but "auto": the reference's tree
            Scene::default_bvh_build = std::string(b) == "gpu" ? Scene::BvhBuild::Gpu : std::string(b) == "sah" ? Scene::BvhBuild::Sah
                                       : std::string(b) == "auto" ? Scene::BvhBuild::Auto : Scene::BvhBuild::Reference;
        Scene scene = Scene::CreateNamedScene(which ? which : "cornell", maxDepth);
        if (const char* w = getenv("HPT_WIDTH")) scene.width = atoi(w);
        if (const char* h = getenv("HPT_HEIGHT")) scene.height = atoi(h);

        std::cout << "Resolution: " << scene.width << "x" << scene.height << "\n";
        std::cout << "SPP: " << spp << "\n";
        std::cout << "Trace Depth: " << maxDepth << "\n";
        std::cout << "HIP Mode: " << (mode == HipRenderMode::SingleKernel ? "SingleKernel" : "Streamed") << std::endl;
        std::cout << "HIP #threads: " << num_threads << std::endl;

        HipRenderer r;
        r.spp = spp;
        r.SetMode(mode);
        r.num_threads = num_threads;
        if (const char* d = getenv("HPT_DEVICE")) r.device = atoi(d);
        if (const char* g = getenv("HPT_GPUS"))
            for (int k = 0; k < atoi(g); ++k) r.devices.push_back(k);
        if (const char* l = getenv("HPT_DEVICES")) {
            r.devices.clear();
            for (const char* q = l; *q;) {
                r.devices.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        if (r.devices.size() > 1) std::cout << "HIP #GPUs: " << r.devices.size() << std::endl;
        if (const char* i = getenv("HPT_INTEGRATOR")) {  // "rr", "regularize" or "rr,regularize"
            const std::string v = i;
            if (v.find("rr") != std::string::npos) r.integrator |= HPT_INTEGRATOR_RUSSIAN_ROULETTE;
            if (v.find("regularize") != std::string::npos) r.integrator |= HPT_INTEGRATOR_REGULARIZE;
        }
        if (const char* t = getenv("HPT_TRAVERSAL")) r.traversal = std::string(t) == "culled" ? HPT_TRAVERSAL_CULLED : HPT_TRAVERSAL_REFERENCE;

        r.PrepareRender(scene);
        r.Render(scene);
        const hpt_stats st = r.Stats();
        r.FinishRender(scene);

        const double rays = (double)(st.closest_queries + st.shadow_queries);
        if (r.last_render_seconds > 0)
            std::cout << "Rays: " << rays << " (" << rays / r.last_render_seconds * 1e-6 << " Mrays/s, " << spp / r.last_render_seconds
                      << " spp/s)\n";
        if (r.devices.size() > 1) std::cout << "Reduce over " << r.devices.size() << " GPUs: " << r.last_reduce_ms << " ms (inside the time above)\n";

        Utility::SavePPM("out/gpu/" + scene.name + ".ppm", r.framebuffer, scene.width, scene.height);
        if (getenv("HPT_PFM")) Utility::SavePFM("out/gpu/" + scene.name + ".pfm", r.framebuffer, scene.width, scene.height);
    } catch (const std::exception& e) {
        std::cerr << "gpu_path_tracer: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

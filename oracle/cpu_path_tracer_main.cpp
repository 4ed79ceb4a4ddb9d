// TEST INFRASTRUCTURE ONLY -- the CPU baseline as a command-line program.
//
// cpu_path_tracer <spp> <maxDepth> <threads>: the reference's CPU front end
// (cpu_path_tracer/main.cpp:16-56: defaults 32 / 8 / 8, the built-in Cornell box,
// Renderer::Render, out/cpu/<scene name>.ppm, the same lines on stdout) over the CPU
// restatement in oracle.cpp, with the reference's random stream (mt19937 + utilhash) and libm,
// i.e. the configuration in which the restatement reproduces the compiled reference's image bit
// for bit.  It exists so that the reference's CPU-runnable benchmark configuration
// (`cpu_path_tracer 16 5 8`, 512 x 512) can be timed and its picture compared on a machine that
// has no /root/reference.  Nothing in cudapathtracer_amd/ builds, links or runs it.
//
// Environment (not in the reference): HPT_MODELS_DIR (default "models"), HPT_WIDTH / HPT_HEIGHT
// (default 512 x 512, common/Scene.cpp:29), HPT_OUT_DIR (default "out/cpu"), HPT_PFM=1 (also write the linear
// radiance as <scene name>.pfm).
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include "oracle.h"

static bool save_ppm(const std::string& path, const std::vector<float>& rgb, int width, int height) {
    // Utility::SavePPM (common/Utility.cpp:4-22)
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) {
        std::cerr << "Failed to open file '" << path << "'\n";
        return false;
    }
    std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
    for (int i = 0; i < width * height; ++i) {
        unsigned char color[3];
        for (int c = 0; c < 3; ++c) color[c] = (unsigned char)orc_tonemap(rgb[3 * (size_t)i + c]);
        std::fwrite(color, 1, 3, fp);
    }
    std::fclose(fp);
    return true;
}

int main(int argc, char** argv) {
    int spp = 32;
    int maxDepth = 8;
    int omp_num_threads = 8;
    if (argc > 1) spp = std::atoi(argv[1]);
    if (argc > 2) maxDepth = std::atoi(argv[2]);
    if (argc > 3) omp_num_threads = std::atoi(argv[3]);
    if (spp < 1 || maxDepth < 0 || omp_num_threads < 1) {
        std::cerr << "usage: cpu_path_tracer <spp >= 1> <maxDepth >= 0> <threads >= 1>\n";
        return 2;
    }
    const char* models = std::getenv("HPT_MODELS_DIR");
    const char* w = std::getenv("HPT_WIDTH");
    const char* h = std::getenv("HPT_HEIGHT");
    const char* out_dir = std::getenv("HPT_OUT_DIR");
    const int width = w ? std::atoi(w) : 512, height = h ? std::atoi(h) : 512;
    orc_scene* scene = orc_scene_create_cornell(models ? models : "models", width, height);
    if (!scene) {
        std::cerr << "cannot read the Cornell box OBJ files under '" << (models ? models : "models") << "/cornellbox/'\n";
        return 1;
    }
#if defined(_OPENMP)
    std::cout << "OpenMP: Enabled, #threads=" << omp_num_threads << "\n";
#else
    std::cout << "OpenMP: Disabled\n";
#endif
    std::vector<float> framebuffer((size_t)3 * width * height, 0.f);
    orc_stats st{};
    auto start = std::chrono::system_clock::now();
    orc_render(scene, ORC_RNG_MT, ORC_TRIG_LIBM, 0u, maxDepth, 0, spp, spp, omp_num_threads, framebuffer.data(), nullptr, &st);
    auto stop = std::chrono::system_clock::now();
    orc_scene_destroy(scene);

    const std::string dir = out_dir ? out_dir : "out/cpu";
    ::mkdir("out", 0777);
    ::mkdir(dir.c_str(), 0777);
    if (!save_ppm(dir + "/Cornell Box.ppm", framebuffer, width, height)) return 1;
    if (std::getenv("HPT_PFM")) {  // linear radiance, rows bottom to top, little-endian floats (what tools/image_diff.py reads)
        FILE* fp = std::fopen((dir + "/Cornell Box.pfm").c_str(), "wb");
        if (!fp) return 1;
        std::fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);
        for (int y = height - 1; y >= 0; --y) std::fwrite(&framebuffer[(size_t)3 * y * width], sizeof(float), (size_t)3 * width, fp);
        std::fclose(fp);
    }

    const double seconds = std::chrono::duration_cast<std::chrono::milliseconds>(stop - start).count() / 1000.0;
    std::cout << "Render complete: \n";
    std::cout << "Time taken: " << std::chrono::duration_cast<std::chrono::hours>(stop - start).count() << " hours\n";
    std::cout << "          : " << std::chrono::duration_cast<std::chrono::minutes>(stop - start).count() << " minutes\n";
    std::cout << "          : " << (float)seconds << " seconds\n";
    // not in the reference: the benchmark's own units
    const double rays = (double)st.closest_queries + (double)st.shadow_queries;
    std::cout << "Rays: " << (unsigned long long)rays << " (" << st.closest_queries << " closest + " << st.shadow_queries << " shadow), "
              << (seconds > 0 ? rays / seconds / 1e6 : 0.0) << " Mrays/s, " << (seconds > 0 ? spp / seconds : 0.0) << " spp/s\n";
    return 0;
}

#include "Utility.hpp"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>

namespace {
unsigned char tone(float v) {
    if (!(v == v)) return 0;  // NaN: the reference's conversion of a NaN product yields 0 on x86-64
    const float c = vmax(0.f, vmin(1.f, v));
    return (unsigned char)(255 * std::pow(c, 0.6f));
}
}  // namespace

void Utility::SavePPM(const std::string& path, const std::vector<Vec3>& frameBuffer, int width, int height) {
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("Failed to open file '" + path + "'. Error: " + std::strerror(errno));
    std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const Vec3& p = frameBuffer[(size_t)y * width + x];
            row[3 * x] = tone(p.x);
            row[3 * x + 1] = tone(p.y);
            row[3 * x + 2] = tone(p.z);
        }
        std::fwrite(row.data(), 1, row.size(), fp);
    }
    std::fclose(fp);
}

void Utility::SavePFM(const std::string& path, const std::vector<Vec3>& frameBuffer, int width, int height) {
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("Failed to open file '" + path + "'. Error: " + std::strerror(errno));
    std::fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);  // negative scale = little-endian floats
    for (int y = height - 1; y >= 0; --y) std::fwrite(&frameBuffer[(size_t)y * width], sizeof(Vec3), (size_t)width, fp);
    std::fclose(fp);
}

void Utility::UpdateProgress(float progress) {
    const int barWidth = 70;
    const int pos = (int)(barWidth * progress);
    std::cout << "[";
    for (int i = 0; i < barWidth; ++i) std::cout << (i < pos ? "=" : (i == pos ? ">" : " "));
    std::cout << "] " << int(progress * 100.0) << " %\r";
    std::cout.flush();
}

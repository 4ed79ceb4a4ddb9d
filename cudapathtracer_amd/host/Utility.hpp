// Image output and console progress (the reference's Utility, common/Utility.hpp).
#pragma once
#include <string>
#include <vector>

#include "Vector.hpp"

namespace Utility {
// Binary PPM (P6); each channel is 255 * clamp(v, 0, 1)^0.6 truncated to a byte,
// the tone curve of common/Utility.cpp:16-18.  Throws if the file cannot be opened.
void SavePPM(const std::string& path, const std::vector<Vec3>& frameBuffer, int width, int height);
// Portable float map (PF, little-endian, rows bottom to top): the linear radiance as it is, for
// image-diff tooling.  Not in the reference (its only output is the tone-mapped PPM).
void SavePFM(const std::string& path, const std::vector<Vec3>& frameBuffer, int width, int height);
void UpdateProgress(float progress);
}  // namespace Utility

// Checks the exact-arithmetic shortcuts of cudapathtracer_amd/csrc/hpt_device.h AS COMPILED FROM THAT HEADER
// (nothing is pasted here) on the GPU:
//   rcp_rn, sqrt_rn, two_over          every float (all 2^32 bit patterns) against 1.0f / x, sqrtf(x), 2.0f / x
//   reciprocal_of_float                every float with |x| >= 1e-5 (the callers' precondition) against 1. / (double)x
//   tri_classify + nearest_update + tri_occludes_fast
//                                      2^30 generated (det, un, vn, tn) tuples -- uniform bit patterns, and tuples
//                                      placed within a few ulps of every decision boundary -- against the
//                                      reference's sequence of double operations (Triangle::getIntersection,
//                                      common/Triangle.hpp:172-188; the t comparisons of BVH.hpp:121-130)
// Build (from the repository root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -o tools/micro/exact_math_check tools/micro/exact_math_check.hip
// Prints one line per check and returns non-zero if anything differs.
#include "../../cudapathtracer_amd/csrc/hpt_device.h"

#include <cstdio>

using namespace hpt;

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // 32-bit finaliser (generates test inputs only)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int kWhat>
__global__ void k_unary(unsigned long long* bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (kWhat == 0) mine += __float_as_uint(rcp_rn(x)) != __float_as_uint(1.0f / x) && !(x != x);
        if (kWhat == 1) mine += __float_as_uint(sqrt_rn(x)) != __float_as_uint(sqrtf(x)) && !(x != x) && !(x < 0);
        if (kWhat == 2) mine += __float_as_uint(two_over(x)) != __float_as_uint(2.0f / x) && !(x != x);
        if (kWhat == 3) {
            if (!(fabsf(x) >= 1e-5f)) continue;
            mine += __double_as_longlong(reciprocal_of_float(x)) != __double_as_longlong(1. / (double)x);
        }
    }
    if (mine) atomicAdd(bad, mine);
}

// the reference's verdict and t from the four floats
__device__ __forceinline__ bool exact_verdict(const TriNum& n, double& t) {
    const double det = (double)n.det;
    if (fabs(det) < (double)kEpsilon5) return false;
    const double det_inv = 1. / det;
    const double u = (double)n.un * det_inv;
    if (u < 0 || u > 1) return false;
    const double v = (double)n.vn * det_inv;
    if (v < 0 || u + v > 1) return false;
    t = (double)n.tn * det_inv;
    return !(t < 0);
}

__device__ __forceinline__ float nudge(float x, int ulps) { return __uint_as_float(__float_as_uint(x) + (uint32_t)ulps); }

__device__ TriNum make_tuple(uint32_t i, uint32_t salt) {
    const uint32_t a = mix(i * 4u + salt), b = mix(a + 1u), c = mix(b + 2u), d = mix(c + 3u);
    TriNum n;
    const uint32_t kind = a & 7u;
    if (kind == 0) {  // uniform bit patterns (NaNs, infinities, denormals, negatives included)
        n.det = __uint_as_float(a); n.un = __uint_as_float(b); n.vn = __uint_as_float(c); n.tn = __uint_as_float(d);
        return n;
    }
    // a plausible hit, then pushed onto a boundary
    n.det = ldexpf(1.0f + (float)(b >> 9) * (1.0f / 8388608.0f), (int)(c % 60u) - 20);
    const float u = (float)(c >> 8) * (1.0f / 16777216.0f), v = (float)(d >> 8) * (1.0f / 16777216.0f) * (1.0f - u);
    n.un = u * n.det; n.vn = v * n.det;
    n.tn = ldexpf(1.0f + (float)(a >> 9) * (1.0f / 8388608.0f), (int)(d % 50u) - 20) * n.det;
    const int j = (int)(b & 15u) - 8;
    if (kind == 1) { n.vn = nudge(n.det - n.un, j); }                          // u + v at 1
    if (kind == 2) { n.un = nudge(n.det, j); n.vn = (d & 1u) ? 0.f : -0.f; }   // u at 1
    if (kind == 3) { n.un = (d & 1u) ? 0.f : -0.f; if (d & 2u) n.un = nudge(0.f, (int)(b & 7u)); }  // u at 0 (zeros, denormals)
    if (kind == 4) { n.tn = (d & 1u) ? 0.f : -0.f; if (d & 2u) n.tn = -nudge(0.f, (int)(b & 7u)); }  // t at 0
    if (kind == 5) { n.det = nudge(kEpsilon5, j); n.un *= 1e-5f / n.det; }    // det at the cut-off
    if (kind == 6) { n.det = -n.det; if (d & 1u) { n.un = -n.un; n.vn = -n.vn; n.tn = -n.tn; } }   // negative det
    if (kind == 7) { n.det = nudge(1.0e30f, j); }                              // det at the fast rule's upper end
    return n;
}

__global__ void k_classify(unsigned long long* bad, uint32_t salt) {
    // bad[0] verdict, [1] accuracy of ta, [2] tri_exact_t, [3] closest update (reference order), [4] closest update (tie rule), [5] any-hit
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t r = 0; r < 256u; ++r) {
        const TriNum n = make_tuple(i * 256u + r, salt);
        float ta;
        double t = 0;
        const bool got = tri_classify(n, ta), want = exact_verdict(n, t);
        if (got != want) { ++mine[0]; continue; }
        if (!want) continue;
        if (ta == ta) {  // the float quotient's stated accuracy
            const double err = fabs((double)ta - t);
            const bool overflow = ta == __builtin_inff() && t > 3.4028e38;  // beyond the float range: reads as "farther than anything"
            if (!(err <= 3e-7 * fabs(t) + 1.2e-38) && !overflow) {
                if (!mine[1]) printf("accuracy: det %a un %a vn %a tn %a ta %a t %a\n", n.det, n.un, n.vn, n.tn, ta, t);
                ++mine[1];
            }
        }
        if (__double_as_longlong(tri_exact_t(n.tn, n.det)) != __double_as_longlong(t)) ++mine[2];
        // closest-hit update against a second accepted tuple whose t is this one's, a few ulps away, or unrelated
        TriNum m = make_tuple(i * 256u + r, salt + 77u);
        if (r & 1u) { m.det = n.det; m.tn = nudge(n.tn, (int)(r >> 1 & 7u) - 4); m.un = n.un; m.vn = n.vn; }
        float tb;
        double t2 = 0;
        if (tri_classify(m, tb) && exact_verdict(m, t2)) {
            Nearest nb = nearest_none();
            nearest_update<false>(nb, 5, n, ta);
            nearest_update<false>(nb, 9, m, tb);
            double best_t = (double)kInfinity;
            int best = -1;
            if (t < best_t) { best = 5; best_t = t; }
            if (t2 < best_t) { best = 9; best_t = t2; }
            if (nb.tri != best || __double_as_longlong(nearest_t(nb)) != __double_as_longlong(best_t)) ++mine[3];
            Nearest nt = nearest_none();
            nearest_update<true>(nt, 9, n, ta);
            nearest_update<true>(nt, 5, m, tb);  // the later candidate has the lower index: wins ties
            best_t = (double)kInfinity; best = -1;
            if (t < best_t) { best = 9; best_t = t; }
            if (t2 < best_t || (t2 == best_t && 5 < best)) { best = 5; best_t = t2; }
            if (nt.tri != best || __double_as_longlong(nearest_t(nt)) != __double_as_longlong(best_t)) ++mine[4];
        }
        // any-hit rule around this t
        const double tmaxs[4] = {t + 1e-4, t + 1.0001e-4, (double)(float)t, t * 1.5 + 1.0};
        for (double tmax : tmaxs) {
            const double limit = (double)(float)tmax - (double)kEpsilon4;
            const int occ = tri_occludes_fast(ta, (float)limit);
            const bool want_occ = t < limit;
            if (occ >= 0 && (occ > 0) != want_occ) ++mine[5];
        }
    }
    for (int c = 0; c < 6; ++c)
        if (mine[c]) atomicAdd(bad + c, mine[c]);
}

static unsigned long long* g_bad;
static unsigned long long fetch(unsigned long long* all = nullptr) {
    unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
    (void)hipMemcpy(h, g_bad, 48, hipMemcpyDeviceToHost);
    (void)hipMemset(g_bad, 0, 48);
    if (all)
        for (int c = 0; c < 6; ++c) all[c] = h[c];
    return h[0] + h[1] + h[2] + h[3] + h[4] + h[5];
}

int main() {
    (void)hipMalloc(&g_bad, 48);
    (void)hipMemset(g_bad, 0, 48);
    unsigned long long total = 0, h;
    k_unary<0><<<4096, 256>>>(g_bad); h = fetch(); total += h; printf("rcp_rn(x) vs 1.0f / x, every float:              %llu differ\n", h);
    k_unary<1><<<4096, 256>>>(g_bad); h = fetch(); total += h; printf("sqrt_rn(x) vs sqrtf(x), every float >= 0:        %llu differ\n", h);
    k_unary<2><<<4096, 256>>>(g_bad); h = fetch(); total += h; printf("two_over(x) vs 2.0f / x, every float:            %llu differ\n", h);
    k_unary<3><<<4096, 256>>>(g_bad); h = fetch(); total += h; printf("reciprocal_of_float(x) vs 1. / (double)x, |x| >= 1e-5: %llu differ\n", h);
    for (uint32_t salt = 0; salt < 4u; ++salt) k_classify<<<4096, 256>>>(g_bad, salt * 1000003u);
    unsigned long long c[6];
    h = fetch(c); total += h;
    printf("tri_classify / nearest_update / tri_occludes_fast vs the double sequence, 2^30 tuples: %llu differ\n"
           "  (verdict %llu, accuracy of the float quotient %llu, exact t %llu, closest update %llu, with tie rule %llu, any-hit %llu)\n",
           h, c[0], c[1], c[2], c[3], c[4], c[5]);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    return total != 0;
}

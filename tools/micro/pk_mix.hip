// Throughput of the slab-test instruction mix with scalar vs packed f32 subtract/multiply,
// 8 waves per SIMD, no memory access.  Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int kMode>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float ox = seed + threadIdx.x, oy = seed * 2, oz = seed * 3, ix = 1.1f, iy = 1.2f, iz = 1.3f;
    float lx = 1, hx = 2, ly = 3, hy = 4, lz = 5, hz = 6;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        float tn, tf;
        if (kMode == 0) {
            float ax = (lx - ox) * ix, bx = (hx - ox) * ix, ay = (ly - oy) * iy, by = (hy - oy) * iy, az = (lz - oz) * iz, bz = (hz - oz) * iz;
            asm volatile("" : "+v"(ax), "+v"(bx), "+v"(ay), "+v"(by), "+v"(az), "+v"(bz));
            tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
            tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        } else {
            v2f tx = ((v2f){lx, hx} - (v2f){ox, ox}) * (v2f){ix, ix};
            v2f ty = ((v2f){ly, hy} - (v2f){oy, oy}) * (v2f){iy, iy};
            v2f tz = ((v2f){lz, hz} - (v2f){oz, oz}) * (v2f){iz, iz};
            asm volatile("" : "+v"(tx), "+v"(ty), "+v"(tz));
            tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
            tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
        }
        const bool hit = tn <= tf;
        acc += hit ? 1.0f : 0.0f;
        // new "node": keeps the loop body from being hoisted
        lx = hit ? hx : ly; ly = hit ? hy : lz; lz = hit ? hz : lx;
        asm volatile("" : "+v"(lx), "+v"(hx), "+v"(ly), "+v"(hy), "+v"(lz), "+v"(hz));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int kMode>
void run(const char* name, float* out) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<kMode><<<blocks, 256>>>(out, 100, 1.f);
    (void)hipEventRecord(e0);
    k<kMode><<<blocks, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("%-10s %8.3f ms  %.1f cycles (2.4 GHz) per loop body per SIMD\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("scalar", out); run<1>("packed", out); run<0>("scalar", out); run<1>("packed", out);
    return 0;
}

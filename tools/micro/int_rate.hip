// Issue rate of the integer multiplies a Philox round is made of, and of the IEEE division /
// square-root sequences of the shading code, on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o int_rate int_rate.hip ; run: ./int_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kMode>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    unsigned a[8];
    unsigned long long w[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = (unsigned)(seed * 977) + i * 7919u + threadIdx.x; w[i] = a[i]; f[i] = seed + i + threadIdx.x; }
    const unsigned m = 0xD2511F53u;
    const float g = 1.0000001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kMode == 0) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 1) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(w[i]) : "v"(m), "v"(a[i]) : "vcc");
            if (kMode == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 4) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 5) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 6) f[i] = f[i] / g;            // correctly rounded (build with -fhip-fp32-correctly-rounded-divide-sqrt)
            if (kMode == 7) f[i] = __builtin_sqrtf(f[i]);
            if (kMode == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (kMode == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += (float)a[i] + (float)w[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int kMode>
void run(const char* name, float* out) {
    const int iters = 5000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<kMode><<<blocks, 256>>>(out, 100, 1.f);
    hipEventRecord(e0);
    k<kMode><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
    printf("%-22s %8.3f ms  %.2f cycles per wave-operation per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_mul_lo_u32", out); run<1>("v_mul_hi_u32", out); run<2>("v_mad_u64_u32", out); run<3>("v_xor_b32", out);
    run<4>("v_mul_u32_u24", out); run<5>("v_mul_hi_u32_u24", out); run<6>("IEEE f32 division", out); run<7>("IEEE f32 sqrt", out);
    run<8>("v_rcp_f32", out); run<9>("v_sqrt_f32", out);
    return 0;
}

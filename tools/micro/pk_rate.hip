// Issue rate of packed-f32 VALU instructions against their scalar-f32 forms on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int kMode>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){seed + i + threadIdx.x, seed - i};
    v2f m = {1.0000001f, 0.9999999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kMode == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 3) asm volatile("v_max3_f32 %0, %1, %0, %0" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 4) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 5) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 6) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (kMode == 7) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 8) asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[10:11]" : "+v"(a[i].x) : "v"(m.x) : "s10", "s11");
            if (kMode == 9) asm volatile("v_cmp_le_f32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i].x) : "v"(m.x) : "vcc");
            if (kMode == 10) asm volatile("v_cmp_le_f32 vcc, %1, %0" : "+v"(a[i].x) : "v"(m.x) : "vcc");
            if (kMode == 11) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 12) asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 13) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 14) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
            if (kMode == 15) asm volatile("v_cmp_le_f32 vcc, %1, %0\n v_mov_b32 %0, vcc_lo" : "+v"(a[i].x) : "v"(m.x) : "vcc");
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int kMode>
void run(const char* name, float* out) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<kMode><<<blocks, 256>>>(out, 100, 1.f);
    hipEventRecord(e0);
    k<kMode><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves * iters * 8 / 1024 SIMDs
    double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
    printf("%-14s %8.3f ms  %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.2f cyc)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_mul_f32", out); run<1>("v_pk_mul_f32", out); run<2>("v_pk_add_f32", out); run<3>("v_max3_f32", out);
    run<4>("v_pk_fma_f32", out); run<5>("v_fma_f64", out); run<6>("v_mul_f64", out); run<7>("v_cndmask_b32", out);
    run<8>("cndmask sgpr", out); run<9>("cmp+nop+cndmask", out); run<10>("v_cmp_le_f32", out); run<11>("v_min_f32", out);
    run<12>("v_lshl_add_u32", out); run<13>("v_or_b32", out); run<14>("v_sub_f32", out); run<15>("cmp+mov vcc_lo", out);
    return 0;
}

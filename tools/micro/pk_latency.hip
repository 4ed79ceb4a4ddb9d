// Dependent-chain latency of f32 VALU instructions (one wave, in-kernel cycle counter).
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_latency pk_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int kMode>
__global__ void k(unsigned long long* out, float seed) {
    v2f a = {seed + threadIdx.x, seed}, m = {1.0000001f, 0.9999999f};
    float lo = a.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (kMode == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(lo) : "v"(m.x));
            if (kMode == 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a) : "v"(m));
            if (kMode == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a) : "v"(m));
            if (kMode == 3) asm volatile("v_max3_f32 %0, %1, %0, %0" : "+v"(lo) : "v"(m.x));
            if (kMode == 4) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(a) : "v"(m));
            if (kMode == 5) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(lo) : "v"(m.x));
            if (kMode == 6) asm volatile("v_pk_mul_f32 %0, %2, %0\n v_min_f32 %1, %1, %1" : "+v"(a), "+v"(lo) : "v"(m));
            if (kMode == 7) asm volatile("v_mul_f32 %0, %2, %0\n v_min_f32 %1, %1, %1" : "+v"(a.x), "+v"(lo) : "v"(m.x));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a.x + a.y + lo == 12345.f) out[1] = 1;
}
template <int kMode>
void run(const char* name, unsigned long long* out) {
    k<kMode><<<1, 64>>>(out, 1.f);
    hipDeviceSynchronize();
    k<kMode><<<1, 64>>>(out, 1.f);
    hipDeviceSynchronize();
    printf("%-28s %.2f cycles per step\n", name, (double)out[0] / 16000.0);
}
int main() {
    unsigned long long* out; (void)hipHostMalloc(&out, 64);
    run<0>("v_mul_f32 chain", out); run<1>("v_pk_mul_f32 chain", out); run<2>("v_pk_add_f32 chain", out);
    run<3>("v_max3_f32 chain", out); run<4>("v_fma_f64 chain", out); run<5>("v_cndmask chain", out);
    run<6>("pk_mul + indep v_min", out); run<7>("v_mul + indep v_min", out);
    return 0;
}

// Exhaustive check of reciprocal_of_float (cudapathtracer_amd/csrc/hpt_device.h) against the IEEE
// double division 1. / (double)x for EVERY float x (all 2^32 bit patterns).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../cudapathtracer_amd/csrc -o rcp_check rcp_check.hip
// (Kept as the stand-alone derivation of the sequence; the regression check that compiles the function from
// hpt_device.h itself is tools/micro/exact_math_check.hip.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ double reciprocal_of_float(float x) {
    const double d = (double)x;
    if (!(fabsf(x) < 1.0e30f)) return 1. / d;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}
__global__ void k(unsigned long long* bad, unsigned* first_bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        if (!(fabsf(x) >= 1e-5f)) continue;  // the callers' precondition (also drops NaN)
        const double a = reciprocal_of_float(x), b = 1. / (double)x;
        if (__double_as_longlong(a) != __double_as_longlong(b)) {
            ++mine;
            atomicMin(first_bad, (unsigned)i);
        }
    }
    if (mine) atomicAdd(bad, mine);
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
    k<<<4096, 256>>>(bad, first);
    unsigned long long h = 0; unsigned f = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("floats with |x| >= 1e-5 (incl. infinities): %llu results differ from 1./(double)x", h);
    if (h) printf(" (first bit pattern 0x%08x)", f);
    printf("\n");
    return h != 0;
}

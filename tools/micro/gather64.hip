// How fast can every lane of a wave fetch its OWN 64-byte record (a dependent chain of random
// records, like a BVH walk)?
//   mode 0: four global_load_dwordx4 per lane (each instruction touches 64 different lines)
//   mode 1: quad-cooperative: in instruction k the four lanes of a quad load the four 16-byte
//           pieces of quad-lane k's record (16 different lines per instruction), straight into LDS
//           (global_load_lds_dwordx4: wave-uniform base + lane * 16); each lane then reads its
//           record back with four ds_read_b128
//   mode 2: as 1, through registers (global_load_dwordx4, ds_write_b128, ds_read_b128)
// Build: hipcc --offload-arch=gfx950 -O3 -o gather64 gather64.hip ; run: ./gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float raw_float4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned quad_bcast(unsigned v, int k) {
    switch (k) {
    case 0: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
    case 1: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
    case 2: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
    default: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);  // [3,3,3,3]
    }
}

template <int kMode>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ rec, unsigned n_rec, int iters, unsigned* out) {
    __shared__ float4 stage[4 * 256];  // 4 KiB per wave: 4 pieces x 64 lanes x 16 B
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4* ws = stage + wave * 256;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        float4 q0, q1, q2, q3;
        if (kMode == 0) {
            q0 = rec[4 * (size_t)idx]; q1 = rec[4 * (size_t)idx + 1]; q2 = rec[4 * (size_t)idx + 2]; q3 = rec[4 * (size_t)idx + 3];
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned r = quad_bcast(idx, kk);
                const float4* src = rec + 4 * (size_t)r + (lane & 3u);
                if (kMode == 1) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(ws + kk * 64), 16, 0, 0);
                } else {
                    ws[kk * 64 + lane] = *src;
                }
            }
            if (kMode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = ws + (lane & 3u) * 64 + (lane & ~3u);
            q0 = mine[0]; q1 = mine[1]; q2 = mine[2]; q3 = mine[3];
            asm volatile("" ::: "memory");
        }
        // next record: depends on all four pieces
        const unsigned h = __float_as_uint(q0.x) ^ __float_as_uint(q1.y) ^ __float_as_uint(q2.z) ^ __float_as_uint(q3.w);
        acc += h;
        idx = (h * 2654435761u + it) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int kMode>
void run(const char* name, const float4* rec, unsigned n_rec, unsigned* out) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<kMode><<<blocks, 256>>>(rec, n_rec, 50, out);
    hipEventRecord(e0);
    k<kMode><<<blocks, 256>>>(rec, n_rec, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fetches = (double)blocks * 256 * iters;
    unsigned h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s %9u records  %8.3f ms  %7.2f G records/s  %6.1f CU-cycles per wave-fetch  (check %08x)\n", name, n_rec, ms,
           fetches / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / (fetches / 64), h[0] ^ h[1] ^ h[2] ^ h[3]);
}

int main() {
    for (unsigned n_rec : {10000u, 2000000u}) {
        std::vector<unsigned> host((size_t)n_rec * 16);
        unsigned s = 12345u;
        for (auto& v : host) { s = s * 1664525u + 1013904223u; v = s; }
        float4* rec; unsigned* out;
        hipMalloc(&rec, host.size() * 4); hipMalloc(&out, 256 * 4 * 256 * 4);
        hipMemcpy(rec, host.data(), host.size() * 4, hipMemcpyHostToDevice);
        run<0>("4 x global_load_dwordx4 per lane", rec, n_rec, out);
        run<1>("quad-cooperative, LDS-DMA + ds_read_b128", rec, n_rec, out);
        run<2>("quad-cooperative, registers + ds_write/read", rec, n_rec, out);
        hipFree(rec); hipFree(out);
    }
    return 0;
}

// gfx950 kernels and the C ABI (include/hpt.h) of the path-tracing core.
//
// Kernels
//   k_intersect_batch  closest/any-hit queries on a caller-supplied ray batch
//   k_render_mega      mode 1: persistent waves draw (pixel tile, sample chunk) units from a
//                      global counter; a lane whose path ends starts the next (pixel, sample)
//                      of the wave's unit at once, so the wave stays full although path
//                      lengths differ from 1 to max_depth bounces
//   k_wf_generate / k_wf_bounces
//                      mode 2: wavefront; path state lives in 64-byte SoA records in one
//                      queue segment per workgroup, survivors are compacted with a wave
//                      ballot + prefix count and one LDS atomic per wave, the live counts
//                      stay on the device (no host round trip per bounce)
//   k_resolve          fixed-point accumulators -> float RGB / spp
//
// Small scenes are staged into LDS once per workgroup in a pre-processed form (the Cornell
// box: 20 KiB), larger ones are read through L1/L2 as two-child records; see DevScene::variant.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <mutex>
#include <vector>

#include "hpt_device.h"

using namespace hpt;

// In host code this name is only a handle for hipGetSymbolAddress: the array's device address
// (per GPU) is looked up once per device in scratch_of() and only that pointer is passed on.
static __device__ __align__(256) unsigned char scratch[(1ull << 18) * 128ull + 4096ull * 4ull + 256ull];

namespace {

constexpr uint32_t kLdsLimit = 32u * 1024u;
constexpr uint32_t kLdsLimitTiny = 44u * 1024u;  // nine node copies of a 64-triangle scene (127 nodes) + its triangles
constexpr uint32_t kMaxLdsPerBlock = 160u * 1024u;  // LDS of a gfx950 compute unit; one workgroup may use all of it
constexpr uint32_t kMaxStackDepth = 40;  // per-lane stack entries (4 B each in LDS); median-split trees need ~log2(triangles)
// Traversal variants that run under the wave scheduler by default (bit = variant number), per mode:
// the two-child-record walks through L1/L2 (5) and their culled form (7).  Measured on MI355X: the
// scheduler gains 1.6x (bunny) and 1.55x / 2.2x (1.0 M-triangle height field, streamed / single
// kernel) there, where a walk is tens of dependent fetches long and most lanes are in the "step"
// state most of the time.  It LOSES where the walk is short and the shading weighs as much as the
// walk (Cornell box, LDS marking variants: 9.9 vs 29.2 Grays/s; LDS pair records: 7.0 vs 8.8):
// 64 lanes spread over four states leave every kind of round a third full, and a scheduling
// decision costs as much as a box step.  Those keep the lock-step kernels.
#ifndef HPT_MEGA_SCHEDULER_DEFAULT
#define HPT_MEGA_SCHEDULER_DEFAULT 0xA0u
#endif
#ifndef HPT_WF_SCHEDULER_DEFAULT
#define HPT_WF_SCHEDULER_DEFAULT 0xA0u
#endif
constexpr uint32_t kMegaSchedulerVariants = HPT_MEGA_SCHEDULER_DEFAULT, kWfSchedulerVariants = HPT_WF_SCHEDULER_DEFAULT;
constexpr int kStatClosest = 0, kStatShadow = 1, kStatPaths = 2, kStatBox = 3, kStatTri = 4, kNumStats = 6;

thread_local std::string g_last_error;

// Scratch of hpt_render_static, the entry point that allocates nothing: one work-unit counter,
// the per-segment live counters and the two wavefront path queues of a render of at most
// kStaticPaths camera paths.  Every GPU the library runs on holds its own copy of the array until
// the process ends; all calls on a GPU share it.
constexpr uint64_t kStaticPaths = 1ull << 18;      // 512 x 512 pixels x 1 sample
constexpr uint32_t kStaticSegments = 4096;         // live counters (one per resident workgroup)
constexpr size_t kScratchCounterBytes = 256;
constexpr size_t kScratchLiveBytes = kStaticSegments * sizeof(uint32_t);
constexpr size_t kScratchQueueBytes = 8ull * kStaticPaths * 16ull;  // 8 planes of float4: 32 MiB
constexpr size_t kScratchBytes = kScratchCounterBytes + kScratchLiveBytes + kScratchQueueBytes;
constexpr int kMaxDevices = 64;

static_assert(sizeof(scratch) == kScratchBytes, "scratch array and its layout constants disagree");

int fail(int code, const char* fmt, ...);

// device address of `scratch` on `device` (which must be current); looked up once per device
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
// All hpt_render_static calls on a GPU share its scratch: each call makes its stream wait for the event the
// previous call on that GPU recorded, then records it again.  One event per GPU, created here with the GPU's
// first context and kept like the scratch itself, so the call creates nothing.
hipEvent_t g_scratch_event[kMaxDevices] = {};
bool g_scratch_event_recorded[kMaxDevices] = {};
int scratch_of(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(HPT_ERR_INVALID_ARGUMENT, "device %d: the static scratch serves devices 0..%d", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(HPT_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
        hipEvent_t ev = nullptr;
        const hipError_t e2 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e2 != hipSuccess) return fail(HPT_ERR_HIP, "hipEventCreateWithFlags on device %d failed: %s", device, hipGetErrorString(e2));
        g_scratch_event[device] = ev;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return HPT_OK;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(HPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// device allocation: HPT_ERR_OUT_OF_MEMORY when the device has no room, HPT_ERR_HIP otherwise
// HPT_ALLOC_LIMIT=<bytes> (tests): a single allocation beyond it is refused as if the card were full.
int device_alloc(void** p, size_t bytes, const char* what) {
    *p = nullptr;
    if (const char* lim = getenv("HPT_ALLOC_LIMIT"))
        if (bytes > (size_t)strtoull(lim, nullptr, 0))
            return fail(HPT_ERR_OUT_OF_MEMORY, "out of device memory allocating %zu bytes for %s (HPT_ALLOC_LIMIT)", bytes, what);
    const hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e == hipSuccess) return HPT_OK;
    (void)hipGetLastError();  // the failed allocation must not poison the next hipGetLastError()
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation)
        return fail(HPT_ERR_OUT_OF_MEMORY, "out of device memory allocating %zu bytes for %s", bytes, what);
    return fail(HPT_ERR_HIP, "hipMalloc(%zu bytes, %s) failed: %s", bytes, what, hipGetErrorString(e));
}
#define ALLOC_TRY(ptr, bytes, what)                                          \
    do {                                                                     \
        if (int rc_ = device_alloc((void**)&(ptr), (bytes), (what))) return rc_; \
    } while (0)

// ------------------------------------------------------------------ kernels --

// pixel owned by a global lane index: 8x8 pixel tiles, one tile per wave, so
// that the 64 primary rays of a wave are coherent and neighbouring pixels
// (similar path-length statistics) share a wave
__device__ __forceinline__ bool lane_to_pixel(const DevScene& S, uint32_t g, uint32_t& pixel) {
    const uint32_t tiles_x = ((uint32_t)S.width + 7u) >> 3;
    const uint32_t tile = g >> 6, lane = g & 63u;
    const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u), y = (tile / tiles_x) * 8u + (lane >> 3);
    pixel = y * (uint32_t)S.width + x;
    return x < (uint32_t)S.width && y < (uint32_t)S.height;
}

// counters: one atomic per wave (every lane adding to the same address would serialise in L2).
// The wave sum is taken bit by bit over ballots, so lanes that do not exist (last wave of a
// workgroup size that is no multiple of 64) cannot contribute.  Called from wave-uniform code.
__device__ __forceinline__ void stat_add(unsigned long long* slot, uint32_t v) {
    unsigned long long sum = 0;
    for (int b = 0; b < 32; ++b) sum += (unsigned long long)__popcll(__ballot((v >> b) & 1u)) << b;
    if ((threadIdx.x & 63u) == 0u && sum) atomicAdd(slot, sum);
}

#ifdef HPT_COUNT_TESTS
__device__ __forceinline__ void count_flush(unsigned long long* stats, TestCount& c) {
    // low words through the wave sum, the (rare) high words lane by lane
    stat_add(stats + kStatBox, (uint32_t)c.box);
    stat_add(stats + kStatTri, (uint32_t)c.tri);
    if (c.box >> 32) atomicAdd(stats + kStatBox, (c.box >> 32) << 32);
    if (c.tri >> 32) atomicAdd(stats + kStatTri, (c.tri >> 32) << 32);
    c.box = c.tri = 0;
}
#define COUNT_DECL TestCount tcount{0ull, 0ull}
#define COUNT_FLUSH(stats) count_flush(stats, tcount)
#else
#define COUNT_DECL
#define COUNT_FLUSH(stats)
#endif

template <int kScene>
__global__ void k_intersect_batch(DevScene S, size_t n, const float* __restrict__ o, const float* __restrict__ d,
                                  const uint8_t* __restrict__ shadow, const double* __restrict__ tmax,
                                  int32_t* __restrict__ prim, double* __restrict__ t_out) {
    extern __shared__ float4 lds[];
    stage_scene<kScene>(S, lds);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const TriQueue tq = make_tri_queue<kScene>(reinterpret_cast<uint32_t*>(lds + S.lds_float4s));
#ifdef HPT_PROFILE
    __shared__ unsigned long long prof_dummy[3 * kProfSections];
    ProfAcc prof;
    prof.w = prof_dummy;
#endif
    COUNT_DECL;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const f3 ro = mk(o[3 * r], o[3 * r + 1], o[3 * r + 2]);
        const f3 rd = mk(d[3 * r], d[3 * r + 1], d[3 * r + 2]);
        int tri;
        double t;
        if (shadow && shadow[r]) trace<kScene, true>(S, lds, tq, ro, rd, tmax[r], tri, t PROF_PASS COUNT_PASS);
        else trace<kScene, false>(S, lds, tq, ro, rd, 0.0, tri, t PROF_PASS COUNT_PASS);
        prim[r] = tri < 0 ? -1 : __float_as_int(S.tris[4 * tri + 2].w);
        t_out[r] = t;
    }
}

// Scene::castRay (common/Scene.hpp:274-285) for caller-supplied eye rays, one lane per ray: the bounce loop of the render
// kernels on a path that starts at (o, d) instead of at the camera and draws from the keyed stream of
// (pixel_key, sample_key).  The radiance is written as floats, unnormalised.
template <int kScene, bool kExt>
__global__ void __launch_bounds__(256)
k_cast_rays(DevScene S, size_t n, const float* __restrict__ o, const float* __restrict__ d, const uint32_t* __restrict__ pixel_key,
            const uint32_t* __restrict__ sample_key, int max_depth, float* __restrict__ rgb, unsigned long long* __restrict__ stats) {
    extern __shared__ float4 lds[];
    stage_scene<kScene>(S, lds);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const TriQueue tq = make_tri_queue<kScene>(reinterpret_cast<uint32_t*>(lds + S.lds_float4s));
#ifdef HPT_PROFILE
    __shared__ unsigned long long prof_dummy[3 * kProfSections];
    ProfAcc prof;
    prof.w = prof_dummy;
#endif
    COUNT_DECL;
    uint32_t n_closest = 0, n_shadow = 0, n_paths = 0;  // per lane; a lane runs at most 2^32 / max_depth rays (checked by the host)
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        Path p;
        start_camera_path(S, pixel_key[r], sample_key[r], mk(d[3 * r], d[3 * r + 1], d[3 * r + 2]), p);
        p.o = mk(o[3 * r], o[3 * r + 1], o[3 * r + 2]);
        bool alive = max_depth > 0;
        while (alive) {
            int tri;
            double t;
            ++n_closest;
            trace<kScene, false>(S, lds, tq, p.o, p.d, 0.0, tri, t PROF_PASS COUNT_PASS);
            alive = shade_bounce<kScene, kExt>(S, lds, tq, p, tri, t, n_shadow PROF_PASS COUNT_PASS) && p.depth < max_depth;
        }
        rgb[3 * r] = p.rad.x;
        rgb[3 * r + 1] = p.rad.y;
        rgb[3 * r + 2] = p.rad.z;
        ++n_paths;
    }
    stat_add(stats + kStatClosest, n_closest);
    stat_add(stats + kStatShadow, n_shadow);
    stat_add(stats + kStatPaths, n_paths);
    COUNT_FLUSH(stats);
}

__device__ __forceinline__ void accumulate_pixel(int64_t* accum, uint32_t pixel, f3 rad) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(accum) + 3ull * pixel;
    atomicAdd(a + 0, (unsigned long long)radiance_to_fixed(rad.x));
    atomicAdd(a + 1, (unsigned long long)radiance_to_fixed(rad.y));
    atomicAdd(a + 2, (unsigned long long)radiance_to_fixed(rad.z));
}

// Waves per SIMD the single-kernel variant is compiled for.  4 = 118 VGPRs, no spills.  The
// kernel is VALU-issue bound: builds for 3 and for 5 waves (the latter spills to scratch) run
// within 2 % of it.
// Camera directions of the single-kernel mode are generated a whole wave at a time into a ring of 128 entries per wave
// (3 floats each) and handed to lanes as their paths end: generating them where they are needed costs a Philox block,
// two divisions and a normalisation for the five or six lanes of a wave that need a new path in a given iteration.
constexpr uint32_t kCamRingWords = 128u * 3u;

#ifndef HPT_MEGA_WAVES
#define HPT_MEGA_WAVES 4
#endif
// Mode 1.  Persistent waves.  A work unit is (8x8 pixel tile, chunk of s_chunk samples); waves
// draw unit numbers from a global counter until it runs past n_units.  The (pixel, sample) items
// of a wave's current unit form a pool that its lanes draw from: a lane whose path has ended
// takes the next item (ballot + prefix count), so all 64 lanes stay busy although path lengths
// vary from 1 to max_depth bounces.  When the pool runs dry the wave opens its next unit at
// once and fills idle lanes from it while the last paths of the previous unit finish: paths of
// two units (`cur`, `old`) can be in flight, each unit with its own 64-pixel accumulator in LDS
// (any lane may finish any pixel of its tile); a unit's accumulator is added to the global
// buffer when its last path has ended.  Units are numbered chunk-major, so the chunks of an
// expensive tile are spread over the launch.
template <int kScene, bool kExt, int kMaxBlock>
__global__ void __launch_bounds__(kMaxBlock, (kMaxBlock == 256 ? HPT_MEGA_WAVES : 1))
k_render_mega(DevScene S, int max_depth, uint32_t s_begin_all, uint32_t s_count_all, uint32_t s_chunk, uint32_t n_units,
              int64_t* __restrict__ accum, unsigned long long* __restrict__ stats, uint32_t* __restrict__ unit_counter) {
    extern __shared__ float4 lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = (blockDim.x + 63u) >> 6;
    const uint32_t wave_lanes = min(64u, blockDim.x - wave * 64u);  // the last wave of an odd-sized block is partial
    // two accumulators (unit generations 0 / 1) of 64 pixels x RGB per wave
    unsigned long long* tile_acc = reinterpret_cast<unsigned long long*>(lds + S.lds_float4s) + wave * 384u;
    for (uint32_t i = lane; i < 384u; i += wave_lanes) tile_acc[i] = 0ull;
    // triangle queues / stacks (variants that use them) lie behind all waves' accumulators
#ifdef HPT_PROFILE
    const uint32_t prof_words = waves_per_block * 3u * kProfSections * 2u;  // per-wave section counters, behind the accumulators
#else
    const uint32_t prof_words = 0u;
#endif
    uint32_t* const behind_acc = reinterpret_cast<uint32_t*>(lds + S.lds_float4s) + waves_per_block * 768u + prof_words;
    float* const cam_ring = reinterpret_cast<float*>(behind_acc) + wave * kCamRingWords;
    const TriQueue tq = make_tri_queue<kScene>(behind_acc + waves_per_block * kCamRingWords);
    stage_scene<kScene>(S, lds);

    const uint32_t W = (uint32_t)S.width, H = (uint32_t)S.height;
    const uint32_t tiles_x = (W + 7u) >> 3, n_tiles = tiles_x * ((H + 7u) >> 3);
    unsigned long long* out = reinterpret_cast<unsigned long long*>(accum);
    // add a finished unit's accumulator to the image and clear it for the unit after next
    auto flush = [&](unsigned long long* acc, uint32_t x0, uint32_t y0, uint32_t vw, uint32_t nv) {
        for (uint32_t j = lane; j < nv; j += wave_lanes) {
            const unsigned long long pix = (unsigned long long)(y0 + j / vw) * W + x0 + j % vw;
            for (uint32_t c = 0; c < 3u; ++c) {
                const unsigned long long v = acc[3u * j + c];
                if (v) atomicAdd(out + 3ull * pix + c, v);
                acc[3u * j + c] = 0ull;
            }
        }
    };
    // wave-uniform unit state: the unit whose pool is being handed out, and the one before it
    uint32_t cur_x0 = 0, cur_y0 = 0, cur_vw = 1, cur_nv = 0, cur_s0 = 0, cur_total = 0, cur_next = 0;
    uint32_t old_x0 = 0, old_y0 = 0, old_vw = 1, old_nv = 0;
    bool cur_open = false, old_open = false, queue_empty = false;
    uint32_t gen = 0;  // accumulator of `cur`; `old` uses the other one
    uint32_t ring_end = 0;  // items [.., ring_end) of `cur` have their camera directions in the ring (entry = item mod 128)
    // per lane: closest queries in the low, shadow queries in the high 16 bits of one register,
    // added to the global counters whenever the wave opens a unit (a wave runs at most
    // (s_chunk + 2) * max_depth <= 65535 loop iterations between two openings, checked by the host)
    uint32_t n_rays = 0, n_paths_wave = 0;
    uint32_t slot = 0, lane_gen = 0;
    COUNT_DECL;
    bool alive = false;
    Path p;
#ifdef HPT_PROFILE
    ProfAcc prof;
    prof.w = tile_acc + 384u * (waves_per_block - wave) + wave * 3u * kProfSections;
    for (uint32_t i = lane; i < 3u * kProfSections; i += wave_lanes) prof.w[i] = 0ull;
    const unsigned long long prof_start = __builtin_readcyclecounter();
#endif
    for (;;) {
        PROF_ITER(kProfDone);
        const unsigned long long need = __ballot(!alive);
        if (need) {
            if (!(cur_open && cur_next < cur_total) && !queue_empty) {
                // pool exhausted: retire `old` once its paths have ended, then open the next unit
                const bool old_busy = old_open && __ballot(alive && lane_gen != gen) != 0ull;
                if (!old_busy) {
                    if (old_open) {
                        flush(tile_acc + (gen ^ 1u) * 192u, old_x0, old_y0, old_vw, old_nv);
                        old_open = false;
                    }
                    uint32_t u = 0;
                    if (lane == 0) u = atomicAdd(unit_counter, 1u);
                    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
                    if (u < n_units) {
                        stat_add(stats + kStatClosest, n_rays & 0xFFFFu);
                        stat_add(stats + kStatShadow, n_rays >> 16);
                        n_rays = 0;
                        old_x0 = cur_x0;
                        old_y0 = cur_y0;
                        old_vw = cur_vw;
                        old_nv = cur_nv;
                        old_open = cur_open;
                        gen ^= 1u;
                        const uint32_t tile = u % n_tiles, chunk = u / n_tiles;
                        cur_x0 = (tile % tiles_x) * 8u;
                        cur_y0 = (tile / tiles_x) * 8u;
                        cur_vw = min(8u, W - cur_x0);
                        cur_nv = cur_vw * min(8u, H - cur_y0);  // pixels of an edge tile
                        cur_s0 = s_begin_all + chunk * s_chunk;
                        // u < n_units = tiles * ceil(s_count_all / s_chunk): the chunk starts inside the range
                        cur_total = cur_nv * min(s_chunk, s_count_all - chunk * s_chunk);
                        cur_next = 0;
                        ring_end = 0;
                        cur_open = true;
                        n_paths_wave += cur_total;  // every item of a pool is traced
                    } else {
                        queue_empty = true;
                    }
                }
            }
            if (cur_open && cur_next < cur_total) {
                // item k of the pool = (pixel slot, sample) of the unit
                auto item = [&](uint32_t k, uint32_t& sl, uint32_t& pixel, uint32_t& sample) {
                    uint32_t px, py, smp;
                    if (cur_nv == 64u) {  // full tile (wave-uniform): no integer divisions
                        sl = k & 63u;
                        smp = k >> 6;
                        px = sl & 7u;
                        py = sl >> 3;
                    } else {
                        sl = k % cur_nv;
                        smp = k / cur_nv;
                        px = sl % cur_vw;
                        py = sl / cur_vw;
                    }
                    pixel = (cur_y0 + py) * W + cur_x0 + px;
                    sample = cur_s0 + smp;
                };
                // the lanes to refill take at most wave_lanes items: have their directions in the ring.  A batch is
                // generated only when fewer than wave_lanes items are ahead, so ring_end stays below cur_next + 128
                // and no entry is overwritten before its item has been handed out.
                const uint32_t want_end = min(cur_total, cur_next + wave_lanes);
                while (ring_end < want_end) {
                    const uint32_t kk = ring_end + lane;
                    if (kk < cur_total) {
                        uint32_t sl, pixel, sample;
                        item(kk, sl, pixel, sample);
                        const f3 dir = camera_direction(S, pixel, sample);
                        float* e = cam_ring + 3u * (kk & 127u);
                        e[0] = dir.x;
                        e[1] = dir.y;
                        e[2] = dir.z;
                    }
                    ring_end += wave_lanes;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the ring is read by other lanes of the wave
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t k = cur_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                cur_next = min(cur_total, cur_next + (uint32_t)__popcll(need));
                if (!alive && k < cur_total) {
                    uint32_t pixel, sample;
                    item(k, slot, pixel, sample);
                    const float* e = cam_ring + 3u * (k & 127u);
                    start_camera_path(S, pixel, sample, mk(e[0], e[1], e[2]), p);
                    lane_gen = gen;
                    alive = true;
                }
            }
        }
        if (!__ballot(alive)) {
            if (queue_empty && !(cur_open && cur_next < cur_total)) break;  // reached by every wave: the counter only grows
            continue;
        }
        if (alive) {
            int tri;
            double t;
            ++n_rays;
            trace<kScene, false>(S, lds, tq, p.o, p.d, 0.0, tri, t PROF_PASS COUNT_PASS);
            alive = shade_bounce<kScene, kExt, 0x10000u>(S, lds, tq, p, tri, t, n_rays PROF_PASS COUNT_PASS) && p.depth < max_depth;
            if (!alive) {
                unsigned long long* acc = tile_acc + lane_gen * 192u + 3u * slot;
                atomicAdd(acc + 0u, (unsigned long long)radiance_to_fixed(p.rad.x));
                atomicAdd(acc + 1u, (unsigned long long)radiance_to_fixed(p.rad.y));
                atomicAdd(acc + 2u, (unsigned long long)radiance_to_fixed(p.rad.z));
            }
        }
    }
#ifdef HPT_PROFILE
    prof_add(prof, kProfDone, __builtin_readcyclecounter() - prof_start);
    if (lane == 0 && S.prof)
        for (int i = 0; i < 3 * kProfSections; ++i) atomicAdd(S.prof + i, prof.w[i]);
#endif
    if (old_open) flush(tile_acc + (gen ^ 1u) * 192u, old_x0, old_y0, old_vw, old_nv);
    if (cur_open) flush(tile_acc + gen * 192u, cur_x0, cur_y0, cur_vw, cur_nv);
    stat_add(stats + kStatClosest, n_rays & 0xFFFFu);
    stat_add(stats + kStatShadow, n_rays >> 16);
    stat_add(stats + kStatPaths, lane == 0 ? n_paths_wave : 0u);
    COUNT_FLUSH(stats);
}

// ---- wavefront: 64-byte path records as four float4 planes -----------------
struct WfQueue {
    float4* a;  // o.xyz, bsdfSamplePdf
    float4* b;  // d.xyz, bits(pixel)
    float4* c;  // throughput.xyz, bits(sample)
    float4* d;  // radiance.xyz, bits(depth | anyNonSpecularBounces << 30 | specularBounce << 31)
};
__device__ __forceinline__ void wf_store(const WfQueue& q, size_t i, const Path& p) {
    q.a[i] = make_float4(p.o.x, p.o.y, p.o.z, p.pdf);
    q.b[i] = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(p.pixel));
    q.c[i] = make_float4(p.thr.x, p.thr.y, p.thr.z, __uint_as_float(p.sample));
    q.d[i] = make_float4(p.rad.x, p.rad.y, p.rad.z, __uint_as_float((uint32_t)p.depth | (p.specular ? 0x80000000u : 0u) | (p.diffused ? 0x40000000u : 0u)));
}

struct WfRecord {
    float4 a, b, c, d;
};
// record i of the segment that starts at `base` (zeros beyond the segment's live count)
__device__ __forceinline__ WfRecord wf_fetch(const WfQueue& q, size_t base, uint32_t i, uint32_t count) {
    WfRecord r;
    r.a = r.b = r.c = r.d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < count) {
        r.a = q.a[base + i];
        r.b = q.b[base + i];
        r.c = q.c[base + i];
        r.d = q.d[base + i];
    }
    return r;
}
__device__ __forceinline__ void wf_unpack(const WfRecord& r, Path& p) {
    p.o = mk(r.a.x, r.a.y, r.a.z);
    p.pdf = r.a.w;
    p.d = mk(r.b.x, r.b.y, r.b.z);
    p.pixel = __float_as_uint(r.b.w);
    p.thr = mk(r.c.x, r.c.y, r.c.z);
    p.sample = __float_as_uint(r.c.w);
    p.rad = mk(r.d.x, r.d.y, r.d.z);
    const uint32_t f = __float_as_uint(r.d.w);
    p.depth = (int)(f & 0x3FFFFFFFu);
    p.specular = (f >> 31) != 0;
    p.diffused = (f & 0x40000000u) != 0;
}

// Wavefront queues are split into one SEGMENT per workgroup: workgroup b owns records
// [b * seg, (b + 1) * seg) of both queues and the counter live[bounce][b].  A path stays in the
// segment it was generated into, so compaction needs no global counter: the workgroup reserves
// output slots with an LDS atomic (a single global counter serialises one atomic per wave and
// bounce, 12 ns each: that alone took as long as the whole render).  Paths are dealt to the
// segments in groups of 64 round-robin, so every segment holds the same mix of pixels and the
// segments drain at the same rate.

// GenerateCameraRay (gpu_path_tracer/CudaPathTrace.cu:34-62) for paths [first, first + count) of
// the render's linear path index space (sample-major; the low index walks 8x8 pixel tiles).
// Group g (64 consecutive paths = one pixel tile of one sample) goes to position k = g / G of
// segment (g + k) % G, G = gridDim.x: the skew by k keeps a segment from collecting the same
// tile column of every sample when G divides the number of tiles.
__global__ void k_wf_generate(DevScene S, uint32_t s_begin, uint64_t first, uint32_t count, uint32_t lanes_per_sample,
                              uint32_t seg, WfQueue q, uint32_t* __restrict__ live, unsigned long long* __restrict__ stats) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = (blockDim.x + 63u) >> 6;
    const uint32_t groups = (count + 63u) >> 6, G = gridDim.x;
    const uint32_t rows = groups / G, rem = groups % G;  // full rows of G groups, groups of the last row
    const uint32_t mine = rows + ((blockIdx.x + G - rows % G) % G < rem ? 1u : 0u);  // groups of this segment
    uint32_t made = 0;
    if (lane < min(64u, blockDim.x - wave * 64u)) {  // (always, for the block sizes the host uses: multiples of 64)
        for (uint32_t k = wave; k < mine; k += waves) {
            const uint32_t i = (k * G + (blockIdx.x + G - k % G) % G) * 64u + lane;  // path of the batch
            Path p;
            bool valid = false;
            uint32_t sample = s_begin;
            if (i < count) {
                const uint64_t g = first + i;
                sample = s_begin + (uint32_t)(g / lanes_per_sample);
                uint32_t pixel;
                valid = lane_to_pixel(S, (uint32_t)(g % lanes_per_sample), pixel);
                if (valid) generate_camera_path(S, pixel, sample, p);
            }
            if (valid) {
                ++made;
            } else {
                // lanes of partial edge tiles (and the tail of the last group) own no pixel: they get
                // a dead record that the first bounce drops without tracing
                p.o = p.d = p.thr = p.rad = mk(0.f);
                p.pdf = 0.f;
                p.depth = 0x3FFFFFFF;
                p.specular = false;
                p.diffused = false;
                p.pixel = 0xFFFFFFFFu;
                p.sample = sample;
            }
            wf_store(q, (size_t)blockIdx.x * seg + k * 64u + lane, p);
        }
    }
    if (threadIdx.x == 0) live[blockIdx.x] = mine * 64u;
    stat_add(stats + kStatPaths, made);
}

// All bounces of a segment's paths: per bounce ComputeIntersections + IntegratePathSegment + the
// two thrust compactions + FinalGather (gpu_path_tracer/CudaPathTrace.cu:64-111, :149-157) in one
// pass.  A lane takes a path record of its workgroup's segment, finds the closest hit, shades it
// (light sample with the inline any-hit shadow query, BSDF sample) and then either appends the
// survivor to the segment of the other queue or adds the finished path's radiance to its pixel
// (int64 atomics).  Survivors are compacted with __ballot + prefix popcount and one LDS atomic
// per wave.  Segments are independent, so the bounce loop runs inside the kernel with workgroup
// barriers only: one launch per batch, the scene staged once, no workgroup waits for another.
#ifndef HPT_WF_WAVES
#define HPT_WF_WAVES 1
#endif
template <int kScene, bool kExt>
__global__ void __launch_bounds__(256, HPT_WF_WAVES)
k_wf_bounces(DevScene S, int max_depth, uint32_t seg, WfQueue q0, WfQueue q1, const uint32_t* __restrict__ live,
             int64_t* __restrict__ accum, unsigned long long* __restrict__ stats) {
    extern __shared__ float4 lds[];
    __shared__ uint32_t s_out, s_next;
    uint32_t count = live[blockIdx.x];
    if (count == 0) return;
    stage_scene<kScene>(S, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const size_t seg_base = (size_t)blockIdx.x * seg;
    const TriQueue tq = make_tri_queue<kScene>(reinterpret_cast<uint32_t*>(lds + S.lds_float4s));
    uint32_t n_closest = 0, n_shadow = 0;
    COUNT_DECL;
#ifdef HPT_PROFILE
    __shared__ unsigned long long prof_dummy[3 * kProfSections];
    ProfAcc prof;
    prof.w = prof_dummy;
#endif
    const uint32_t wave_lanes = min(64u, blockDim.x - (threadIdx.x - lane));  // the last wave of an odd-sized workgroup is partial
    for (int bounce = 0; bounce < max_depth && count; ++bounce) {
        const WfQueue& q_in = (bounce & 1) ? q1 : q0;
        const WfQueue& q_out = (bounce & 1) ? q0 : q1;
        if (threadIdx.x == 0) {
            s_out = 0;
            s_next = 0;
        }
        __syncthreads();  // also: the records the previous bounce stored are visible to the whole workgroup
        // waves take groups of records (one per lane) as they become free: their traversal times differ
        for (;;) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, wave_lanes);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= count) break;
            const uint32_t i = base + lane;
            const WfRecord cur = wf_fetch(q_in, seg_base, i, count);
            bool keep = false;
            Path p;
            if (i < count) {
                wf_unpack(cur, p);
                if (p.pixel != 0xFFFFFFFFu) {  // not a dead record
                    int tri;
                    double t;
                    ++n_closest;
                    trace<kScene, false>(S, lds, tq, p.o, p.d, 0.0, tri, t PROF_PASS COUNT_PASS);
                    keep = shade_bounce<kScene, kExt>(S, lds, tq, p, tri, t, n_shadow PROF_PASS COUNT_PASS) && p.depth < max_depth;
                    if (!keep) accumulate_pixel(accum, p.pixel, p.rad);
                }
            }
            const unsigned long long mask = __ballot(keep);
            if (mask) {
                const int leader = __ffsll((long long)mask) - 1;
                uint32_t slot = 0;
                if ((int)lane == leader) slot = atomicAdd(&s_out, (uint32_t)__popcll(mask));
                slot = __shfl(slot, leader);
                if (keep) wf_store(q_out, seg_base + slot + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)), p);
            }
        }
        __syncthreads();
        count = s_out;  // survivors = the next bounce's paths
        __syncthreads();
    }
    stat_add(stats + kStatClosest, n_closest);
    stat_add(stats + kStatShadow, n_shadow);
    COUNT_FLUSH(stats);
}

// ---- wave scheduler (scenes read through two-child records) -----------------
// In the kernels above a lane traces its ray from start to end inside trace<>, so a wave walks
// until its slowest lane is done: on the bunny and the height fields only a fifth of the VALU
// lanes were active (PMC).  Here every lane carries a small state machine
//     IDLE -> WALK(closest) -> SHADE_A -> WALK(any-hit) -> SHADE_B -> IDLE | WALK(closest)
// and the WAVE decides what to run next from how many lanes want what: a round of box steps
// (walk_step) for the lanes that can step, a round of triangle tests for the lanes whose queue
// is full or whose walk has ended, the first or second half of the shading for the lanes that
// wait for it, or a refill of the idle lanes with new paths.  A kind of round runs when enough
// lanes want it (thresholds below); when none reaches its threshold the most wanted one runs, so
// some lane always advances.  A lane whose ray ends early no longer idles to the end of the
// slowest ray of its wave: its next ray joins the following rounds.
#ifndef HPT_SM_T_STEP
#define HPT_SM_T_STEP 24
#endif
#ifndef HPT_SM_T_TRI
#define HPT_SM_T_TRI 12
#endif
#ifndef HPT_SM_T_SHADE
#define HPT_SM_T_SHADE 32
#endif
#ifndef HPT_SM_T_REFILL
#define HPT_SM_T_REFILL 24
#endif
enum { kLaneIdle = 0, kLaneWalk = 1, kLaneShadeA = 2, kLaneShadeB = 3 };

// The scheduler loop of one wave.  `Policy` is the mode's side of it:
//   uint32_t refillable(m_idle, m_busy)   how many of the idle lanes could be given a path now (wave-uniform)
//   void refill(idle, p, got)             collective: hand paths to idle lanes (got = this lane received one)
//   void count_closest() / count_shadow() ray counters
//   void finish(p)                        per lane: the path has ended, add its radiance to the picture
//   bool next_bounce(keep, p)             collective: the bounce is shaded and the path goes on (keep); returns
//                                         whether THIS lane carries the path into its next bounce itself
template <int kScene, bool kExt, class Policy>
__device__ __forceinline__ void schedule_wave(const DevScene& S, const float4* lds, TriQueue tq, int max_depth, Policy& pol COUNT_ARG) {
    using Trav = Traversal<kScene>;
    typename Trav::State W;
    Trav::init(W, tq);
    const ShadeSrc src = shade_source<kScene, kExt>(S, lds);
    int st = kLaneIdle;
    Path p;
    int hit_tri = -1;
    double hit_t = 0.0;
    constexpr bool kHave = Trav::kKeepsHitData && !kExt;  // the walk keeps the hit's normal and material index: they go into `ls` when it ends
    LightSample ls{};
    bool lit = false;  // kLaneShadeB: a shadow ray was traced (its answer is in W)
    // walks that have just ended: the marking variants' back-face filter, then the lane goes to the shading
    auto finish_walks = [&]() {
        if (st == kLaneWalk) {
            Trav::settle(S, lds, W COUNT_PASS);
            if (Trav::done(W, tq)) {
                if (W.shadow) {
                    st = kLaneShadeB;
                } else {
                    hit_tri = W.best;
                    hit_t = W.best_t;
                    if constexpr (kHave) {
                        ls.normal = W.best_n;
                        ls.material = W.best_m;
                    }
                    st = kLaneShadeA;
                }
            }
        }
    };
    for (;;) {
        // The walking rounds form a loop of their own: most rounds are box steps and triangle tests, and inside this
        // loop the path and shading state is untouched, so it stays where it is while the wave alternates between them.
        int what;  // 0 refill, 1 step, 2 triangles, 3 shade A, 4 shade B, -1 nothing left
        for (;;) {
            const unsigned long long m_walk = lane_mask(st == kLaneWalk);
            const uint32_t c_step = lane_count(m_walk & Trav::can_step_mask(W, tq));
            const uint32_t c_tri = lane_count(m_walk & Trav::wants_test_mask(W, tq));
            const uint32_t c_a = lane_count(lane_mask(st == kLaneShadeA));
            const uint32_t c_b = lane_count(lane_mask(st == kLaneShadeB));
            const unsigned long long m_idle = lane_mask(st == kLaneIdle);
            const uint32_t c_idle = m_idle ? pol.refillable(m_idle, lane_mask(st != kLaneIdle)) : 0u;
            // what to run: the first kind of round that reaches its threshold, else the most wanted
            if (c_idle >= HPT_SM_T_REFILL) what = 0;
            else if (c_step >= HPT_SM_T_STEP) what = 1;
            else if (c_tri >= HPT_SM_T_TRI) what = 2;
            else if (c_b >= HPT_SM_T_SHADE) what = 4;
            else if (c_a >= HPT_SM_T_SHADE) what = 3;
            else {
                uint32_t best = c_step;
                what = 1;
                if (c_tri > best) { best = c_tri; what = 2; }
                if (c_b > best) { best = c_b; what = 4; }
                if (c_a > best) { best = c_a; what = 3; }
                if (c_idle > best) { best = c_idle; what = 0; }
                if (best == 0u) what = -1;  // every lane idle and nothing left to hand out
            }
            if (what == 1) {
                // rounds of box steps while enough lanes take part
                do {
                    const bool stepping = st == kLaneWalk && Trav::can_step(W, tq);
                    Trav::fetch(S, lds, tq, W, stepping);  // every lane of the wave (the cooperative record fetch; nothing otherwise)
                    if (stepping) Trav::step(S, lds, tq, W COUNT_PASS);
                } while (lane_count(m_walk & Trav::can_step_mask(W, tq)) >= HPT_SM_T_STEP);  // (no lane changes its state in here)
            } else if (what == 2) {
                // rounds of triangle tests: every lane that has a triangle to test takes part (also lanes
                // that could still step: the round costs the same), until too few lanes are blocked
                do {
                    if (st == kLaneWalk && Trav::may_test(W, tq)) Trav::template test_one<kExt>(S, lds, tq, W COUNT_PASS);
                } while (lane_count(m_walk & Trav::wants_test_mask(W, tq)) >= HPT_SM_T_TRI);
            } else {
                break;
            }
            finish_walks();
        }
        if (what < 0) break;
        if (what == 0) {
            bool got = false;
            pol.refill(st == kLaneIdle, p, got);
            if (got) {
                pol.count_closest();
                Trav::begin(S, lds, tq, W, p.o, p.d, false, 0.f COUNT_PASS);
                st = kLaneWalk;
            }
        } else if (what == 3) {
            if (st == kLaneShadeA) {
                f3 so, wi;
                float tmax;
                const int rc = shade_a<kExt, kHave>(S, src, p, hit_tri, hit_t, so, wi, tmax, ls);
                if (rc == kShadeMiss) {
                    pol.finish(p);
                    st = kLaneIdle;
                } else if (rc == kShadeShadowRay) {
                    pol.count_shadow();
                    Trav::begin(S, lds, tq, W, so, wi, true, tmax COUNT_PASS);
                    lit = true;
                    st = kLaneWalk;
                } else {
                    lit = false;
                    st = kLaneShadeB;
                }
            }
        } else {
            bool keep = false;
            const bool shading = st == kLaneShadeB;
            if (shading) {
                shade_b<kExt>(S, src, p, hit_tri, hit_t, lit && W.best < 0, W.d, ls);
                keep = p.depth < max_depth;
                if (!keep) pol.finish(p);
                st = kLaneIdle;
            }
            if (pol.next_bounce(keep, p)) {  // this lane traces the path's next bounce itself
                pol.count_closest();
                Trav::begin(S, lds, tq, W, p.o, p.d, false, 0.f COUNT_PASS);
                st = kLaneWalk;
            }
        }
        finish_walks();  // (a walk can end where it begins: the root's box is missed)
    }
}

// ---- streamed mode under the scheduler ----
struct WfPolicy {
    const WfQueue& q_in;
    const WfQueue& q_out;
    size_t seg_base;
    uint32_t count;                 // records of this bounce in the segment
    uint32_t* s_next;               // LDS: next record nobody has taken
    uint32_t* s_out;                // LDS: survivors stored so far
    int64_t* accum;
    uint32_t lane;
    bool more = true;               // wave-uniform: the segment may still hold records nobody has taken
    uint32_t n_closest = 0, n_shadow = 0;
    __device__ __forceinline__ uint32_t refillable(unsigned long long m_idle, unsigned long long) const { return more ? lane_count(m_idle) : 0u; }
    __device__ __forceinline__ void refill(bool idle, Path& p, bool& got) {
        const unsigned long long m = __ballot(idle);
        const uint32_t n = (uint32_t)__popcll(m);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(s_next, n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base + n >= count) more = false;
        if (idle) {
            const uint32_t i = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (i < count) {
                const WfRecord cur = wf_fetch(q_in, seg_base, i, count);
                wf_unpack(cur, p);
                got = p.pixel != 0xFFFFFFFFu;  // not a dead record
            }
        }
    }
    __device__ __forceinline__ void count_closest() { ++n_closest; }
    __device__ __forceinline__ void count_shadow() { ++n_shadow; }
    __device__ __forceinline__ void finish(const Path& p) { accumulate_pixel(accum, p.pixel, p.rad); }
    // survivors are compacted into the other queue: ballot, prefix count, one LDS atomic per wave
    __device__ __forceinline__ bool next_bounce(bool keep, const Path& p) {
        const unsigned long long mask = __ballot(keep);
        if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            uint32_t slot = 0;
            if ((int)lane == leader) slot = atomicAdd(s_out, (uint32_t)__popcll(mask));
            slot = __shfl(slot, leader);
            if (keep) wf_store(q_out, seg_base + slot + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)), p);
        }
        return false;
    }
};

#ifndef HPT_SM_WAVES
#define HPT_SM_WAVES 1
#endif
// All bounces of a segment's paths, like k_wf_bounces, under the wave scheduler.
template <int kScene, bool kExt>
__global__ void __launch_bounds__(256, HPT_SM_WAVES)
k_wf_bounces_sm(DevScene S, int max_depth, uint32_t seg, WfQueue q0, WfQueue q1, const uint32_t* __restrict__ live,
                int64_t* __restrict__ accum, unsigned long long* __restrict__ stats) {
    extern __shared__ float4 lds[];
    __shared__ uint32_t s_out, s_next;
    uint32_t count = live[blockIdx.x];
    if (count == 0) return;
    stage_scene<kScene>(S, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const size_t seg_base = (size_t)blockIdx.x * seg;
    const TriQueue tq = make_tri_queue<kScene>(reinterpret_cast<uint32_t*>(lds + S.lds_float4s));
    uint32_t n_closest = 0, n_shadow = 0;
    COUNT_DECL;
    for (int bounce = 0; bounce < max_depth && count; ++bounce) {
        if (threadIdx.x == 0) {
            s_out = 0;
            s_next = 0;
        }
        __syncthreads();  // also: the records the previous bounce stored are visible to the whole workgroup
        WfPolicy pol{(bounce & 1) ? q1 : q0, (bounce & 1) ? q0 : q1, seg_base, count, &s_next, &s_out, accum, lane};
        schedule_wave<kScene, kExt>(S, lds, tq, max_depth, pol COUNT_PASS);
        n_closest += pol.n_closest;
        n_shadow += pol.n_shadow;
        __syncthreads();
        count = s_out;  // survivors = the next bounce's paths
        __syncthreads();
    }
    stat_add(stats + kStatClosest, n_closest);
    stat_add(stats + kStatShadow, n_shadow);
    COUNT_FLUSH(stats);
}

// ---- single-kernel mode under the scheduler ----
// The work pool of k_render_mega (units of (8x8 pixel tile, s_chunk samples) drawn from a global counter) as the
// scheduler's source of paths.  A finished path adds its radiance to its pixel with 64-bit global atomics, as the
// streamed mode does: the per-tile accumulators k_render_mega keeps in LDS (3 KiB per wave) would cost these
// kernels, whose per-lane stacks and queues fill the LDS, a workgroup per compute unit (1.0 M-triangle height
// field: three workgroups instead of four), and the atomics are few next to the record fetches of such scenes.
struct MegaPolicy {
    const DevScene& S;
    uint32_t s_begin_all, s_count_all, s_chunk, n_units;
    int64_t* accum;
    unsigned long long* stats;
    uint32_t* unit_counter;
    uint32_t lane, wave_lanes, W, H, tiles_x, n_tiles;
    // wave-uniform: the unit whose pool is being handed out
    uint32_t cur_x0 = 0, cur_y0 = 0, cur_vw = 1, cur_nv = 0, cur_s0 = 0, cur_total = 0, cur_next = 0;
    bool cur_open = false, queue_empty = false;
    // per lane
    uint32_t n_closest = 0, n_shadow = 0, n_paths_wave = 0;

    __device__ __forceinline__ bool pool_has_items() const { return cur_open && cur_next < cur_total; }
    __device__ __forceinline__ uint32_t refillable(unsigned long long m_idle, unsigned long long) const {
        return (pool_has_items() || !queue_empty) ? lane_count(m_idle) : 0u;
    }
    __device__ __forceinline__ void refill(bool idle, Path& p, bool& got) {
        const unsigned long long need = __ballot(idle);
        if (!pool_has_items() && !queue_empty) {
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(unit_counter, 1u);
            u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
            if (u < n_units) {
                // the per-lane ray counters are emptied here: between two openings a lane traces at most
                // 64 * s_chunk * max_depth rays (< 2^32 for the sizes the host allows)
                stat_add(stats + kStatClosest, n_closest);
                stat_add(stats + kStatShadow, n_shadow);
                n_closest = n_shadow = 0;
                const uint32_t tile = u % n_tiles, chunk = u / n_tiles;
                cur_x0 = (tile % tiles_x) * 8u;
                cur_y0 = (tile / tiles_x) * 8u;
                cur_vw = min(8u, W - cur_x0);
                cur_nv = cur_vw * min(8u, H - cur_y0);  // pixels of an edge tile
                cur_s0 = s_begin_all + chunk * s_chunk;
                cur_total = cur_nv * min(s_chunk, s_count_all - chunk * s_chunk);
                cur_next = 0;
                cur_open = true;
                n_paths_wave += cur_total;  // every item of a pool is traced
            } else {
                queue_empty = true;
            }
        }
        if (pool_has_items()) {
            const uint32_t k = cur_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            cur_next = min(cur_total, cur_next + (uint32_t)__popcll(need));
            if (idle && k < cur_total) {
                uint32_t slot, px, py, smp;
                if (cur_nv == 64u) {  // full tile (wave-uniform): no integer divisions
                    slot = k & 63u;
                    smp = k >> 6;
                    px = slot & 7u;
                    py = slot >> 3;
                } else {
                    slot = k % cur_nv;
                    smp = k / cur_nv;
                    px = slot % cur_vw;
                    py = slot / cur_vw;
                }
                generate_camera_path(S, (cur_y0 + py) * W + cur_x0 + px, cur_s0 + smp, p);
                got = true;
            }
        }
    }
    __device__ __forceinline__ void count_closest() { ++n_closest; }
    __device__ __forceinline__ void count_shadow() { ++n_shadow; }
    __device__ __forceinline__ void finish(const Path& p) { accumulate_pixel(accum, p.pixel, p.rad); }
    __device__ __forceinline__ bool next_bounce(bool keep, const Path&) const { return keep; }  // the lane keeps its path
    __device__ __forceinline__ void close() {
        stat_add(stats + kStatClosest, n_closest);
        stat_add(stats + kStatShadow, n_shadow);
        stat_add(stats + kStatPaths, lane == 0 ? n_paths_wave : 0u);
    }
};

#ifndef HPT_MEGA_SM_WAVES
#define HPT_MEGA_SM_WAVES 4
#endif
template <int kScene, bool kExt, int kMaxBlock>
__global__ void __launch_bounds__(kMaxBlock, (kMaxBlock == 256 ? HPT_MEGA_SM_WAVES : 1))
k_render_mega_sm(DevScene S, int max_depth, uint32_t s_begin_all, uint32_t s_count_all, uint32_t s_chunk, uint32_t n_units,
                 int64_t* __restrict__ accum, unsigned long long* __restrict__ stats, uint32_t* __restrict__ unit_counter) {
    extern __shared__ float4 lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wave_lanes = min(64u, blockDim.x - wave * 64u);  // the last wave of an odd-sized block is partial
    const TriQueue tq = make_tri_queue<kScene>(reinterpret_cast<uint32_t*>(lds + S.lds_float4s));
    stage_scene<kScene>(S, lds);
    const uint32_t W = (uint32_t)S.width, H = (uint32_t)S.height;
    const uint32_t tiles_x = (W + 7u) >> 3;
    MegaPolicy pol{S, s_begin_all, s_count_all, s_chunk, n_units, accum, stats, unit_counter, lane, wave_lanes, W, H, tiles_x, tiles_x * ((H + 7u) >> 3)};
    COUNT_DECL;
    schedule_wave<kScene, kExt>(S, lds, tq, max_depth, pol COUNT_PASS);
    pol.close();
    COUNT_FLUSH(stats);
}

// dst += src (two accumulation buffers on one device)
__global__ void k_accum_add(int64_t* __restrict__ dst, const int64_t* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

__global__ void k_resolve(const int64_t* __restrict__ accum, size_t n, double denom, float* __restrict__ rgb) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        rgb[i] = (float)((double)accum[i] / denom);
}

}  // namespace

// variants that have a resumable Traversal<> and therefore run under the wave scheduler
constexpr bool has_scheduler(int variant) {
    return variant == kSceneTiny32 || variant == kSceneTiny64 || variant == kScenePair || variant == kScenePairLds || variant == kScenePairCull;
}

// run CALL with kS bound to the scene's traversal variant as a compile-time constant
#define HPT_DISPATCH_SCENE(variant, ...)                                                 \
    switch (variant) {                                                                   \
    case kSceneGlobal: { constexpr int kS = kSceneGlobal; __VA_ARGS__; } break;          \
    case kSceneLds: { constexpr int kS = kSceneLds; __VA_ARGS__; } break;                \
    case kSceneTiny32: { constexpr int kS = kSceneTiny32; __VA_ARGS__; } break;          \
    case kSceneTiny64: { constexpr int kS = kSceneTiny64; __VA_ARGS__; } break;          \
    case kScenePair: { constexpr int kS = kScenePair; __VA_ARGS__; } break;              \
    case kScenePairCull: { constexpr int kS = kScenePairCull; __VA_ARGS__; } break;      \
    default: { constexpr int kS = kScenePairLds; __VA_ARGS__; } break;                   \
    }
// the same with kG = "the scene has a Glass material" for the kernels that shade
#define HPT_DISPATCH_RENDER(variant, glass, ...)                                         \
    if (glass) {                                                                         \
        constexpr bool kG = true;                                                        \
        HPT_DISPATCH_SCENE(variant, __VA_ARGS__)                                         \
    } else {                                                                             \
        constexpr bool kG = false;                                                       \
        HPT_DISPATCH_SCENE(variant, __VA_ARGS__)                                         \
    }

// csrc/hpt_bvh_build.hip
namespace hpt_build {
std::string build_pairs(float4* tris, float4* v12, uint32_t n, const uint32_t* obj_first, uint32_t num_objects, float4* nodes, uint32_t num_nodes,
                        float4* rec_out, uint32_t* stack_depth, uint32_t* stack_depth_any, float root_lo[3], float root_hi[3], double* build_ms);
}

// ------------------------------------------------------------------ context --

struct hpt_ctx {
    int device = 0;
    int num_cus = 256;
    bool has_scene = false;
    int traversal = HPT_TRAVERSAL_REFERENCE;
    int bvh_build = HPT_BVH_BUILD_HOST;   // who derives the traversal records at hpt_scene_upload
    double bvh_build_ms = 0.0;            // device time of the last GPU build
    // the scene has a Glass material or a sphere, or the integrator runs Russian roulette / path
    // regularisation: selects the kernels built with kExt, which carry those branches
    bool extended = false;
    bool scene_extended = false;      // the scene's share of `extended`
    bool has_sphere = false;          // a record of the triangle array is a sphere: no marking variant
    uint32_t integrator = 0;          // kIntegratorRoulette | kIntegratorRegularize (hpt_set_integrator)
    float russian_roulette = 0.8f;    // Scene::RussianRoulette (common/Scene.hpp:30)
    DevScene S{};
    // scene arrays (device)
    float4* d_nodes = nullptr;
    float* d_node_area = nullptr;
    float4* d_tris = nullptr;
    float4* d_tri_v12 = nullptr;
    float4* d_pairs = nullptr;
    float4* d_ref_nodes = nullptr;  // the reference's tree beside the caller's (threaded walk through L1/L2 only)
    float4* d_fast_light = nullptr; // DevScene::fast_light
    float4* d_oct_nodes = nullptr;
    uint32_t* d_oct_table = nullptr;
    DevMaterial* d_mats = nullptr;
    DevLight* d_lights = nullptr;
    // accumulation + stats
    int64_t* d_accum = nullptr;
    size_t accum_elems = 0;
    unsigned long long* d_stats = nullptr;
    uint32_t* d_unit_counter = nullptr;  // work-unit counter of the single-kernel mode
    unsigned long long* d_prof = nullptr;  // diagnostic builds only
    float* d_rgb = nullptr;  // staging for hpt_resolve
    // wavefront state
    // paths per wavefront batch: 2 queues x 64 B each = 32 GiB of the 288 GB when a render has that many paths (the
    // queues are allocated for what a call needs, up to this).  Long batches matter: every bounce of a batch ends in
    // a workgroup barrier and every batch in a tail of few long paths (bunny 4.6 -> 5.5, 1.0 M-triangle height field
    // 3.4 -> 4.5 Grays/s from 2^24 to 2^28)
    uint64_t wf_capacity = 1ull << 28;
    uint64_t wf_allocated = 0;
    float4* d_wf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_live = nullptr;  // per-bounce live counters
    size_t live_slots = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    hipStream_t last_stream = nullptr;  // stream of the most recent hpt_render
    unsigned char* scratch_dev = nullptr;  // this GPU's copy of the static scratch (hpt_render_static)
    std::vector<hipEvent_t> free_events;
    double kernel_ms = 0.0;
    uint32_t launches = 0;
    uint32_t spp_per_launch = 0;    // samples per pixel per single-kernel launch (HPT_SPP_PER_LAUNCH); 0 = about 2^28 paths
    uint32_t spp_per_unit = 16;     // samples per pixel per wave work unit (HPT_SPP_PER_UNIT)
};

namespace {

int set_device(hpt_ctx* ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return HPT_OK;
}

void free_scene(hpt_ctx* ctx) {
    (void)hipFree(ctx->d_nodes);
    (void)hipFree(ctx->d_node_area);
    (void)hipFree(ctx->d_tris);
    (void)hipFree(ctx->d_tri_v12);
    (void)hipFree(ctx->d_pairs);
    (void)hipFree(ctx->d_ref_nodes);
    ctx->d_ref_nodes = nullptr;
    (void)hipFree(ctx->d_fast_light);
    ctx->d_fast_light = nullptr;
    (void)hipFree(ctx->d_oct_nodes);
    (void)hipFree(ctx->d_oct_table);
    (void)hipFree(ctx->d_mats);
    (void)hipFree(ctx->d_lights);
    (void)hipFree(ctx->d_accum);
    (void)hipFree(ctx->d_rgb);
    ctx->d_nodes = nullptr;
    ctx->d_node_area = nullptr;
    ctx->d_tris = nullptr;
    ctx->d_tri_v12 = nullptr;
    ctx->d_pairs = nullptr;
    ctx->d_oct_nodes = nullptr;
    ctx->d_oct_table = nullptr;
    ctx->d_mats = nullptr;
    ctx->d_lights = nullptr;
    ctx->d_accum = nullptr;
    ctx->d_rgb = nullptr;
    ctx->accum_elems = 0;
    ctx->has_scene = false;
}

void free_wavefront(hpt_ctx* ctx) {
    for (auto& p : ctx->d_wf) {
        (void)hipFree(p);
        p = nullptr;
    }
    (void)hipFree(ctx->d_live);
    ctx->d_live = nullptr;
    ctx->wf_allocated = 0;
    ctx->live_slots = 0;
}

int ensure_wavefront(hpt_ctx* ctx, uint64_t paths, uint32_t segments) {
    if (ctx->wf_allocated < paths) {
        for (auto& p : ctx->d_wf) {
            (void)hipFree(p);
            p = nullptr;
        }
                ctx->wf_allocated = 0;
        for (auto& p : ctx->d_wf) ALLOC_TRY(p, paths * sizeof(float4), "wavefront path queues");
        ctx->wf_allocated = paths;
    }
    const size_t want = segments;  // one live counter per segment (written by the generate kernel)
    if (ctx->live_slots < want) {
        (void)hipFree(ctx->d_live);
        ctx->d_live = nullptr;
        ctx->live_slots = 0;
        ALLOC_TRY(ctx->d_live, want * sizeof(uint32_t), "live");
        ctx->live_slots = want;
    }
    return HPT_OK;
}

// validate the flattened tree on the host so that no kernel can walk out of
// bounds or loop forever on a malformed upload
int validate_scene(const hpt_scene_desc* sc) {
    if (!sc) return fail(HPT_ERR_INVALID_ARGUMENT, "scene is null");
    if (sc->width <= 0 || sc->height <= 0 || (uint64_t)sc->width * (uint64_t)sc->height > (1ull << 28))
        return fail(HPT_ERR_INVALID_ARGUMENT, "bad resolution %dx%d", sc->width, sc->height);
    if (sc->num_nodes == 0 || sc->num_tris == 0 || !sc->nodes || !sc->tris || !sc->tri_v12 || !sc->node_area)
        return fail(HPT_ERR_INVALID_ARGUMENT, "scene has no nodes/triangles");
    if (sc->num_nodes > (1u << 30) || sc->num_tris > (1u << 29)) return fail(HPT_ERR_INVALID_ARGUMENT, "scene too large");
    if (sc->num_materials == 0 || !sc->materials) return fail(HPT_ERR_INVALID_ARGUMENT, "scene has no materials");
    if (sc->num_lights && !sc->lights) return fail(HPT_ERR_INVALID_ARGUMENT, "lights pointer is null");
    for (uint32_t m = 0; m < sc->num_materials; ++m)
        if (sc->materials[m].type < 0 || sc->materials[m].type > 2)
            return fail(HPT_ERR_INVALID_ARGUMENT, "material %u: type %d is none of 0 Lambertian, 1 MetallicWorkflow, 2 Glass", m,
                        sc->materials[m].type);
    auto check_tree = [&](const float* nodes, uint32_t num_nodes, const char* name) -> int {
        const int32_t nn = (int32_t)num_nodes;
        for (int32_t i = 0; i < nn; ++i) {
            int32_t miss, tri;
            std::memcpy(&miss, &nodes[8 * (size_t)i + 3], 4);
            std::memcpy(&tri, &nodes[8 * (size_t)i + 7], 4);
            if (miss <= i || miss > nn) return fail(HPT_ERR_INVALID_ARGUMENT, "%s %d: miss link %d does not point forward", name, i, miss);
            if (tri >= (int32_t)sc->num_tris) return fail(HPT_ERR_INVALID_ARGUMENT, "%s %d: triangle %d out of range", name, i, tri);
            if (tri >= 0) {
                if (miss != i + 1) return fail(HPT_ERR_INVALID_ARGUMENT, "%s %d (a leaf): miss link must be the next node", name, i);
            } else {
                // interior: left child = i+1, right child = miss link of the left child
                if (i + 1 >= nn) return fail(HPT_ERR_INVALID_ARGUMENT, "interior %s %d has no children", name, i);
                int32_t lmiss;
                std::memcpy(&lmiss, &nodes[8 * (size_t)(i + 1) + 3], 4);
                if (lmiss <= i + 1 || lmiss >= nn || lmiss >= miss)
                    return fail(HPT_ERR_INVALID_ARGUMENT, "interior %s %d: right child %d outside its subtree", name, i, lmiss);
            }
        }
        return HPT_OK;
    };
    if (int rc = check_tree(sc->nodes, sc->num_nodes, "node")) return rc;
    if (sc->num_ref_nodes || sc->ref_nodes) {
        if (!sc->ref_nodes || sc->num_ref_nodes == 0 || sc->num_ref_nodes > (1u << 30)) return fail(HPT_ERR_INVALID_ARGUMENT, "ref_nodes and num_ref_nodes must be given together");
        if (int rc = check_tree(sc->ref_nodes, sc->num_ref_nodes, "ref_node")) return rc;
        // both trees stand over the same leaves, so their roots' boxes are the same min/max of the same numbers
        if (std::memcmp(sc->nodes, sc->ref_nodes, 12) != 0 || std::memcmp(sc->nodes + 4, sc->ref_nodes + 4, 12) != 0)
            return fail(HPT_ERR_INVALID_ARGUMENT, "nodes and ref_nodes have different root boxes: they are not trees over the same leaves");
    }
    for (uint32_t t = 0; t < sc->num_tris; ++t) {
        int32_t mat;
        std::memcpy(&mat, &sc->tris[16 * (size_t)t + 7], 4);
        if (mat < 0 || mat >= (int32_t)sc->num_materials) return fail(HPT_ERR_INVALID_ARGUMENT, "triangle %u: material %d out of range", t, mat);
    }
    for (uint32_t l = 0; l < sc->num_lights; ++l) {
        if (sc->lights[l].root_node >= sc->num_nodes || sc->lights[l].material >= sc->num_materials)
            return fail(HPT_ERR_INVALID_ARGUMENT, "light %u: bad node/material index", l);
    }
    return HPT_OK;
}

// A box of an octant copy's tree while it is built: a leaf (tri >= 0; bits = the triangles it stands for) or an
// interior node over `kids` (indices into the same vector).
struct OctBox {
    float lo[3], hi[3];
    int32_t tri;
    uint32_t bits;
    std::vector<int> kids;
};

// Tunes the trees of the eight octant copies for what a WAVE pays.  The lock-step kernels walk until the slowest of
// a wave's 64 lanes is done, so a box walk costs the largest step count among 64 rays, not the mean the area rule
// minimises -- and that rule prices a box by the chance that a random LINE through the scene crosses it, while the
// rays of a render start on the scene's surfaces.  So: draw rays the way an integrator makes them (origins uniform
// over the triangles' area, directions cosine-distributed about the normal; shadow rays from the same points to
// uniform points of the emitting triangles), count for every ray the records its octant's tree tests, and change the
// trees greedily -- dissolve an interior node into its parent's list, put two siblings under a new node, move a
// sibling into a sibling node -- as long as E[max of 64 draws from the step distribution], closest and shadow queries
// added, goes down.  With nested boxes a ray that hits a node hits its ancestors, so the steps of a ray are the length
// of the top list plus, for every interior node it hits, that node's number of children: one pass over the interior
// nodes per candidate.  Whatever comes out is a tree of unions over the same leaves, i.e. as exact as any other
// (build_octant_copies).  Cornell box: 15.3 + 12.6 wave steps per bounce with the area rule, 14.5 + 11.3 after.
void tune_octant_trees(const hpt_scene_desc* sc, std::vector<OctBox> (&pools)[8], std::vector<int> (&tops)[8]) {
    constexpr int kRays = 8192, kWave = 64, kMaxSteps = 160, kMaxEvaluations = 3000;
    constexpr double kMeanWeight = 0.05, kMinGain = 1e-3;
    const uint32_t nt = sc->num_tris;
    // ---- rays ----
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto uniform = [&]() {
        rng += 0x9E3779B97F4A7C15ull;
        uint64_t z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    };
    std::vector<double> cdf(nt, 0.0), light_cdf(nt, 0.0);
    double total = 0.0, light_total = 0.0;
    for (uint32_t t = 0; t < nt; ++t) {
        const float* r = &sc->tris[16 * (size_t)t];
        const double e1[3] = {r[4], r[5], r[6]}, e2[3] = {r[8], r[9], r[10]};
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        double a = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        if (!(a > 0.0) || !std::isfinite(a)) a = 0.0;
        uint32_t m;
        std::memcpy(&m, &r[7], 4);
        const bool emits = m < sc->num_materials && (sc->materials[m].emission[0] > 0.f || sc->materials[m].emission[1] > 0.f || sc->materials[m].emission[2] > 0.f);
        total += a;
        cdf[t] = total;
        if (emits) light_total += a;
        light_cdf[t] = light_total;
    }
    if (!(total > 0.0)) return;
    auto pick = [&](const std::vector<double>& c, double upto) {
        const double x = uniform() * upto;
        return (uint32_t)std::min<size_t>(nt - 1, (size_t)(std::upper_bound(c.begin(), c.end(), x) - c.begin()));
    };
    auto point_on = [&](uint32_t t, double* p) {
        const float* r = &sc->tris[16 * (size_t)t];
        const double su = std::sqrt(uniform()), v = uniform();
        for (int c = 0; c < 3; ++c) p[c] = r[c] + su * (1.0 - v) * r[4 + c] + su * v * r[8 + c];
    };
    struct RaySet {
        std::vector<float> o, inv;  // 3 per ray
        size_t size() const { return o.size() / 3; }
    };
    RaySet rays[8][2];  // [octant][closest | shadow]
    auto add_ray = [&](int kind, const double* p, const double* d) {
        float o[3], inv[3];
        uint32_t k = 0;
        for (int c = 0; c < 3; ++c) {
            const float dc = (float)d[c];
            o[c] = (float)p[c];
            inv[c] = 1.0f / dc;
            if (dc == 0.f || !std::isfinite(inv[c])) return;
            if (dc < 0.f) k |= 1u << c;
        }
        rays[k][kind].o.insert(rays[k][kind].o.end(), o, o + 3);
        rays[k][kind].inv.insert(rays[k][kind].inv.end(), inv, inv + 3);
    };
    for (int i = 0; i < kRays; ++i) {
        const uint32_t t = pick(cdf, total);
        const float* r = &sc->tris[16 * (size_t)t];
        double p[3];
        point_on(t, p);
        const double n[3] = {r[12], r[13], r[14]};
        const double r1 = uniform(), phi = 6.283185307179586 * uniform();
        const double x = std::sqrt(r1) * std::cos(phi), y = std::sqrt(r1) * std::sin(phi), z = std::sqrt(1.0 - r1);
        const double h[3] = {std::fabs(n[0]) > 0.9 ? 0.0 : 1.0, std::fabs(n[0]) > 0.9 ? 1.0 : 0.0, 0.0};
        double tx[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        const double tl = std::sqrt(tx[0] * tx[0] + tx[1] * tx[1] + tx[2] * tx[2]);
        if (!(tl > 0.0)) continue;
        for (double& c : tx) c /= tl;
        const double b[3] = {n[1] * tx[2] - n[2] * tx[1], n[2] * tx[0] - n[0] * tx[2], n[0] * tx[1] - n[1] * tx[0]};
        const double d[3] = {tx[0] * x + b[0] * y + n[0] * z, tx[1] * x + b[1] * y + n[1] * z, tx[2] * x + b[2] * y + n[2] * z};
        add_ray(0, p, d);
        if (light_total > 0.0) {
            double q[3];
            point_on(pick(light_cdf, light_total), q);
            double s[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
            const double sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            if (sl > 0.0 && s[0] * n[0] + s[1] * n[1] + s[2] * n[2] > 0.0) {
                for (double& c : s) c /= sl;
                add_ray(1, p, s);
            }
        }
    }
    // ---- trees as nested values over leaf numbers (bit i of a mask = leaf i of the octant) ----
    struct Tree {
        int leaf = -1;  // >= 0: leaf number; -1: interior (or the top list)
        uint64_t mask = 0;
        std::vector<Tree> kids;
    };
    struct Octant {
        std::vector<int> leaf_box;  // leaf number -> index in pools[k]
        Tree top;
        std::vector<std::pair<uint64_t, std::vector<uint8_t>>> hits[2];  // per kind: leaf set -> hit flag per ray
        std::vector<int> hist[2];
    };
    Octant oc[8];
    for (uint32_t k = 0; k < 8; ++k) {
        const std::vector<OctBox>& pool = pools[k];
        std::vector<int> leaf_of(pool.size(), -1);
        for (size_t j = 0; j < pool.size(); ++j)
            if (pool[j].tri >= 0) {
                leaf_of[j] = (int)oc[k].leaf_box.size();
                oc[k].leaf_box.push_back((int)j);
            }
        if (oc[k].leaf_box.size() > 64) return;
        std::function<Tree(int)> import = [&](int n) {
            Tree t;
            if (pool[n].tri >= 0) {
                t.leaf = leaf_of[n];
                t.mask = 1ull << t.leaf;
            } else {
                for (int c : pool[n].kids) {
                    t.kids.push_back(import(c));
                    t.mask |= t.kids.back().mask;
                }
            }
            return t;
        };
        for (int n : tops[k]) {
            oc[k].top.kids.push_back(import(n));
            oc[k].top.mask |= oc[k].top.kids.back().mask;
        }
    }
    auto hit_flags = [&](uint32_t k, int kind, uint64_t mask) -> const std::vector<uint8_t>& {
        for (auto& e : oc[k].hits[kind])
            if (e.first == mask) return e.second;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t l = 0; l < oc[k].leaf_box.size(); ++l)
            if (mask >> l & 1ull)
                for (int c = 0; c < 3; ++c) {
                    lo[c] = std::min(lo[c], pools[k][oc[k].leaf_box[l]].lo[c]);
                    hi[c] = std::max(hi[c], pools[k][oc[k].leaf_box[l]].hi[c]);
                }
        const RaySet& rs = rays[k][kind];
        std::vector<uint8_t> f(rs.size());
        for (size_t i = 0; i < rs.size(); ++i) {
            float tn = -INFINITY, tf = INFINITY;
            for (int c = 0; c < 3; ++c) {
                const float a = (lo[c] - rs.o[3 * i + c]) * rs.inv[3 * i + c], b = (hi[c] - rs.o[3 * i + c]) * rs.inv[3 * i + c];
                tn = std::max(tn, std::min(a, b));
                tf = std::min(tf, std::max(a, b));
            }
            f[i] = tn <= tf;
        }
        oc[k].hits[kind].emplace_back(mask, std::move(f));
        return oc[k].hits[kind].back().second;
    };
    auto histogram = [&](uint32_t k, int kind, const Tree& top) {
        const size_t n = rays[k][kind].size();
        std::vector<uint16_t> steps(n, (uint16_t)top.kids.size());
        std::function<void(const Tree&)> add = [&](const Tree& t) {
            for (const Tree& c : t.kids)
                if (c.leaf < 0) {
                    const std::vector<uint8_t>& f = hit_flags(k, kind, c.mask);  // (used up before the recursion below can grow the cache)
                    const uint16_t w = (uint16_t)c.kids.size();
                    for (size_t i = 0; i < n; ++i) steps[i] = (uint16_t)(steps[i] + (f[i] ? w : 0));
                    add(c);
                }
        };
        add(top);
        std::vector<int> h(kMaxSteps, 0);
        for (size_t i = 0; i < n; ++i) ++h[std::min<int>(steps[i], kMaxSteps - 1)];
        return h;
    };
    auto objective = [&](uint32_t changed, const std::vector<int>* replacement) {
        double value = 0.0;
        for (int kind = 0; kind < 2; ++kind) {
            double count = 0.0, sum = 0.0, pooled[kMaxSteps];
            for (int s = 0; s < kMaxSteps; ++s) {
                pooled[s] = 0.0;
                for (uint32_t k = 0; k < 8; ++k) pooled[s] += (k == changed && replacement) ? replacement[kind][s] : oc[k].hist[kind][s];
                count += pooled[s];
                sum += pooled[s] * s;
            }
            if (!(count > 0.0)) continue;
            double below = 0.0, wave = 0.0;
            for (int s = 0; s < kMaxSteps; ++s) {
                below += pooled[s];
                wave += 1.0 - std::pow(below / count, (double)kWave);
            }
            value += wave + kMeanWeight * sum / count;
        }
        return value;
    };
    for (uint32_t k = 0; k < 8; ++k)
        for (int kind = 0; kind < 2; ++kind) oc[k].hist[kind] = histogram(k, kind, oc[k].top);
    double cost = objective(8, nullptr);
    // ---- greedy search, first improvement ----
    auto at_path = [](Tree& root, const std::vector<int>& path) -> Tree& {
        Tree* t = &root;
        for (int i : path) t = &t->kids[(size_t)i];
        return *t;
    };
    auto remask = [](Tree& t) {
        t.mask = 0;
        for (const Tree& c : t.kids) t.mask |= c.mask;
    };
    int evaluations = 0;
    bool improved = true;
    while (improved && evaluations < kMaxEvaluations) {
        improved = false;
        for (uint32_t k = 0; k < 8; ++k) {
            if (rays[k][0].size() + rays[k][1].size() == 0) continue;
            bool again = true;
            while (again && evaluations < kMaxEvaluations) {
                again = false;
                // the lists of the tree (the top list and every interior node), by path
                std::vector<std::vector<int>> lists;
                std::function<void(const Tree&, std::vector<int>&)> collect = [&](const Tree& t, std::vector<int>& path) {
                    lists.push_back(path);
                    for (size_t i = 0; i < t.kids.size(); ++i)
                        if (t.kids[i].leaf < 0) {
                            path.push_back((int)i);
                            collect(t.kids[i], path);
                            path.pop_back();
                        }
                };
                std::vector<int> path;
                collect(oc[k].top, path);
                auto try_tree = [&](const Tree& cand) {
                    ++evaluations;
                    std::vector<int> h[2] = {histogram(k, 0, cand), histogram(k, 1, cand)};
                    const double v = objective(k, h);
                    if (v < cost - kMinGain) {
                        cost = v;
                        oc[k].top = cand;
                        oc[k].hist[0] = h[0];
                        oc[k].hist[1] = h[1];
                        return true;
                    }
                    return false;
                };
                for (size_t li = 0; li < lists.size() && !again; ++li) {
                    const Tree& lst = at_path(oc[k].top, lists[li]);
                    const size_t n = lst.kids.size();
                    for (size_t i = 0; i < n && !again; ++i)  // dissolve child i
                        if (lst.kids[i].leaf < 0) {
                            Tree cand = oc[k].top;
                            Tree& l = at_path(cand, lists[li]);
                            std::vector<Tree> moved = std::move(l.kids[i].kids);
                            l.kids.erase(l.kids.begin() + (long)i);
                            l.kids.insert(l.kids.begin() + (long)i, moved.begin(), moved.end());
                            again = try_tree(cand);
                        }
                    if (n > 2) {
                        for (size_t i = 0; i < n && !again; ++i)  // children i and j under a new node
                            for (size_t j = i + 1; j < n && !again; ++j) {
                                Tree cand = oc[k].top;
                                Tree& l = at_path(cand, lists[li]);
                                Tree u;
                                u.kids = {l.kids[i], l.kids[j]};
                                remask(u);
                                l.kids.erase(l.kids.begin() + (long)j);
                                l.kids.erase(l.kids.begin() + (long)i);
                                l.kids.push_back(u);
                                again = try_tree(cand);
                            }
                        for (size_t i = 0; i < n && !again; ++i)  // child i into the interior sibling j
                            for (size_t j = 0; j < n && !again; ++j)
                                if (i != j && lst.kids[j].leaf < 0) {
                                    Tree cand = oc[k].top;
                                    Tree& l = at_path(cand, lists[li]);
                                    l.kids[j].kids.push_back(l.kids[i]);
                                    remask(l.kids[j]);
                                    l.kids.erase(l.kids.begin() + (long)i);
                                    again = try_tree(cand);
                                }
                    }
                }
                improved = improved || again;
            }
        }
    }
    // ---- back to boxes ----
    for (uint32_t k = 0; k < 8; ++k) {
        std::vector<OctBox> pool;
        for (int j : oc[k].leaf_box) pool.push_back(pools[k][(size_t)j]);  // leaf number = index
        std::function<int(const Tree&)> emit = [&](const Tree& t) -> int {
            if (t.leaf >= 0) return t.leaf;
            OctBox u;
            u.tri = -1;
            u.bits = 0u;
            for (const Tree& c : t.kids) u.kids.push_back(emit(c));
            for (int c = 0; c < 3; ++c) {
                u.lo[c] = pool[(size_t)u.kids[0]].lo[c];
                u.hi[c] = pool[(size_t)u.kids[0]].hi[c];
                for (int kid : u.kids) {
                    u.lo[c] = std::min(u.lo[c], pool[(size_t)kid].lo[c]);
                    u.hi[c] = std::max(u.hi[c], pool[(size_t)kid].hi[c]);
                }
            }
            pool.push_back(u);
            return (int)pool.size() - 1;
        };
        std::vector<int> top;
        for (const Tree& c : oc[k].top.kids) top.push_back(emit(c));
        pools[k] = std::move(pool);
        tops[k] = std::move(top);
    }
}

// The nine node-array copies of the marking variants (layout: stage_scene in hpt_device.h).
// out_nodes: 9 * nn records of 8 floats; out_table: 32 dwords.
//
// A triangle with normal n is rejected by Triangle::getIntersection's first test,
// dot(d, n) > 0 (common/Triangle.hpp:168-169), for EVERY ray of a sign octant (s_i = sign of
// d_i, no d_i zero) when n_i * s_i >= 0 for all i and n_i * s_i >= 0.5 for one i: no product
// d_i * n_i is negative and that one is at least 0.5 * |d_i| > 0 (|d_i| >= 2^-128 for a finite
// reciprocal, so the product does not underflow to zero).  It passes the test for every ray of
// the octant when n_i * s_i <= 0 for all i.  Subtrees that hold only always-rejected triangles
// are left out of the octant's copy, and an interior node with one such child is replaced by
// its other child when that child's box lies inside its own: then the child's slab test can
// only succeed where the parent's would (float subtraction and multiplication are monotone),
// so the set of triangles reached is unchanged.
//
// The same monotonicity goes further.  When every node's box lies inside its parent's (true of
// any tree whose interior boxes are min/max unions, as BVHAccel::recursiveBuild makes them,
// common/BVH.cpp:66-118), a ray with finite reciprocals that hits a leaf's box hits all of the
// leaf's ancestors, so the triangles the reference tests are exactly those whose LEAF box is
// hit -- the interior nodes decide nothing, they only save tests.  The octant copies are
// therefore free to use a tree of their own over the octant's live leaves, as long as each of
// its boxes is again the union of the leaf boxes below it.  `nested` scenes get one built for
// few box tests: leaves are merged bottom-up, always the pair with the smallest union surface
// area (the chance that a line crossing the scene crosses a box is proportional to its area),
// and an interior node is then dropped again when testing it costs more than it saves: with q
// = area / scene area, its test is made with probability q(parent) and spares its k children's
// tests with probability q(parent) - q(node).  The Cornell box goes from 17.6 to 10.2 box tests
// per ray.  Copy 8 (rays with a zero direction component, for which the argument does not
// hold: 0 * inf = NaN) always keeps the reference's tree.
void build_octant_copies(const hpt_scene_desc* sc_in, bool leaf_bits, std::vector<float>& out_nodes, std::vector<uint32_t>& out_table) {
    const uint32_t nn = sc_in->num_nodes, nt = sc_in->num_tris;
    out_nodes.assign((size_t)9 * nn * 8, 0.f);
    out_table.assign(32, 0u);
    // copy 8 (rays with a zero direction component) is the REFERENCE's tree: `ref_nodes` when the scene brings one
    // (then `nodes` is some other tree over the same leaves; the caller has checked that both have nn nodes)
    hpt_scene_desc whole_desc = *sc_in;
    if (sc_in->ref_nodes) whole_desc.nodes = sc_in->ref_nodes;
    const hpt_scene_desc* sc = sc_in;
    auto node_int = [&](uint32_t i, int word) {
        int32_t v;
        std::memcpy(&v, &sc->nodes[8 * (size_t)i + word], 4);
        return v;
    };
    struct Rec {
        float lo[3], hi[3];
        uint32_t miss;  // index of the miss target inside the copy
        int32_t tri;
        uint32_t bits;  // leaf_bits: the leaf's triangles in the pending mask
    };
    // every node's box inside its parent's (NaN bounds compare false: not nested)
    bool nested = true;
    for (uint32_t i = 0; i < nn && nested; ++i) {
        if (node_int(i, 7) >= 0) continue;
        const uint32_t kids[2] = {i + 1, (uint32_t)node_int(i + 1, 3)};
        for (uint32_t ch : kids)
            for (int c = 0; c < 3; ++c)
                if (!(sc->nodes[8 * (size_t)ch + c] >= sc->nodes[8 * (size_t)i + c] && sc->nodes[8 * (size_t)ch + 4 + c] <= sc->nodes[8 * (size_t)i + 4 + c]))
                    nested = false;
    }
    if (const char* e = getenv("HPT_OCTANT_TREES")) nested = nested && atoi(e) != 0;  // 0: keep the reference's topology (A/B runs)
    // the trees of the eight octant copies of a nested scene: by the area rule first, then tuned on sampled rays
    std::vector<OctBox> pools[8];
    std::vector<int> tops[8];
    auto rejected_in = [&](uint32_t k, uint32_t t) {
        const float sgn[3] = {(k & 1u) ? -1.f : 1.f, (k & 2u) ? -1.f : 1.f, (k & 4u) ? -1.f : 1.f};
        const float* n = &sc->tris[16 * (size_t)t + 12];
        const float a = n[0] * sgn[0], b = n[1] * sgn[1], c = n[2] * sgn[2];
        return a >= 0.f && b >= 0.f && c >= 0.f && (a >= 0.5f || b >= 0.5f || c >= 0.5f);
    };
    if (nested) {
        for (uint32_t k = 0; k < 8; ++k) {
            std::vector<OctBox>& pool = pools[k];
            std::vector<int>& top = tops[k];
            std::vector<uint8_t> rejected(nt, 0);
            for (uint32_t t = 0; t < nt; ++t) rejected[t] = rejected_in(k, t);
            std::vector<int> roots;
            for (uint32_t i = 0; i < nn; ++i) {
                const int32_t tri = node_int(i, 7);
                if (tri < 0 || rejected[tri]) continue;
                OctBox b;
                for (int c = 0; c < 3; ++c) {
                    b.lo[c] = sc->nodes[8 * (size_t)i + c];
                    b.hi[c] = sc->nodes[8 * (size_t)i + 4 + c];
                }
                b.tri = tri;
                b.bits = leaf_bits ? 1u << tri : 0u;
                // leaves with the same box pass or fail together: with a mask as the leaf word one
                // record serves them all (the two triangles of a rectangle have the same bounds)
                bool merged = false;
                if (leaf_bits)
                    for (int r : roots)
                        if (std::memcmp(pool[r].lo, b.lo, sizeof b.lo) == 0 && std::memcmp(pool[r].hi, b.hi, sizeof b.hi) == 0) {
                            pool[r].bits |= b.bits;
                            merged = true;
                            break;
                        }
                if (merged) continue;
                roots.push_back((int)pool.size());
                pool.push_back(b);
            }
            auto area = [](const float* lo, const float* hi) {
                const double dx = std::max(0.0, (double)hi[0] - lo[0]), dy = std::max(0.0, (double)hi[1] - lo[1]), dz = std::max(0.0, (double)hi[2] - lo[2]);
                return 2.0 * (dx * dy + dy * dz + dx * dz);
            };
            while (roots.size() > 1) {
                size_t bx = 0, by = 1;
                double best = -1.0;
                for (size_t x = 0; x < roots.size(); ++x)
                    for (size_t y = x + 1; y < roots.size(); ++y) {
                        float lo[3], hi[3];
                        for (int c = 0; c < 3; ++c) {
                            lo[c] = std::min(pool[roots[x]].lo[c], pool[roots[y]].lo[c]);
                            hi[c] = std::max(pool[roots[x]].hi[c], pool[roots[y]].hi[c]);
                        }
                        const double a = area(lo, hi);
                        if (best < 0.0 || a < best) {
                            best = a;
                            bx = x;
                            by = y;
                        }
                    }
                OctBox u;
                for (int c = 0; c < 3; ++c) {
                    u.lo[c] = std::min(pool[roots[bx]].lo[c], pool[roots[by]].lo[c]);
                    u.hi[c] = std::max(pool[roots[bx]].hi[c], pool[roots[by]].hi[c]);
                }
                u.tri = -1;
                u.bits = 0u;
                u.kids = {roots[bx], roots[by]};
                roots.erase(roots.begin() + (long)by);
                roots.erase(roots.begin() + (long)bx);
                roots.push_back((int)pool.size());
                pool.push_back(u);
            }
            const double scene_area = area(&sc->nodes[0], &sc->nodes[4]);
            auto q_of = [&](int n) { return scene_area > 0.0 ? std::min(1.0, area(pool[n].lo, pool[n].hi) / scene_area) : 1.0; };
            // children lists after dropping the interior nodes that do not pay
            std::function<void(int, double, std::vector<int>&)> keep = [&](int n, double q_parent, std::vector<int>& out) {
                if (pool[n].tri >= 0) {
                    out.push_back(n);
                    return;
                }
                const double q = q_of(n);
                std::vector<int> kids;
                for (int c : std::vector<int>(pool[n].kids)) keep(c, q, kids);
                if (q > q_parent * (1.0 - 1.0 / (double)kids.size())) {
                    out.insert(out.end(), kids.begin(), kids.end());
                } else {
                    pool[n].kids = kids;
                    out.push_back(n);
                }
            };
            if (!roots.empty()) keep(roots[0], 1.0, top);
        }
        bool tune = true;
        if (const char* e = getenv("HPT_OCTANT_TUNE")) tune = atoi(e) != 0;  // 0: the area rule alone (A/B runs)
        if (tune) tune_octant_trees(sc, pools, tops);
    }
    for (uint32_t k = 0; k < 9; ++k) {
        const bool whole = k == 8;
        if (whole) sc = &whole_desc;
        const float sgn[3] = {(k & 1u) ? -1.f : 1.f, (k & 2u) ? -1.f : 1.f, (k & 4u) ? -1.f : 1.f};
        std::vector<uint8_t> rejected(nt, 0), passes(nt, 0);
        if (!whole)
            for (uint32_t t = 0; t < nt; ++t) {
                const float* n = &sc->tris[16 * (size_t)t + 12];
                const float a = n[0] * sgn[0], b = n[1] * sgn[1], c = n[2] * sgn[2];
                rejected[t] = a >= 0.f && b >= 0.f && c >= 0.f && (a >= 0.5f || b >= 0.5f || c >= 0.5f);
                passes[t] = a <= 0.f && b <= 0.f && c <= 0.f;
            }
        // live[i]: the subtree of node i holds a triangle that is not always rejected
        std::vector<uint8_t> live(nn, 0);
        for (uint32_t i = nn; i-- > 0;) {
            const int32_t tri = node_int(i, 7);
            if (tri >= 0) live[i] = !rejected[tri];
            else live[i] = live[i + 1] || live[(uint32_t)node_int(i + 1, 3)];
        }
        auto inside = [&](uint32_t child, uint32_t parent) {
            for (int c = 0; c < 3; ++c)
                if (!(sc->nodes[8 * (size_t)child + c] >= sc->nodes[8 * (size_t)parent + c] &&
                      sc->nodes[8 * (size_t)child + 4 + c] <= sc->nodes[8 * (size_t)parent + 4 + c]))
                    return false;
            return true;
        };
        std::vector<Rec> recs;
        if (!whole && nested) {
            const std::vector<OctBox>& pool = pools[k];
            const std::vector<int>& top = tops[k];
            std::function<void(int)> emit = [&](int n) {
                Rec rc;
                for (int c = 0; c < 3; ++c) {
                    rc.lo[c] = pool[n].lo[c];
                    rc.hi[c] = pool[n].hi[c];
                }
                rc.tri = pool[n].tri;
                rc.bits = pool[n].bits;
                rc.miss = 0;
                const size_t at = recs.size();
                recs.push_back(rc);
                if (rc.tri < 0)
                    for (int c : pool[n].kids) emit(c);
                recs[at].miss = (uint32_t)recs.size();
            };
            for (int n : top) emit(n);
        }
        const bool rebuilt = !whole && nested && recs.size() <= nn;
        if (!rebuilt) recs.clear();
        // explicit stack of (node, index of the record to patch when the subtree is complete | ~0u)
        struct Item {
            uint32_t node;
            bool close;
            uint32_t rec;
        };
        std::vector<Item> stack;
        if (!rebuilt) stack.push_back({0u, false, 0u});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.close) {
                recs[it.rec].miss = (uint32_t)recs.size();
                continue;
            }
            uint32_t i = it.node;
            if (!live[i]) continue;
            // skip interior nodes that only lead to one live child lying inside them
            for (;;) {
                if (node_int(i, 7) >= 0) break;
                const uint32_t l = i + 1, r = (uint32_t)node_int(l, 3);
                if (live[l] && live[r]) break;
                const uint32_t only = live[l] ? l : r;
                if (!inside(only, i)) break;
                i = only;
            }
            Rec rc;
            for (int c = 0; c < 3; ++c) {
                rc.lo[c] = sc->nodes[8 * (size_t)i + c];
                rc.hi[c] = sc->nodes[8 * (size_t)i + 4 + c];
            }
            rc.tri = node_int(i, 7);
            rc.bits = rc.tri >= 0 && rc.tri < 32 ? 1u << rc.tri : 0u;
            rc.miss = (uint32_t)recs.size() + 1u;  // leaf: the next record
            recs.push_back(rc);
            if (rc.tri < 0) {
                const uint32_t l = i + 1, r = (uint32_t)node_int(l, 3);
                stack.push_back({0u, true, (uint32_t)recs.size() - 1u});
                if (live[r]) stack.push_back({r, false, 0u});
                if (live[l]) stack.push_back({l, false, 0u});
            }
        }
        out_table[k] = (uint32_t)recs.size();
        for (size_t j = 0; j < recs.size(); ++j) {
            float* q = &out_nodes[((size_t)k * nn + j) * 8];
            for (int c = 0; c < 3; ++c) {
                const bool neg = !whole && (k >> c) & 1u;
                q[c] = neg ? recs[j].hi[c] : recs[j].lo[c];
                q[4 + c] = neg ? recs[j].lo[c] : recs[j].hi[c];
            }
            const uint32_t link = (k * nn + recs[j].miss) * 32u;
            std::memcpy(&q[3], &link, 4);
            uint32_t leaf = (uint32_t)recs[j].tri;
            if (leaf_bits) leaf = recs[j].bits;
            std::memcpy(&q[7], &leaf, 4);
        }
        if (!whole) {
            unsigned long long undecided = 0;
            for (uint32_t t = 0; t < nt && t < 64; ++t)
                if (!rejected[t] && !passes[t]) undecided |= 1ull << t;
            out_table[16 + 2 * k] = (uint32_t)undecided;
            out_table[17 + 2 * k] = (uint32_t)(undecided >> 32);
        }
    }
}

// Timing events: a render takes a pair from the free list, records it around its launches and
// parks it in `pending_events`; pairs whose second event has completed are read and recycled by
// the next hpt_render (no host wait), so a loop of renders that never reads the statistics holds
// a bounded number of events.
int take_event(hpt_ctx* ctx, hipEvent_t* out) {
    if (!ctx->free_events.empty()) {
        *out = ctx->free_events.back();
        ctx->free_events.pop_back();
        return HPT_OK;
    }
    HIP_TRY(hipEventCreate(out));
    return HPT_OK;
}

constexpr size_t kMaxPendingEvents = 64;  // pairs; beyond that hpt_render waits for the oldest

// settle the pairs at the front of the list that have completed (all of them with wait = true)
int settle_events(hpt_ctx* ctx, bool wait) {
    size_t done = 0;
    for (auto& pr : ctx->pending_events) {
        if (wait || ctx->pending_events.size() - done > kMaxPendingEvents) {
            HIP_TRY(hipEventSynchronize(pr.second));
        } else {
            const hipError_t q = hipEventQuery(pr.second);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return fail(HPT_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->kernel_ms += ms;
        ctx->free_events.push_back(pr.first);
        ctx->free_events.push_back(pr.second);
        ++done;
    }
    ctx->pending_events.erase(ctx->pending_events.begin(), ctx->pending_events.begin() + (long)done);
    return HPT_OK;
}

// dynamic LDS of the traversal kernels: the staged scene (if any) + the per-lane
// triangle queues of the variants that use them
uint32_t lds_bytes(const DevScene& S, int block) {
    const bool pair = S.variant == kScenePair || S.variant == kScenePairLds || S.variant == kScenePairCull;
    const bool queue = pair || S.variant == kSceneGlobal || S.variant == kSceneLds;
    const uint32_t staging = coop_variant(S.variant) ? (((uint32_t)block + 63u) >> 6) * kCoopWaveBytes : 0u;  // cooperative record fetch
    return S.lds_float4s * 16u + staging + (queue ? (uint32_t)block * tri_queue_entries(S.variant) * 4u : 0u) + (pair ? (uint32_t)block * S.stack_depth * 4u : 0u);
}

int grid_for(const hpt_ctx* ctx, uint64_t work, int block) {
    // grid-stride kernels: twice the workgroups that are resident at full
    // occupancy (2048 lanes per CU), never more than the work needs
    const uint64_t want = (work + (uint64_t)block - 1) / (uint64_t)block;
    const uint64_t per_cu = 2048ull / (uint64_t)(block < 64 ? 64 : block);
    const uint64_t cap = (uint64_t)ctx->num_cus * (per_cu ? per_cu : 1ull) * 2ull;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

// RCCL, bound at the first cross-device reduce (the library has no link-time dependency on it).
// Only the few entry points hpt_accum_reduce uses; types as in rccl.h (ncclInt64 = 4, ncclSum = 0).
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return HPT_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return fail(HPT_ERR_HIP, "cannot load RCCL (librccl.so): %s", dlerror());
    Rccl r;
    r.lib = lib;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(lib, "ncclReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Reduce || !r.GetErrorString) {
        dlclose(lib);
        return fail(HPT_ERR_HIP, "librccl.so lacks an entry point hpt_accum_reduce needs");
    }
    g_rccl = r;
    return HPT_OK;
}
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        const int r_ = (expr);                                                                                \
        if (r_ != 0) return fail(HPT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));             \
    } while (0)

// Communicators are kept for the life of the process's contexts: one set per ordered list of devices,
// created by the first reduce over that list (or ahead of it by hpt_accum_reduce_prepare) and destroyed
// when the last context of the process is destroyed.  Creating a communicator costs tens to hundreds of
// milliseconds (bootstrap, topology search, buffer registration), a reduce of 25 MB over xGMI a fraction
// of one: a reduce that created its communicator every time would be communicator creation and little else.
struct CommSet {
    std::vector<int> devs;
    std::vector<void*> comms;
};
std::mutex g_comm_mutex;
std::vector<CommSet> g_comm_sets;
unsigned long long g_comm_created = 0;  // communicator sets created since the process started
int g_live_contexts = 0;
double g_last_reduce_ms = 0.0;

// the set for `devs`, created if the process has none yet (g_comm_mutex held by the caller)
int comm_set_for(const std::vector<int>& devs, const CommSet** out) {
    for (const CommSet& s : g_comm_sets)
        if (s.devs == devs) {
            *out = &s;
            return HPT_OK;
        }
    if (int rc = load_rccl()) return rc;
    CommSet s;
    s.devs = devs;
    s.comms.assign(devs.size(), nullptr);
    RCCL_TRY(g_rccl.CommInitAll(s.comms.data(), (int)devs.size(), devs.data()));
    ++g_comm_created;
    g_comm_sets.push_back(std::move(s));
    *out = &g_comm_sets.back();
    return HPT_OK;
}

void comm_sets_destroy_all() {
    for (CommSet& s : g_comm_sets)
        for (void* c : s.comms)
            if (c) (void)g_rccl.CommDestroy(c);
    g_comm_sets.clear();
}

// Buffers a render needs beyond its inputs and outputs.  nullptr for `fixed` in render_launch =
// the context's own (the wavefront queues grow on demand); otherwise exactly these are used and
// nothing is allocated.
struct RenderBuffers {
    uint32_t* unit_counter;  // work-unit counter of the single-kernel mode
    float4* wf[8];           // two wavefront queues of four planes each
    uint64_t wf_paths;       // records each plane holds
    uint32_t* live;          // per-segment live counters
    size_t live_slots;
};

int check_render_args(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (mode != HPT_MODE_SINGLE_KERNEL && mode != HPT_MODE_STREAMED) return fail(HPT_ERR_INVALID_ARGUMENT, "mode must be 1 or 2, got %d", mode);
    if (block < 1 || block > 1024) return fail(HPT_ERR_INVALID_ARGUMENT, "block size %d not in [1, 1024]", block);
    if (max_depth < 0 || max_depth > 16384) return fail(HPT_ERR_INVALID_ARGUMENT, "max_depth %d not in [0, 16384]", max_depth);
    if ((uint64_t)sample_begin + sample_count > 0xFFFFFFFFull) return fail(HPT_ERR_INVALID_ARGUMENT, "sample range overflows 32 bits");
    return HPT_OK;
}

// the kernel launches of one render request (arguments already checked, device current)
int render_launch(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count,
                  int64_t* accum, hipStream_t stream, const RenderBuffers* fixed, uint32_t* launches_out) {
    uint32_t* const unit_counter = fixed ? fixed->unit_counter : ctx->d_unit_counter;
    const DevScene& S = ctx->S;
    const uint32_t tiles = (((uint32_t)S.width + 7u) >> 3) * (((uint32_t)S.height + 7u) >> 3);
    const uint64_t lanes = (uint64_t)tiles * 64ull;  // lanes per sample (edge tiles are partly idle)
    const uint32_t lds = lds_bytes(S, 0);  // scene only; queues are added per launch below

    // A workgroup stages the scene in LDS for itself (variants 1, 2, 3, 6: 20 KiB for the Cornell
    // box), so small workgroups leave most of a CU's wave slots empty (64 lanes: half the speed of
    // 256).  No result depends on the workgroup size, so such scenes are run with at least 256
    // lanes per workgroup; HPT_EXACT_BLOCK=1 keeps the size asked for.
    if (S.mat_lds_off > 0 && block < 256) {
        const char* e = getenv("HPT_EXACT_BLOCK");
        if (!(e && atoi(e) != 0)) block = 256;
    }

    uint32_t launches = 0;

    if (mode == HPT_MODE_SINGLE_KERNEL) {
        const uint32_t waves_per_block = ((uint32_t)block + 63u) >> 6;
        // HPT_SCHEDULER_VARIANTS: bit mask of the traversal variants that run under the scheduler in this mode.
        const char* sched_env = getenv("HPT_SCHEDULER");
        const char* sched_mask = getenv("HPT_MEGA_SCHEDULER_VARIANTS");
        const uint32_t mega_mask = sched_mask ? (uint32_t)strtoul(sched_mask, nullptr, 0) : kMegaSchedulerVariants;
        const bool scheduler = has_scheduler(S.variant) && !(sched_env && atoi(sched_env) == 0) && ((mega_mask >> S.variant) & 1u);
        // + queues; the lock-step kernel also keeps two tile accumulators per wave and its ring of camera directions there
        uint32_t lds_mega = lds + (scheduler ? 0u : waves_per_block * (384u * 8u + kCamRingWords * 4u)) + (lds_bytes(S, block) - lds);
#ifdef HPT_PROFILE
        lds_mega += waves_per_block * 3u * kProfSections * 8u;
#endif
        // per-lane ray counters are 16 bits wide and are emptied when a wave opens a unit; while a
        // pool is not empty the wave's 64 lanes all advance, so between two openings a wave runs at
        // most (c + 2) * max_depth loop iterations for units of c samples
        const uint32_t cap = 65535u / (uint32_t)max_depth;
        const uint32_t chunk = std::max(1u, std::min(ctx->spp_per_unit, cap > 2u ? cap - 2u : 1u));
        // at most 2^22 units per launch
        const uint32_t max_chunks = std::max(1u, (1u << 22) / tiles);
        // launches of about 2^30 paths (1024 spp at 1024x1024: 0.4 s on the Cornell box, about 2 s on the 1.0 M-triangle
        // scene at 2048x2048) amortise the launch's tail: 1 % faster than launches of 2^28 on the Cornell box
        const uint32_t want = ctx->spp_per_launch ? ctx->spp_per_launch : std::max(1u, (1u << 24) / tiles);
        const uint32_t per_launch = std::max(1u, std::min(want, chunk * max_chunks));
        // persistent waves: as many workgroups as are resident at once
        if (lds_mega > kMaxLdsPerBlock)
            return fail(HPT_ERR_INVALID_ARGUMENT, "a workgroup of %d lanes needs %u bytes of LDS for this scene (limit %u): use a smaller block", block, lds_mega, kMaxLdsPerBlock);
        int per_cu = 0;
        // HPT_SCHEDULER=0 (or, for the streamed mode, HPT_WF_SCHEDULER=0): the plain loops, for A/B runs.
#define HPT_MEGA_KERNEL(kS, kG, kB) (scheduler ? reinterpret_cast<const void*>(&k_render_mega_sm<has_scheduler(kS) ? kS : kScenePair, kG, kB>) \
                                               : reinterpret_cast<const void*>(&k_render_mega<kS, kG, kB>))
        HPT_DISPATCH_RENDER(S.variant, ctx->extended, {
            if (block <= 256) {
                if (lds_mega > 48u * 1024u) HIP_TRY(hipFuncSetAttribute(HPT_MEGA_KERNEL(kS, kG, 256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mega));
                if (scheduler) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_mega_sm<has_scheduler(kS) ? kS : kScenePair, kG, 256>, block, lds_mega));
                else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_mega<kS, kG, 256>, block, lds_mega));
            } else {
                if (lds_mega > 48u * 1024u) HIP_TRY(hipFuncSetAttribute(HPT_MEGA_KERNEL(kS, kG, 1024), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mega));
                if (scheduler) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_mega_sm<has_scheduler(kS) ? kS : kScenePair, kG, 1024>, block, lds_mega));
                else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_mega<kS, kG, 1024>, block, lds_mega));
            }
        })
#undef HPT_MEGA_KERNEL
        if (per_cu < 1) return fail(HPT_ERR_INVALID_ARGUMENT, "a workgroup of %d lanes with %u bytes of LDS does not fit a compute unit", block, lds_mega);
        const uint32_t resident = (uint32_t)per_cu * (uint32_t)ctx->num_cus;
        for (uint32_t s = sample_begin; s < sample_begin + sample_count;) {
            const uint32_t n = (sample_begin + sample_count - s > per_launch) ? per_launch : sample_begin + sample_count - s;
            const uint32_t units = tiles * ((n + chunk - 1u) / chunk);
            const uint32_t blocks = std::min(resident, (units + waves_per_block - 1u) / waves_per_block);
            HIP_TRY(hipMemsetAsync(unit_counter, 0, sizeof(uint32_t), stream));
            HPT_DISPATCH_RENDER(S.variant, ctx->extended, {
                constexpr int kT = has_scheduler(kS) ? kS : kScenePair;  // (never launched for variants without a Traversal<>)
                if (scheduler) {
                    if (block <= 256) hipLaunchKernelGGL((k_render_mega_sm<kT, kG, 256>), dim3(blocks), dim3(block), lds_mega, stream, S, max_depth, s, n, chunk, units, accum, ctx->d_stats, unit_counter);
                    else hipLaunchKernelGGL((k_render_mega_sm<kT, kG, 1024>), dim3(blocks), dim3(block), lds_mega, stream, S, max_depth, s, n, chunk, units, accum, ctx->d_stats, unit_counter);
                } else {
                    if (block <= 256) hipLaunchKernelGGL((k_render_mega<kS, kG, 256>), dim3(blocks), dim3(block), lds_mega, stream, S, max_depth, s, n, chunk, units, accum, ctx->d_stats, unit_counter);
                    else hipLaunchKernelGGL((k_render_mega<kS, kG, 1024>), dim3(blocks), dim3(block), lds_mega, stream, S, max_depth, s, n, chunk, units, accum, ctx->d_stats, unit_counter);
                }
            })
            HIP_TRY(hipGetLastError());
            ++launches;
            s += n;
        }
    } else {
        // the wavefront kernels are built for workgroups of up to 256 lanes
        const int wblock = block > 256 ? 256 : block;
        const uint64_t total = lanes * (uint64_t)sample_count;
        const uint64_t cap_limit = fixed ? fixed->wf_paths : ctx->wf_capacity;
        uint64_t cap = cap_limit < total ? cap_limit : ((total + 63ull) & ~63ull);
        // one queue segment per workgroup; exactly the workgroups that are resident at once
        int per_cu = 0;
        // scenes read through two-child records run the wave scheduler (HPT_WF_SCHEDULER=0: the plain loop, for A/B runs)
        const char* sched_env = getenv("HPT_SCHEDULER");
        const char* sched_env2 = getenv("HPT_WF_SCHEDULER");
        const char* sched_mask = getenv("HPT_WF_SCHEDULER_VARIANTS");
        const uint32_t wf_mask = sched_mask ? (uint32_t)strtoul(sched_mask, nullptr, 0) : kWfSchedulerVariants;
        const bool scheduler = has_scheduler(S.variant) && !(sched_env && atoi(sched_env) == 0) && !(sched_env2 && atoi(sched_env2) == 0) && ((wf_mask >> S.variant) & 1u);
        HPT_DISPATCH_RENDER(S.variant, ctx->extended, {
            if constexpr (has_scheduler(kS)) {
                if (scheduler) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_bounces_sm<kS, kG>, wblock, lds_bytes(S, wblock)));
                else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_bounces<kS, kG>, wblock, lds_bytes(S, wblock)));
            } else {
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_bounces<kS, kG>, wblock, lds_bytes(S, wblock)));
            }
        })
        if (per_cu < 1) return fail(HPT_ERR_INVALID_ARGUMENT, "the wavefront kernel does not fit a compute unit with %u bytes of LDS", lds_bytes(S, wblock));
        uint32_t segments = 0, seg = 0;
        float4* const* wf = ctx->d_wf;
        uint32_t* live = nullptr;
        for (;;) {
            const uint64_t cap_groups = (cap + 63ull) >> 6;
            // The lock-step kernels get 16 workgroups (segments) per resident slot: a segment's late bounces hold few
            // paths, and with one generation of workgroups every compute unit is in that thin tail at the same time;
            // with several, the tail of one workgroup runs beside the first bounces of the next (Cornell box, 2^28
            // paths per batch: 27.0 Grays/s with 1 generation, 28.7 / 30.2 / 31.5 / 31.9 / 31.6 / 30.7 with 2 / 4 / 8 /
            // 16 / 32 / 64).  Under the wave scheduler the lanes of a thin wave are refilled anyway and more
            // generations gain nothing (bunny +1 %, 1.0 M-triangle height field -2 % at 4).  HPT_WF_GENERATIONS overrides.
            uint64_t generations = scheduler ? 1u : 16u;
            if (const char* e = getenv("HPT_WF_GENERATIONS")) generations = (uint64_t)std::max(1, atoi(e));
            if (fixed) generations = 1;  // (the fixed buffers hold one live counter per resident workgroup)
            // A batch is not cut finer than 128 groups of 64 paths per segment (unless one generation gets fewer
            // anyway): with 16 groups per segment a 1024 x 1024 x 16-spp batch LOST 14 %; measured at 64 / 128 / 256 /
            // 512 groups, batches of 2^24, 2^26 and 2^28 paths: 28.1 30.5 32.0 / 28.3 31.4 32.1 / 26.4 30.5 31.3 /
            // 26.4 28.3 31.6 Grays/s (one generation: 26.3 26.0 26.9).
            const uint64_t resident = (uint64_t)per_cu * (uint64_t)ctx->num_cus;
            uint64_t min_groups = 128;
            if (const char* e = getenv("HPT_WF_MIN_GROUPS")) min_groups = (uint64_t)std::max(1, atoi(e));
            segments = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(generations * resident, std::max<uint64_t>(resident, cap_groups / min_groups)), cap_groups);
            if (fixed) segments = (uint32_t)std::min<uint64_t>(segments, fixed->live_slots);
            seg = (uint32_t)((cap_groups + segments - 1ull) / segments) * 64u;  // records per segment
            if (fixed) {
                // fixed buffers: whole segments must fit (rounding each segment up to 64 records can
                // take the batch past the buffer: then the segments shrink and the batch with them)
                if ((uint64_t)seg * segments > fixed->wf_paths) seg = (uint32_t)(fixed->wf_paths / segments) & ~63u;
                if (seg == 0) return fail(HPT_ERR_TOO_LARGE, "the fixed path queues (%llu records) are too small for %u segments", (unsigned long long)fixed->wf_paths, segments);
                wf = fixed->wf;
                live = fixed->live;
                break;
            }
            const int rc = ensure_wavefront(ctx, (uint64_t)seg * segments, segments);
            if (rc == HPT_OK) {
                live = ctx->d_live;
                break;
            }
            // the queues of a long batch can be tens of GiB: if the card cannot give them, trace shorter batches
            if (rc != HPT_ERR_OUT_OF_MEMORY || cap <= (1ull << 16)) return rc;
            cap = ((cap >> 1) + 63ull) & ~63ull;
        }
        const uint64_t batch = std::min<uint64_t>(cap, (uint64_t)seg * segments);  // paths per generate + bounces pair
        WfQueue q[2] = {{wf[0], wf[1], wf[2], wf[3]}, {wf[4], wf[5], wf[6], wf[7]}};
        // the generate kernel needs whole waves
        const int gblock = (wblock + 63) & ~63;
        for (uint64_t first = 0; first < total; first += batch) {
            const uint32_t count = (uint32_t)((total - first < batch) ? total - first : batch);
            hipLaunchKernelGGL(k_wf_generate, dim3(segments), dim3(gblock), 0, stream, S, sample_begin, first, count, (uint32_t)lanes, seg, q[0], live, ctx->d_stats);
            HIP_TRY(hipGetLastError());
            HPT_DISPATCH_RENDER(S.variant, ctx->extended, {
                if constexpr (has_scheduler(kS)) {
                    if (scheduler) hipLaunchKernelGGL((k_wf_bounces_sm<kS, kG>), dim3(segments), dim3(wblock), lds_bytes(S, wblock), stream, S, max_depth, seg, q[0], q[1], live, accum, ctx->d_stats);
                    else hipLaunchKernelGGL((k_wf_bounces<kS, kG>), dim3(segments), dim3(wblock), lds_bytes(S, wblock), stream, S, max_depth, seg, q[0], q[1], live, accum, ctx->d_stats);
                } else {
                    hipLaunchKernelGGL((k_wf_bounces<kS, kG>), dim3(segments), dim3(wblock), lds_bytes(S, wblock), stream, S, max_depth, seg, q[0], q[1], live, accum, ctx->d_stats);
                }
            })
            HIP_TRY(hipGetLastError());
            launches += 2;
        }
    }
    *launches_out = launches;
    return HPT_OK;
}

}  // namespace

// --------------------------------------------------------------------- ABI --

extern "C" {

const char* hpt_last_error(void) { return g_last_error.c_str(); }
const char* hpt_version(void) { return "hpt 0.1 (gfx950)"; }

int hpt_device_count(int* count) {
    if (!count) return fail(HPT_ERR_INVALID_ARGUMENT, "count is null");
    HIP_TRY(hipGetDeviceCount(count));
    return HPT_OK;
}

int hpt_context_create(int device, hpt_ctx** out) {
    if (!out) return fail(HPT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(HPT_ERR_INVALID_ARGUMENT, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    hpt_ctx* ctx = new hpt_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("HPT_SPP_PER_LAUNCH")) {
        int v = atoi(e);
        if (v > 0) ctx->spp_per_launch = (uint32_t)v;
    }
    if (const char* e = getenv("HPT_SPP_PER_UNIT")) {
        int v = atoi(e);
        if (v > 0) ctx->spp_per_unit = (uint32_t)v;
    }
    hipError_t e = hipMalloc((void**)&ctx->d_stats, kNumStats * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_stats, 0, kNumStats * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_unit_counter, sizeof(uint32_t));
    if (e != hipSuccess) {
        delete ctx;
        return fail(HPT_ERR_HIP, "stats buffer: %s", hipGetErrorString(e));
    }
    if (int rc = scratch_of(device, &ctx->scratch_dev)) {
        (void)hipFree(ctx->d_stats);
        (void)hipFree(ctx->d_unit_counter);
        delete ctx;
        return rc;
    }
#ifdef HPT_PROFILE
    (void)hipMalloc((void**)&ctx->d_prof, 3 * kProfSections * sizeof(unsigned long long));
    (void)hipMemset(ctx->d_prof, 0, 3 * kProfSections * sizeof(unsigned long long));
#endif
    {
        std::lock_guard<std::mutex> lock(g_comm_mutex);
        ++g_live_contexts;
    }
    *out = ctx;
    return HPT_OK;
}

int hpt_context_destroy(hpt_ctx* ctx) {
    if (!ctx) return HPT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)settle_events(ctx, true);
    for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
    free_scene(ctx);
    free_wavefront(ctx);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_unit_counter);
    delete ctx;
    {
        // the communicators of hpt_accum_reduce go with the process's last context
        std::lock_guard<std::mutex> lock(g_comm_mutex);
        if (--g_live_contexts == 0) comm_sets_destroy_all();
    }
    return HPT_OK;
}

int hpt_scene_upload(hpt_ctx* ctx, const hpt_scene_desc* sc) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (int rc = validate_scene(sc)) return rc;
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    free_scene(ctx);
    const size_t nn = sc->num_nodes, nt = sc->num_tris;
    ALLOC_TRY(ctx->d_nodes, nn * 32, "nodes");
    ALLOC_TRY(ctx->d_node_area, nn * 4, "node_area");
    ALLOC_TRY(ctx->d_tris, nt * 64, "tris");
    ALLOC_TRY(ctx->d_tri_v12, nt * 32, "tri_v12");
    ALLOC_TRY(ctx->d_mats, 2 * (size_t)sc->num_materials * sizeof(DevMaterial), "mats");
    ALLOC_TRY(ctx->d_lights, (sc->num_lights ? sc->num_lights : 1) * sizeof(DevLight), "lights");
    HIP_TRY(hipMemcpy(ctx->d_nodes, sc->nodes, nn * 32, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_node_area, sc->node_area, nn * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tri_v12, sc->tri_v12, nt * 32, hipMemcpyHostToDevice));
    std::vector<float> tris(sc->tris, sc->tris + nt * 16);  // the device copy of the triangle records, uploaded below
    {
        // derived tables (see DevMaterial / DevLight); plain float arithmetic in the
        // reference's order, this file is compiled without FMA contraction
        const float pi = 3.1415926535897932384626422832795028841971f;
        auto lum = [](const float* c) {  // Math::luminance, common/MathUtils.hpp:218-222
            const float tx = c[0] * .2126f, ty = c[1] * .7152f, tz = c[2] * .0722f;
            return tx + ty + tz;
        };
        // two tables back to back: the materials as given, and after Material::regularize()
        // (common/Material.hpp:164-171: alpha = sqrt(roughness) < 0.3 -> clamp(2 alpha, 0.1, 0.3)^2)
        std::vector<DevMaterial> mats(2 * (size_t)sc->num_materials);
        bool glass = false;
        for (uint32_t i = 0; i < 2 * sc->num_materials; ++i) {
            hpt_material m = sc->materials[i % sc->num_materials];
            if (i >= sc->num_materials) {
                float alpha_x = std::sqrt(m.roughness);
                if (alpha_x < 0.3f) {
                    alpha_x = std::min(std::max(2 * alpha_x, 0.1f), 0.3f);
                    m.roughness = alpha_x * alpha_x;
                }
            }
            DevMaterial& d = mats[i];
            std::memset(&d, 0, sizeof d);
            d.type = m.type;
            const float ee = (m.emission[0] * m.emission[0] + m.emission[1] * m.emission[1]) + m.emission[2] * m.emission[2];
            const bool specular = (m.type == 1 && std::sqrt(m.roughness) < 1e-3f) || m.type == 2;  // Material.hpp:151-162
            d.flags = (ee > 0.0f ? kMatEmitting : 0u) | (specular ? kMatSpecular : 0u);
            for (int c = 0; c < 3; ++c) {
                d.emission[c] = m.emission[c];
                d.albedo[c] = m.albedo[c];
                d.albedo_over_pi[c] = (m.albedo[c] * 1.f) / pi;
                d.f0[c] = 0.04f * (1.0f - m.metallic) + m.albedo[c] * m.metallic;  // Math::lerp, MathUtils.hpp:182-184
            }
            d.metallic = m.metallic;
            d.one_minus_metallic = 1.0f - m.metallic;
            d.roughness = m.roughness;
            d.rsq = m.roughness * m.roughness;
            d.rsq_minus_1 = d.rsq - 1.0f;
            d.ior = m.ior;
            if (m.type == 2) glass = true;
        }
        std::vector<DevLight> lights(sc->num_lights ? sc->num_lights : 1);
        std::memset(lights.data(), 0, lights.size() * sizeof(DevLight));
        for (uint32_t i = 0; i < sc->num_lights; ++i) {
            const hpt_light& l = sc->lights[i];
            DevLight& d = lights[i];
            d.area = l.area;
            d.root_area = l.root_area;
            d.inv_root_area = 1.0f / l.root_area;
            d.root_node = l.root_node;
            for (int c = 0; c < 3; ++c) d.emission[c] = sc->materials[l.material].emission[c];
        }
        for (size_t t = 0; t < nt; ++t) {
            int32_t mat;
            std::memcpy(&mat, &tris[16 * t + 7], 4);
            const float area = tris[16 * t + 3];
            // luminance(Le) * 2 * pi * triangleArea / sumLightPower (common/Scene.hpp:179, :215)
            tris[16 * t + 3] = lum(sc->materials[mat].emission) * 2.f * pi * area / sc->sum_light_power;
        }
        bool has_sphere = false;
        for (size_t t = 0; t < nt; ++t) {
            uint32_t w15;
            std::memcpy(&w15, &tris[16 * t + 15], 4);
            if (w15 & kSphereBit) has_sphere = true;
        }
        ctx->has_sphere = has_sphere;
        HIP_TRY(hipMemcpy(ctx->d_mats, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice));
        // (kernels built without kExt read their materials from LDS: scenes with more materials than fit there take the others)
        ctx->scene_extended = glass || has_sphere || sc->num_materials > kMaxLdsMaterials;
        ctx->extended = ctx->scene_extended || ctx->integrator != 0;
        HIP_TRY(hipMemcpy(ctx->d_lights, lights.data(), lights.size() * sizeof(DevLight), hipMemcpyHostToDevice));
    }
    // Two-child records of the same tree (see DevScene::pairs).  Interior node i has the children
    // i + 1 and miss(i + 1); records are numbered in pre-order of the interior nodes.
    uint32_t num_pairs = 0, stack_depth = 1, stack_depth_any = 1, ref_root = 0;
    float root_lo[3] = {sc->nodes[0], sc->nodes[1], sc->nodes[2]}, root_hi[3] = {sc->nodes[4], sc->nodes[5], sc->nodes[6]};
    const bool gpu_build = ctx->bvh_build == HPT_BVH_BUILD_GPU;
    if (gpu_build && sc->ref_nodes) return fail(HPT_ERR_INVALID_ARGUMENT, "HPT_BVH_BUILD_GPU builds its own records: a scene with ref_nodes must be uploaded with HPT_BVH_BUILD_HOST");
    if (gpu_build) {
        // the records come from the triangle array alone (csrc/hpt_bvh_build.hip); `nodes` only serves the light sampling
        if (nt < 2) return fail(HPT_ERR_INVALID_ARGUMENT, "HPT_BVH_BUILD_GPU needs at least two primitives");
        HIP_TRY(hipMemcpy(ctx->d_tris, tris.data(), nt * 64, hipMemcpyHostToDevice));
        num_pairs = (uint32_t)nt - 1u;
        ALLOC_TRY(ctx->d_pairs, (size_t)num_pairs * 64, "pairs");
        // an object = a run of primitives with the same object word (word 15 without the sphere bit)
        std::vector<uint32_t> obj_first;
        uint32_t prev = 0xFFFFFFFFu;
        for (size_t t = 0; t < nt; ++t) {
            uint32_t w15;
            std::memcpy(&w15, &tris[16 * t + 15], 4);
            if ((w15 & ~kSphereBit) != prev || (w15 & kSphereBit)) obj_first.push_back((uint32_t)t);  // a sphere is an object of its own
            prev = (w15 & kSphereBit) ? 0xFFFFFFFFu : (w15 & ~kSphereBit);
        }
        const std::string err = hpt_build::build_pairs(ctx->d_tris, ctx->d_tri_v12, (uint32_t)nt, obj_first.data(), (uint32_t)obj_first.size(), ctx->d_nodes,
                                                       (uint32_t)nn, ctx->d_pairs, &stack_depth, &stack_depth_any, root_lo, root_hi, &ctx->bvh_build_ms);
        if (!err.empty()) return fail(HPT_ERR_HIP, "GPU BVH build: %s", err.c_str());
    } else {
        // records of one tree, appended to `rec`; child references count from the start of `rec`
        auto derive = [&](const float* nodes, size_t n_nodes, std::vector<float>& rec, uint32_t& need0, uint32_t& need_any0) -> uint32_t {
            auto node_int = [&](size_t i, int word) {
                int32_t v;
                std::memcpy(&v, &nodes[8 * i + word], 4);
                return v;
            };
            const uint32_t first = (uint32_t)(rec.size() / 16);
            uint32_t count = 0;
            std::vector<int32_t> pair_of(n_nodes, -1);
            for (size_t i = 0; i < n_nodes; ++i)
                if (node_int(i, 7) < 0) pair_of[i] = (int32_t)(first + count++);
            need0 = need_any0 = 1u;
            if (!count) return 0u;
            rec.resize((size_t)(first + count) * 16);
            std::vector<uint32_t> need(count, 0);      // stack entries a walk below this record can hold at once
            std::vector<uint32_t> need_any(count, 0);  // the same when either child may be walked first
            for (size_t i = n_nodes; i-- > 0;) {
                if (pair_of[i] < 0) continue;
                const size_t l = i + 1, r = (size_t)node_int(l, 3);
                float* q = &rec[(size_t)pair_of[i] * 16];
                const int32_t tl = node_int(l, 7), tr = node_int(r, 7);
                const int32_t ref_l = tl >= 0 ? ~tl : pair_of[l], ref_r = tr >= 0 ? ~tr : pair_of[r];
                for (int c = 0; c < 3; ++c) {
                    q[c] = nodes[8 * l + c];
                    q[4 + c] = nodes[8 * l + 4 + c];
                    q[8 + c] = nodes[8 * r + c];
                    q[12 + c] = nodes[8 * r + 4 + c];
                }
                std::memcpy(&q[3], &ref_l, 4);
                std::memcpy(&q[7], &ref_r, 4);
                q[11] = q[15] = 0.f;
                // the right child waits on the stack while the left subtree is walked
                const uint32_t me = (uint32_t)pair_of[i] - first;
                const uint32_t nl = ref_l >= 0 ? need[(uint32_t)ref_l - first] + (ref_r >= 0 ? 1u : 0u) : 0u;
                const uint32_t nr = ref_r >= 0 ? need[(uint32_t)ref_r - first] : 0u;
                need[me] = std::max(nl, nr);
                const uint32_t both = (ref_l >= 0 && ref_r >= 0) ? 1u : 0u;
                need_any[me] = std::max(ref_l >= 0 ? need_any[(uint32_t)ref_l - first] : 0u, ref_r >= 0 ? need_any[(uint32_t)ref_r - first] : 0u) + both;
            }
            need0 = std::max(1u, need[0]);
            need_any0 = std::max(1u, need_any[0]);
            return count;
        };
        std::vector<float> rec;
        num_pairs = derive(sc->nodes, nn, rec, stack_depth, stack_depth_any);
        if (sc->ref_nodes && num_pairs) {
            // The reference's own tree behind the caller's: rays with a zero direction component start at its root
            // (DevScene::ref_root).  For them a box test can come out NaN-false whatever the boxes' nesting, so which
            // leaves they reach depends on the tree; every other ray reaches the same leaves in any tree over them.
            uint32_t d1 = 1, d2 = 1;
            ref_root = num_pairs;
            const uint32_t more = derive(sc->ref_nodes, sc->num_ref_nodes, rec, d1, d2);
            if (!more) ref_root = 0;
            num_pairs += more;
            stack_depth = std::max(stack_depth, d1);
            stack_depth_any = std::max(stack_depth_any, d2);
        }
        if (num_pairs) {
            ALLOC_TRY(ctx->d_pairs, rec.size() * 4, "pairs");
            HIP_TRY(hipMemcpy(ctx->d_pairs, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
        }
    }
    ctx->accum_elems = 3ull * (size_t)sc->width * (size_t)sc->height;
    ALLOC_TRY(ctx->d_accum, ctx->accum_elems * sizeof(int64_t), "accum");
    HIP_TRY(hipMemset(ctx->d_accum, 0, ctx->accum_elems * sizeof(int64_t)));
    ALLOC_TRY(ctx->d_rgb, ctx->accum_elems * sizeof(float), "rgb");

    DevScene& S = ctx->S;
    const uint32_t seed = S.seed;
    S = DevScene{};
    S.integrator = ctx->integrator;
    S.roulette_q = 1 - ctx->russian_roulette;
    S.has_spheres = ctx->has_sphere ? 1u : 0u;
    S.nodes = ctx->d_nodes;
    S.node_area = ctx->d_node_area;
    S.tris = ctx->d_tris;
    S.tri_v12 = ctx->d_tri_v12;
    S.mats = ctx->d_mats;
    S.lights = ctx->d_lights;
    S.num_nodes = sc->num_nodes;
    S.num_tris = sc->num_tris;
    S.num_mats = sc->num_materials;
    S.num_lights = sc->num_lights;
    S.sum_light_area = sc->sum_light_area;
    S.sum_light_power = sc->sum_light_power;
    S.cam_x = sc->cam_pos[0];
    S.cam_y = sc->cam_pos[1];
    S.cam_z = sc->cam_pos[2];
    S.cam_scale = sc->cam_scale;
    S.aspect = sc->width / (float)sc->height;  // Renderer.cpp:21
    S.width = sc->width;
    S.height = sc->height;
    S.seed = seed;
    S.prof = ctx->d_prof;
    S.pairs = ctx->d_pairs;
    S.num_pairs = num_pairs;
    S.stack_depth = stack_depth;
    for (int c = 0; c < 3; ++c) {
        S.root_lo[c] = root_lo[c];
        S.root_hi[c] = root_hi[c];
    }
    const uint64_t f4 = 2ull * nn + 4ull * nt;             // threaded nodes + triangles
    const uint64_t f4_tiny = 18ull * nn + 4ull * nt + 8ull;  // nine copies of the nodes + triangles + table
    const uint64_t f4_pair = 4ull * num_pairs + 4ull * nt;  // two-child records + triangles
    // records built on the GPU: `nodes` is not the scene's tree, so only the two-child-record walks apply
    // (a scene that brings the reference's tree beside its own: the threaded walk from LDS has no second array, and
    // the octant copies take their ninth copy from it, which must then have as many nodes as the others)
    const bool fits = !gpu_build && !sc->ref_nodes && f4 * 16ull <= kLdsLimit, fits_pair = f4_pair * 16ull <= kLdsLimit;
    const bool fits_tiny = !gpu_build && f4_tiny * 16ull <= kLdsLimitTiny && (!sc->ref_nodes || sc->num_ref_nodes == sc->num_nodes);
    const bool pair_ok = num_pairs > 0 && stack_depth <= kMaxStackDepth;
    const bool tiny_ok = fits_tiny && !ctx->has_sphere;  // the marking variants test triangles only
    if (tiny_ok && nt <= 32) S.variant = kSceneTiny32;
    else if (tiny_ok && nt <= 64) S.variant = kSceneTiny64;
    else if (pair_ok) S.variant = fits_pair ? kScenePairLds : kScenePair;
    else S.variant = fits ? kSceneLds : kSceneGlobal;
    if (gpu_build && !pair_ok) return fail(HPT_ERR_INVALID_ARGUMENT, "GPU BVH build: the tree needs %u stack entries (limit %u)", stack_depth, kMaxStackDepth);
    // the culled walk packs a record index into 22 bits of a stack entry
    const bool cull_ok = num_pairs > 0 && num_pairs < (1u << 22) && stack_depth_any <= kMaxStackDepth;
    if (ctx->traversal == HPT_TRAVERSAL_CULLED && cull_ok && (S.variant == kScenePair || S.variant == kScenePairLds)) S.variant = kScenePairCull;
    if (const char* e = getenv("HPT_TRACE_VARIANT")) {  // experiments: force another variant
        const int v = atoi(e);
        if (v == kScenePairCull && cull_ok) S.variant = kScenePairCull;
        if (v == kSceneGlobal && !gpu_build) S.variant = kSceneGlobal;
        else if (v == kSceneLds && fits) S.variant = kSceneLds;
        else if (v == kSceneTiny64 && tiny_ok && nt <= 64) S.variant = kSceneTiny64;
        else if (v == kSceneTiny32 && tiny_ok && nt <= 32) S.variant = kSceneTiny32;
        else if (v == kScenePair && pair_ok) S.variant = kScenePair;
        else if (v == kScenePairLds && pair_ok && fits_pair) S.variant = kScenePairLds;
    }
    // Word 15 of the device copy (the mesh index, which no kernel reads): for scenes of up to 32
    // triangles the mask of the triangles that have this triangle's normal (a zero of either
    // sign counts as the same value: it contributes +-0 to dot(d, n) either way).  The
    // back-face filter of the 32-triangle marking variant settles a whole group with one test.
    if (S.variant == kSceneTiny32) {
        auto same = [](float a, float b) { return (a == 0.f && b == 0.f) || std::memcmp(&a, &b, 4) == 0; };
        for (size_t t = 0; t < nt; ++t) {
            uint32_t group = 0;
            for (size_t u = 0; u < nt; ++u)
                if (same(tris[16 * t + 12], tris[16 * u + 12]) && same(tris[16 * t + 13], tris[16 * u + 13]) && same(tris[16 * t + 14], tris[16 * u + 14]))
                    group |= 1u << u;
            std::memcpy(&tris[16 * t + 15], &group, 4);
        }
    }
    if (!gpu_build) HIP_TRY(hipMemcpy(ctx->d_tris, tris.data(), nt * 64, hipMemcpyHostToDevice));  // (the GPU build has gathered its copy into leaf order)
    if (S.variant == kScenePairCull) {
        S.stack_depth = stack_depth_any;
        double diag2 = 0;
        for (int c = 0; c < 3; ++c) diag2 += ((double)S.root_hi[c] - S.root_lo[c]) * ((double)S.root_hi[c] - S.root_lo[c]);
        const double diag = std::sqrt(diag2) > 0 ? std::sqrt(diag2) : 1.0;
        S.cull_slack = (float)(1e-4 * diag);
        S.cull_qscale = (float)(1023.0 / diag);
        S.cull_qinv = (float)(diag / 1023.0 * (1.0 - 1e-6));  // rounded down: the unpacked distance never exceeds the packed one
    }
    if (S.variant == kSceneTiny32 || S.variant == kSceneTiny64) {
        std::vector<float> oct_nodes;
        std::vector<uint32_t> oct_table;
        build_octant_copies(sc, S.variant == kSceneTiny32, oct_nodes, oct_table);
        ALLOC_TRY(ctx->d_oct_nodes, oct_nodes.size() * 4, "oct_nodes");
        ALLOC_TRY(ctx->d_oct_table, oct_table.size() * 4, "oct_table");
        HIP_TRY(hipMemcpy(ctx->d_oct_nodes, oct_nodes.data(), oct_nodes.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_oct_table, oct_table.data(), oct_table.size() * 4, hipMemcpyHostToDevice));
        S.oct_nodes = ctx->d_oct_nodes;
        S.oct_table = ctx->d_oct_table;
    }
    S.ref_root = ref_root;
    S.ref_nodes = S.nodes;
    S.num_ref_nodes = S.num_nodes;
    if (sc->ref_nodes && S.variant == kSceneGlobal) {
        ALLOC_TRY(ctx->d_ref_nodes, (size_t)sc->num_ref_nodes * 32, "ref_nodes");
        HIP_TRY(hipMemcpy(ctx->d_ref_nodes, sc->ref_nodes, (size_t)sc->num_ref_nodes * 32, hipMemcpyHostToDevice));
        S.ref_nodes = ctx->d_ref_nodes;
        S.num_ref_nodes = sc->num_ref_nodes;
    }
    switch (S.variant) {
    case kSceneLds: S.lds_float4s = (uint32_t)f4; break;
    case kSceneTiny32:
    case kSceneTiny64: S.lds_float4s = (uint32_t)f4_tiny; break;
    case kScenePairLds: S.lds_float4s = (uint32_t)f4_pair; break;
    default: S.lds_float4s = 0u; break;
    }
    // the materials ride behind the staged scene (or alone) in every workgroup's LDS when they are few
    S.mat_lds_off = S.lds_float4s;
    S.mat_lds_float4s = sc->num_materials <= kMaxLdsMaterials ? 2u * sc->num_materials * (uint32_t)(sizeof(DevMaterial) / 16u) : 0u;
    S.lds_float4s += S.mat_lds_float4s;
    // DevScene::fast_light: one light whose mesh tree is a root over two leaves
    if (sc->num_lights == 1) {
        const hpt_light& l = sc->lights[0];
        auto word = [&](size_t node, int w) {
            int32_t v;
            std::memcpy(&v, &sc->nodes[8 * node + w], 4);
            return v;
        };
        const size_t root = l.root_node;
        const char* e = getenv("HPT_FAST_LIGHT");  // 0: the general light-sampling code (A/B runs, tests)
        if (word(root, 7) < 0 && root + 2 < (size_t)sc->num_nodes + 0u && !(e && atoi(e) == 0)) {
            const size_t left = root + 1, right = (size_t)word(left, 3);
            if (right < sc->num_nodes && word(left, 7) >= 0 && word(right, 7) >= 0) {
                float f[4 * kFastLightFloat4s] = {};
                const int32_t t2[2] = {word(left, 7), word(right, 7)};
                for (int k = 0; k < 2; ++k) {
                    const float* r = &tris[16 * (size_t)t2[k]];  // the device copy's values (word 3 = MIS light pdf)
                    const float* v = &sc->tri_v12[8 * (size_t)t2[k]];
                    for (int c = 0; c < 3; ++c) {
                        f[16 * k + c] = r[c];
                        f[16 * k + 4 + c] = v[c];
                        f[16 * k + 8 + c] = v[4 + c];
                        f[16 * k + 12 + c] = r[12 + c];
                    }
                    f[16 * k + 3] = r[3];
                }
                f[32] = l.area;
                f[33] = l.root_area;
                f[34] = 1.0f / l.root_area;
                f[35] = sc->node_area[left];
                for (int c = 0; c < 3; ++c) f[36 + c] = sc->materials[l.material].emission[c];
                ALLOC_TRY(ctx->d_fast_light, sizeof f, "fast_light");
                HIP_TRY(hipMemcpy(ctx->d_fast_light, f, sizeof f, hipMemcpyHostToDevice));
                S.fast_light = ctx->d_fast_light;
                S.light_lds_off = S.lds_float4s;
                S.lds_float4s += kFastLightFloat4s;
            }
        }
    }
    ctx->has_scene = true;
    return HPT_OK;
}

int hpt_set_traversal(hpt_ctx* ctx, int traversal) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (traversal != HPT_TRAVERSAL_REFERENCE && traversal != HPT_TRAVERSAL_CULLED)
        return fail(HPT_ERR_INVALID_ARGUMENT, "traversal %d is neither HPT_TRAVERSAL_REFERENCE nor HPT_TRAVERSAL_CULLED", traversal);
    ctx->traversal = traversal;
    return HPT_OK;
}

int hpt_set_bvh_build(hpt_ctx* ctx, int build) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (build != HPT_BVH_BUILD_HOST && build != HPT_BVH_BUILD_GPU)
        return fail(HPT_ERR_INVALID_ARGUMENT, "build %d is neither HPT_BVH_BUILD_HOST nor HPT_BVH_BUILD_GPU", build);
    ctx->bvh_build = build;
    return HPT_OK;
}

int hpt_scene_build_info(hpt_ctx* ctx, double* gpu_build_ms, uint32_t* num_records, uint32_t* stack_depth) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (gpu_build_ms) *gpu_build_ms = ctx->bvh_build == HPT_BVH_BUILD_GPU ? ctx->bvh_build_ms : 0.0;
    if (num_records) *num_records = ctx->S.num_pairs;
    if (stack_depth) *stack_depth = ctx->S.stack_depth;
    return HPT_OK;
}

int hpt_set_integrator(hpt_ctx* ctx, uint32_t flags) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (flags & ~(uint32_t)(HPT_INTEGRATOR_RUSSIAN_ROULETTE | HPT_INTEGRATOR_REGULARIZE))
        return fail(HPT_ERR_INVALID_ARGUMENT, "unknown integrator flag in 0x%x", flags);
    ctx->integrator = flags;
    ctx->S.integrator = flags;
    ctx->extended = ctx->scene_extended || flags != 0;
    return HPT_OK;
}

int hpt_set_seed(hpt_ctx* ctx, uint32_t seed) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    ctx->S.seed = seed;
    return HPT_OK;
}

int hpt_set_wavefront_capacity(hpt_ctx* ctx, uint64_t paths) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (paths < 64 || paths > (1ull << 31)) return fail(HPT_ERR_INVALID_ARGUMENT, "wavefront capacity must be in [64, 2^31]");
    ctx->wf_capacity = paths & ~63ull;
    return HPT_OK;
}

int hpt_accum_buffer(hpt_ctx* ctx, int64_t** accum_dev, size_t* num_elements) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (accum_dev) *accum_dev = ctx->d_accum;
    if (num_elements) *num_elements = ctx->accum_elems;
    return HPT_OK;
}

int hpt_synchronize(hpt_ctx* ctx) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return HPT_OK;
}

int hpt_accum_download(hpt_ctx* ctx, const int64_t* accum_dev, int64_t* accum_host) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (!accum_host) return fail(HPT_ERR_INVALID_ARGUMENT, "accum_host is null");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_host, accum_dev ? accum_dev : ctx->d_accum, ctx->accum_elems * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HPT_OK;
}

static int check_reduce_args(hpt_ctx* const* ctxs, int n, int root) {
    if (!ctxs || n < 1) return fail(HPT_ERR_INVALID_ARGUMENT, "hpt_accum_reduce: no contexts");
    if (root < 0 || root >= n) return fail(HPT_ERR_INVALID_ARGUMENT, "root %d not in [0, %d)", root, n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->has_scene) return fail(HPT_ERR_NO_SCENE, "context %d has no scene", i);
        if (ctxs[i]->accum_elems != ctxs[root]->accum_elems) return fail(HPT_ERR_INVALID_ARGUMENT, "context %d renders another resolution than the root", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(HPT_ERR_INVALID_ARGUMENT, "context %d is listed twice", i);
    }
    return HPT_OK;
}

// one context index per distinct device, in order of first appearance: the root if it lives on that
// device, else the first context listed there
static std::vector<int> reduce_representatives(hpt_ctx* const* ctxs, int n, int root) {
    std::vector<int> rep;
    for (int i = 0; i < n; ++i) {
        bool seen = false;
        for (int k : rep)
            if (ctxs[k]->device == ctxs[i]->device) seen = true;
        if (!seen) rep.push_back(ctxs[root]->device == ctxs[i]->device ? root : i);
    }
    return rep;
}

static bool reduce_forced_through_rccl() {
    const char* force = getenv("HPT_REDUCE_FORCE_RCCL");  // a one-device communicator all the same (tests on a one-GPU machine)
    return force && atoi(force) != 0;
}

int hpt_accum_reduce_prepare(hpt_ctx* const* ctxs, int n) {
    if (int rc = check_reduce_args(ctxs, n, 0)) return rc;
    const std::vector<int> rep = reduce_representatives(ctxs, n, 0);
    if (rep.size() == 1 && !reduce_forced_through_rccl()) return HPT_OK;
    std::vector<int> devs;
    for (int k : rep) devs.push_back(ctxs[k]->device);
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    const CommSet* set = nullptr;
    return comm_set_for(devs, &set);
}

int hpt_accum_reduce(hpt_ctx* const* ctxs, int n, int root) {
    if (int rc = check_reduce_args(ctxs, n, root)) return rc;
    const size_t count = ctxs[root]->accum_elems;
    // every context's queued renders must have finished before another stream or device reads its buffer
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(ctxs[i]->device));
        HIP_TRY(hipDeviceSynchronize());
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto stamp = [&]() {
        std::lock_guard<std::mutex> lock(g_comm_mutex);
        g_last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // 1. contexts that share a device are summed there, into the device's representative
    const std::vector<int> rep = reduce_representatives(ctxs, n, root);
    for (int i = 0; i < n; ++i) {
        int r = -1;
        for (int k : rep)
            if (ctxs[k]->device == ctxs[i]->device) r = k;
        if (r == i) continue;
        HIP_TRY(hipSetDevice(ctxs[i]->device));
        const int grid = (int)std::min<size_t>((count + 255) / 256, 4096);
        hipLaunchKernelGGL(k_accum_add, dim3(grid), dim3(256), 0, nullptr, ctxs[r]->d_accum, ctxs[i]->d_accum, count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    // 2. one RCCL reduce (sum of int64, associative: the result does not depend on the order) over the
    //    representatives; a single device needs none
    if (rep.size() == 1 && !reduce_forced_through_rccl()) {
        stamp();
        return HPT_OK;
    }
    const int nd = (int)rep.size();
    std::vector<int> devs(nd);
    int root_rank = 0;
    for (int k = 0; k < nd; ++k) {
        devs[k] = ctxs[rep[k]]->device;
        if (rep[k] == root) root_rank = k;
    }
    // the communicators of this device list: made by the first reduce over it (or by
    // hpt_accum_reduce_prepare), reused by every later one.  The lock also keeps two reduces of one
    // process from interleaving their grouped calls on the same communicators.
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    const CommSet* set = nullptr;
    if (int rc = comm_set_for(devs, &set)) return rc;
    int r = g_rccl.GroupStart();
    for (int k = 0; k < nd && r == 0; ++k) {
        if (hipSetDevice(devs[k]) != hipSuccess) {
            r = -1;
            break;
        }
        int64_t* buf = ctxs[rep[k]]->d_accum;
        r = g_rccl.Reduce(buf, buf, count, /*ncclInt64*/ 4, /*ncclSum*/ 0, root_rank, set->comms[k], nullptr);
    }
    const int e = g_rccl.GroupEnd();
    if (r == 0) r = e;
    if (r != 0) return fail(HPT_ERR_HIP, "RCCL reduce failed: %s", r > 0 ? g_rccl.GetErrorString(r) : "hipSetDevice");
    for (int k = 0; k < nd; ++k)
        if (hipSetDevice(devs[k]) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return fail(HPT_ERR_HIP, "synchronising device %d after the reduce failed", devs[k]);
    g_last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HPT_OK;
}

int hpt_accum_reduce_info(uint64_t* communicators_created, double* last_reduce_ms) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    if (communicators_created) *communicators_created = g_comm_created;
    if (last_reduce_ms) *last_reduce_ms = g_last_reduce_ms;
    return HPT_OK;
}

int hpt_octant_copies(const hpt_scene_desc* sc, int leaf_bits, float* nodes_out, uint32_t* table_out) {
    if (!sc || !nodes_out || !table_out) return fail(HPT_ERR_INVALID_ARGUMENT, "null argument");
    if (int rc = validate_scene(sc)) return rc;
    if (sc->num_tris > (leaf_bits ? 32u : 64u)) return fail(HPT_ERR_INVALID_ARGUMENT, "%u triangles: the octant copies serve scenes of up to %u", sc->num_tris, leaf_bits ? 32u : 64u);
    std::vector<float> nodes;
    std::vector<uint32_t> table;
    build_octant_copies(sc, leaf_bits != 0, nodes, table);
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(float));
    std::memcpy(table_out, table.data(), table.size() * sizeof(uint32_t));
    return HPT_OK;
}

int hpt_accum_clear(hpt_ctx* ctx, int64_t* accum_dev, void* stream) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(accum_dev ? accum_dev : ctx->d_accum, 0, ctx->accum_elems * sizeof(int64_t), (hipStream_t)stream));
    return HPT_OK;
}

int hpt_render(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count,
               int64_t* accum_dev, void* stream_) {
    if (int rc = check_render_args(ctx, mode, block, max_depth, sample_begin, sample_count)) return rc;
    if (int rc = set_device(ctx)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (sample_count == 0 || max_depth == 0) return HPT_OK;  // castRay's loop body never runs: radiance 0
    if (int rc = settle_events(ctx, false)) return rc;
    // The unit counter and the wavefront queues belong to the context: a render on another stream
    // than the previous one waits for it instead of racing with it.
    if (!ctx->pending_events.empty() && ctx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->pending_events.back().second, 0));
    ctx->last_stream = stream;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (int rc = take_event(ctx, &ev0)) return rc;
    if (int rc = take_event(ctx, &ev1)) {
        ctx->free_events.push_back(ev0);
        return rc;
    }
    uint32_t launches = 0;
    int rc = HPT_OK;
    hipError_t e = hipEventRecord(ev0, stream);
    if (e == hipSuccess) {
        rc = render_launch(ctx, mode, block, max_depth, sample_begin, sample_count, accum_dev ? accum_dev : ctx->d_accum, stream, nullptr, &launches);
        if (rc == HPT_OK) e = hipEventRecord(ev1, stream);
    }
    if (rc != HPT_OK || e != hipSuccess) {
        ctx->free_events.push_back(ev0);
        ctx->free_events.push_back(ev1);
        return rc != HPT_OK ? rc : fail(HPT_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    ctx->pending_events.emplace_back(ev0, ev1);
    ctx->launches += launches;
    return HPT_OK;
}

int hpt_render_static(hpt_ctx* ctx, int mode, int block, int max_depth, uint32_t sample_begin, uint32_t sample_count,
                      int64_t* accum_dev, void* stream_) {
    if (int rc = check_render_args(ctx, mode, block, max_depth, sample_begin, sample_count)) return rc;
    if (int rc = set_device(ctx)) return rc;
    if (sample_count == 0 || max_depth == 0) return HPT_OK;
    const DevScene& S = ctx->S;
    const uint64_t lanes = (uint64_t)(((uint32_t)S.width + 7u) >> 3) * (((uint32_t)S.height + 7u) >> 3) * 64ull;
    if (lanes * (uint64_t)sample_count > kStaticPaths)
        return fail(HPT_ERR_TOO_LARGE, "hpt_render_static traces at most %llu camera paths per call (8x8-pixel tiles x 64 x samples); this call asks for %llu",
                    (unsigned long long)kStaticPaths, (unsigned long long)(lanes * sample_count));
    unsigned char* base = ctx->scratch_dev;  // the array's address on this GPU, never the symbol's name
    RenderBuffers buf{};
    buf.unit_counter = reinterpret_cast<uint32_t*>(base);
    buf.live = reinterpret_cast<uint32_t*>(base + kScratchCounterBytes);
    buf.live_slots = kStaticSegments;
    buf.wf_paths = kStaticPaths;
    for (int i = 0; i < 8; ++i) buf.wf[i] = reinterpret_cast<float4*>(base + kScratchCounterBytes + kScratchLiveBytes) + (size_t)i * kStaticPaths;
    // Nothing in the scratch has to start at zero except the unit counter, which render_launch
    // clears on the stream before each single-kernel launch; the generate kernel writes every
    // live counter and every path record the bounce kernel reads.
    uint32_t launches = 0;
    hipStream_t stream = (hipStream_t)stream_;
    // order this call after the previous one on this GPU, whatever context or stream that came from
    // (the lock is held over the launches so that two host threads cannot interleave theirs)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
        (void)hipGetLastError();
        capturing = hipStreamCaptureStatusNone;
    }
    const bool ordered = capturing == hipStreamCaptureStatusNone;
    if (ordered && g_scratch_event_recorded[ctx->device]) HIP_TRY(hipStreamWaitEvent(stream, g_scratch_event[ctx->device], 0));
    if (int rc = render_launch(ctx, mode, block, max_depth, sample_begin, sample_count, accum_dev ? accum_dev : ctx->d_accum, stream, &buf, &launches))
        return rc;
    if (ordered) {
        HIP_TRY(hipEventRecord(g_scratch_event[ctx->device], stream));
        g_scratch_event_recorded[ctx->device] = true;
    }
    ctx->launches += launches;
    return HPT_OK;
}

int hpt_resolve_dev(hpt_ctx* ctx, const int64_t* accum_dev, uint32_t spp_total, float* rgb_dev, void* stream) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (!rgb_dev) return fail(HPT_ERR_INVALID_ARGUMENT, "rgb_dev is null");
    if (spp_total == 0) return fail(HPT_ERR_INVALID_ARGUMENT, "spp_total is 0");
    if (int rc = set_device(ctx)) return rc;
    const size_t n = ctx->accum_elems;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, (hipStream_t)stream, accum_dev ? accum_dev : ctx->d_accum, n,
                       16777216.0 * (double)spp_total, rgb_dev);
    HIP_TRY(hipGetLastError());
    return HPT_OK;
}

int hpt_resolve(hpt_ctx* ctx, const int64_t* accum_dev, uint32_t spp_total, float* rgb_host, void* stream) {
    if (!rgb_host) return fail(HPT_ERR_INVALID_ARGUMENT, "rgb_host is null");
    if (int rc = hpt_resolve_dev(ctx, accum_dev, spp_total, ctx ? ctx->d_rgb : nullptr, stream)) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_host, ctx->d_rgb, ctx->accum_elems * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return HPT_OK;
}

int hpt_intersect_batch_dev(hpt_ctx* ctx, size_t n, const float* o_dev, const float* d_dev, const uint8_t* shadow_dev,
                            const double* tmax_dev, int32_t* prim_dev, double* t_dev, int block, void* stream) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return HPT_OK;
    if (!o_dev || !d_dev || !prim_dev || !t_dev) return fail(HPT_ERR_INVALID_ARGUMENT, "null ray/result pointer");
    if (shadow_dev && !tmax_dev) return fail(HPT_ERR_INVALID_ARGUMENT, "shadow flags given without tmax");
    if (block < 1 || block > 1024) return fail(HPT_ERR_INVALID_ARGUMENT, "block size %d not in [1, 1024]", block);
    if (int rc = set_device(ctx)) return rc;
    const DevScene& S = ctx->S;
    const uint32_t lds = lds_bytes(S, block);
    const int grid = grid_for(ctx, n, block);
    HPT_DISPATCH_SCENE(S.variant, {
        hipLaunchKernelGGL((k_intersect_batch<kS>), dim3(grid), dim3(block), lds, (hipStream_t)stream, S, n, o_dev, d_dev, shadow_dev, tmax_dev, prim_dev, t_dev);
    })
    HIP_TRY(hipGetLastError());
    return HPT_OK;
}

int hpt_intersect_batch(hpt_ctx* ctx, size_t n, const float* o, const float* d, const uint8_t* shadow, const double* tmax,
                        int32_t* prim, double* t) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return HPT_OK;
    if (!o || !d || !prim || !t) return fail(HPT_ERR_INVALID_ARGUMENT, "null ray/result pointer");
    if (shadow && !tmax) return fail(HPT_ERR_INVALID_ARGUMENT, "shadow flags given without tmax");
    if (int rc = set_device(ctx)) return rc;
    float *d_o = nullptr, *d_d = nullptr;
    uint8_t* d_s = nullptr;
    double *d_tm = nullptr, *d_t = nullptr;
    int32_t* d_p = nullptr;
    int rc = HPT_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_s); (void)hipFree(d_tm); (void)hipFree(d_t); (void)hipFree(d_p);
    };
#define TRY_OR_CLEAN(expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) { cleanup(); return fail(HPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
#define ALLOC_OR_CLEAN(ptr, bytes)                                              \
    do {                                                                        \
        if (int rc_ = device_alloc((void**)&(ptr), (bytes), "ray batch staging")) { cleanup(); return rc_; } \
    } while (0)
    ALLOC_OR_CLEAN(d_o, n * 12);
    ALLOC_OR_CLEAN(d_d, n * 12);
    ALLOC_OR_CLEAN(d_p, n * 4);
    ALLOC_OR_CLEAN(d_t, n * 8);
    TRY_OR_CLEAN(hipMemcpy(d_o, o, n * 12, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_d, d, n * 12, hipMemcpyHostToDevice));
    if (shadow) {
        ALLOC_OR_CLEAN(d_s, n);
        ALLOC_OR_CLEAN(d_tm, n * 8);
        TRY_OR_CLEAN(hipMemcpy(d_s, shadow, n, hipMemcpyHostToDevice));
        TRY_OR_CLEAN(hipMemcpy(d_tm, tmax, n * 8, hipMemcpyHostToDevice));
    }
    rc = hpt_intersect_batch_dev(ctx, n, d_o, d_d, d_s, d_tm, d_p, d_t, 256, nullptr);
    if (rc == HPT_OK) {
        TRY_OR_CLEAN(hipDeviceSynchronize());
        TRY_OR_CLEAN(hipMemcpy(prim, d_p, n * 4, hipMemcpyDeviceToHost));
        TRY_OR_CLEAN(hipMemcpy(t, d_t, n * 8, hipMemcpyDeviceToHost));
    }
#undef TRY_OR_CLEAN
#undef ALLOC_OR_CLEAN
    cleanup();
    return rc;
}

int hpt_cast_rays_dev(hpt_ctx* ctx, size_t n, const float* o_dev, const float* d_dev, const uint32_t* pixel_key_dev,
                      const uint32_t* sample_key_dev, int max_depth, float* rgb_dev, int block, void* stream) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return HPT_OK;
    if (!o_dev || !d_dev || !pixel_key_dev || !sample_key_dev || !rgb_dev) return fail(HPT_ERR_INVALID_ARGUMENT, "null ray/key/result pointer");
    if (max_depth < 0 || max_depth > 16384) return fail(HPT_ERR_INVALID_ARGUMENT, "max_depth %d not in [0, 16384]", max_depth);
    if (block < 1 || block > 256) return fail(HPT_ERR_INVALID_ARGUMENT, "block size %d not in [1, 256]", block);
    if (int rc = set_device(ctx)) return rc;
    const DevScene& S = ctx->S;
    const uint32_t lds = lds_bytes(S, block);
    const int grid = grid_for(ctx, n, block);
    // the per-lane query counters are 32 bits wide
    const uint64_t rays_per_lane = (n + (uint64_t)grid * (uint64_t)block - 1u) / ((uint64_t)grid * (uint64_t)block);
    if (rays_per_lane * (uint64_t)(max_depth > 0 ? max_depth : 1) > 0xFFFFFFFFull)
        return fail(HPT_ERR_TOO_LARGE, "%zu rays of up to %d bounces are too many for one call: split the batch", n, max_depth);
    HPT_DISPATCH_RENDER(S.variant, ctx->extended, {
        hipLaunchKernelGGL((k_cast_rays<kS, kG>), dim3(grid), dim3(block), lds, (hipStream_t)stream, S, n, o_dev, d_dev, pixel_key_dev, sample_key_dev,
                           max_depth, rgb_dev, ctx->d_stats);
    })
    HIP_TRY(hipGetLastError());
    return HPT_OK;
}

int hpt_cast_rays(hpt_ctx* ctx, size_t n, const float* o, const float* d, const uint32_t* pixel_key, const uint32_t* sample_key,
                  int max_depth, float* rgb) {
    if (!ctx || !ctx->has_scene) return fail(HPT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return HPT_OK;
    if (!o || !d || !pixel_key || !sample_key || !rgb) return fail(HPT_ERR_INVALID_ARGUMENT, "null ray/key/result pointer");
    if (int rc = set_device(ctx)) return rc;
    // one staging block: origins, directions, radiance (3 floats each), the two keys
    float* stage = nullptr;
    if (int rc = device_alloc((void**)&stage, n * 44, "eye ray staging")) return rc;
    float *d_o = stage, *d_d = stage + 3 * n, *d_rgb = stage + 6 * n;
    uint32_t *d_pk = reinterpret_cast<uint32_t*>(stage + 9 * n), *d_sk = d_pk + n;
    int rc = HPT_OK;
    hipError_t e = hipMemcpy(d_o, o, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, d, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_pk, pixel_key, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sk, sample_key, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = hpt_cast_rays_dev(ctx, n, d_o, d_d, d_pk, d_sk, max_depth, d_rgb, 256, nullptr);
    if (e == hipSuccess && rc == HPT_OK) e = hipDeviceSynchronize();
    if (e == hipSuccess && rc == HPT_OK) e = hipMemcpy(rgb, d_rgb, n * 12, hipMemcpyDeviceToHost);
    (void)hipFree(stage);
    if (rc != HPT_OK) return rc;
    if (e != hipSuccess) return fail(HPT_ERR_HIP, "hpt_cast_rays: %s", hipGetErrorString(e));
    return HPT_OK;
}

int hpt_stats_get(hpt_ctx* ctx, hpt_stats* out) {
    if (!ctx || !out) return fail(HPT_ERR_INVALID_ARGUMENT, "null argument");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = settle_events(ctx, true)) return rc;
    unsigned long long h[kNumStats];
    HIP_TRY(hipMemcpy(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->closest_queries = h[kStatClosest];
    out->shadow_queries = h[kStatShadow];
    out->paths = h[kStatPaths];
    out->box_tests = h[kStatBox];
    out->tri_tests = h[kStatTri];
    out->kernel_ms = ctx->kernel_ms;
    out->launches = ctx->launches;
    out->variant = ctx->has_scene ? (uint32_t)ctx->S.variant : 0u;
#ifdef HPT_PROFILE
    if (ctx->d_prof) {
        unsigned long long pr[3 * kProfSections];
        HIP_TRY(hipMemcpy(pr, ctx->d_prof, sizeof pr, hipMemcpyDeviceToHost));
        const char* names[kProfSections] = {"regen", "box_closest", "tri_closest(+box)", "shadeA", "box_shadow", "tri_shadow(+box)", "shadeB", "wave_total"};
        for (int i = 0; i < kProfSections; ++i)
            fprintf(stderr, "PROF %-18s cycles %14llu  iters %12llu  avg_lanes %6.2f\n", names[i], pr[i], pr[kProfSections + i],
                    pr[kProfSections + i] ? (double)pr[2 * kProfSections + i] / (double)pr[kProfSections + i] : 0.0);
    }
#endif
    return HPT_OK;
}

int hpt_stats_reset(hpt_ctx* ctx) {
    if (!ctx) return fail(HPT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = settle_events(ctx, true)) return rc;
    HIP_TRY(hipMemset(ctx->d_stats, 0, kNumStats * sizeof(unsigned long long)));
    ctx->kernel_ms = 0.0;
    ctx->launches = 0;
    return HPT_OK;
}

}  // extern "C"

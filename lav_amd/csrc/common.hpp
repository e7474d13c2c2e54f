// Shared host/device helpers for liblav_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <thread>
#include <vector>

#include "lav_amd.h"

namespace lav {

inline char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// gfx950 hazard found in round 4 (tools/plan_stress.py, profiles/r04_plan_stress.txt): hipcc merges two adjacent 64-bit LDS stores
// into one ds_write2_b64 (128 bits of data read from the VGPR file over more than one cycle) and then schedules VALU instructions
// that OVERWRITE those data registers right behind it - its hazard recognizer only pads stores whose first data operand is wider than
// 64 bits (ds_write_b96 / b128).  Alone on its SIMD the wave gets away with it; with another kernel's waves issuing matrix
// instructions on the same SIMD the LDS instruction reads its operands late and stores the NEW register contents: the persistent plan
// kernel's gate pre-activations of every fourth row, whenever a 7x7 crop stem (tap-pair split kernel) shared its CUs.  Call this
// between LDS stores that the compiler could pair into 2 x 64 bits; tests/test_capi_host.py fails if a ds_write2_b64 is left in the
// library.
__device__ __forceinline__ void lds_store_fence() { asm volatile("" ::: "memory"); }
// The same stress test then showed the 64-bit form as well (ds_write_b64 of {r0, r1} with a v_pk_add_f32 into the same registers as
// the very next instruction) once the neighbour waves issue matrix instructions AND LDS traffic (the real stem kernel; the
// synthetic one of tools/probes/lds_hog.hip in mode 1): an LDS store waiting in a congested LDS queue reads its data registers
// when it gets there, not when it was issued.  A long-running kernel that shares CUs with such neighbours therefore COMMITS its LDS
// stores: wait until they have left (lgkmcnt 0), and keep every stored value pinned in its register until then -
//     s[i] = v; ...; lds_commit(); lds_keep(v); ...
// (lds_keep is an empty asm that "modifies" v, so the register allocator cannot hand v's register to anything else before it).
__device__ __forceinline__ void lds_commit() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ void lds_keep(T &v) { asm volatile("" : "+v"(v)); }
// What DOES make a victim immune (measured at the end of round 4, profiles/r04_plan_stress.txt): every LDS access issued together
// with its wait in ONE asm block, one access in flight at a time.  With -DLAV_PLAN_LDS_SYNC=1 the persistent plan kernel is
// bit-exact beside the synthetic matrix + LDS neighbour (0 of 400 launches, both polling schemes) and beside the real stem kernel
// with the LDS claims switched OFF (0 of 200; 126 of 150 wrong without), at +60 us per plan (frame 420 -> 410 frames/s).  The same
// accesses batched - several ds_read / ds_write in one asm block, all operands pinned, ONE wait - fail again (68 of 100): it is
// having more than one LDS operation of a wave in flight that goes wrong beside such neighbours, not registers being rewritten.
__device__ __forceinline__ float lds_read_sync(const float *p) {
    float v;
    const unsigned a = (unsigned)(size_t)p;   // (low half of the generic address = the LDS offset)
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_sync(float *p, float v) {
    const unsigned a = (unsigned)(size_t)p;
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}

// Host side of the same finding: the kernels that combine bf16 matrix instructions with heavy LDS traffic (split-operand
// convolutions, the ERFNet pair kernels) corrupt LDS-dependent results of kernels whose waves share their CUs (measured with the
// persistent plan kernel as the victim: wrong in 171 of 300 launches beside the brake net's split convolutions, 3-10 of 300 beside
// ERFNet, 0 of 300 once these kernels leave no LDS on their CUs; the fp32 tiled / direct kernels are harmless).  They therefore
// CLAIM the LDS of their CU: all of it when one workgroup runs per CU, half each when two do.  LAV_LDS_EXCLUSIVE=0: exact sizes.
inline size_t lds_claim(size_t need, size_t static_bytes = 0, bool needs_two_per_cu = false) {
    // LAV_LDS_EXCLUSIVE: 1 (default) as above | 0 exact sizes | 2 the whole CU also for the kernels that could run two per CU (a CU that
    // holds only ONE of their workgroups - grids below 2 x 256, launch tails - still has half its LDS free for a victim: the "rare
    // ERFNet aggressor" of profiles/r04_plan_stress.txt; tools/coresidency.py measures all three settings).  needs_two_per_cu: a
    // persistent launch whose workgroups must all be resident at once and outnumber the CUs keeps the half claim in every mode.
    static const int mode = [] { const char *e = getenv("LAV_LDS_EXCLUSIVE"); return e ? atoi(e) : 1; }();
    const size_t total = 160 * 1024;
    if (mode == 0) return need;
    if (need + static_bytes > total / 2 || (mode == 2 && !needs_two_per_cu && need + static_bytes > total / 3)) return (total - static_bytes) / 16 * 16;   // one workgroup per CU
    if (need + static_bytes > total / 3) return (total / 2 - static_bytes) / 16 * 16;   // two
    return need;                                                                          // three or more: left alone
}

#define LAV_HIP(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t lav_e_ = (expr);                                                                        \
        if (lav_e_ != hipSuccess)                                                                          \
            return lav::fail(LAV_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(lav_e_)); \
    } while (0)

#define LAV_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) return lav::fail(LAV_EINVAL, __VA_ARGS__);    \
    } while (0)

// launch errors are sticky until queried; call after every kernel launch
#define LAV_LAUNCH_CHECK() LAV_HIP(hipGetLastError())

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char *base;
    size_t cap, used;
    Arena(void *p, size_t bytes) : base(static_cast<char *>(p)), cap(bytes), used(0) {}
    template <typename T>
    T *take(size_t count) {
        used = align_up(used, 256);
        T *p = reinterpret_cast<T *>(base + used);
        used += count * sizeof(T);
        return p;
    }
    bool ok() const { return used <= cap; }
};

constexpr int WAVE = 64;

// host-side parallel loop (weight repacking: a training run that logs through the inference engines repacks every layer per step)
template <typename F>
inline void parallel_for(int n, F fn) {
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (n < 2 || nt < 2) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    nt = std::min<unsigned>(nt, (unsigned)n);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=] { for (int i = (int)t; i < n; i += (int)nt) fn(i); });
    for (auto &x : th) x.join();
}

// lav_batch_limit: device-resident row count honoured by the batch-aware launches of this thread (see lav_amd.h)
inline const int *&batch_limit() {
    static thread_local const int *p = nullptr;
    return p;
}

// Optional per-kernel HIP-event timers (lav_profile_enable / lav_profile_read in include/lav_amd.h).
// timer_begin returns a slot token (< 0 when profiling is off); both calls only record events on `st`.
int timer_begin(const char *name, hipStream_t st);
void timer_end(int token, hipStream_t st);

}  // namespace lav

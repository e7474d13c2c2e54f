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

// Finite-but-wrong results beside matrix + LDS heavy neighbours (round 4: profiles/r04_plan_stress.txt; resolved in round 5:
// profiles/r05_coresidency.md, tools/lds_hazard.py).  What round 4 saw: the persistent plan kernel (and, rarely, lav_crop_rotate)
// returned finite but wrong values whenever waves of a split-operand convolution, an ERFNet pair kernel or a synthetic bf16-matrix +
// LDS neighbour shared its CUs.  It was taken for an LDS effect (a ds_write2_b64 whose data registers hipcc rewrites right behind it;
// "more than one LDS operation in flight") and fenced with the helpers below and with LDS claims on the aggressors.  What it is:
// a packed fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) with an op_sel bit set - the low result takes the HIGH
// register of a 64-bit source pair - returns wrong values in lanes 48-63 (the instruction's last pass) while such a neighbour shares
// the SIMD.  One v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,0] in a loop, evaluated twice on identical inputs with no memory access
// at all, disagrees with itself 535 936 times in 1.26e10 beside ERFNet's 16-channel run and never alone; the same instruction without
// op_sel, every LDS pattern tried (store / load write-after-read on data and address registers, 8 loads in flight, partial waits,
// ds_write2_b64 with its data overwritten at once, ds_bpermute beside loads) and registers at rest never fail.  hipcc's SLP vectoriser
// emits the form freely: the old plan kernel held exactly one (its waypoint sum), lav_crop_rotate three.  The library is therefore
// built with -fno-slp-vectorize (lav_amd/build.py), writes the packed instructions it wants by hand with op_sel = 0 (deconv.hip), and
// tests/test_capi_host.py rejects any packed fp32 instruction with an op_sel bit in the disassembly.
//
// The round-4 helpers stay where they are used (they cost nothing): lds_store_fence keeps two 64-bit LDS stores from being merged into
// a ds_write2_b64 (tools/lds_hazard.py patterns 10 / 11 could not make that instruction fail, alone or beside any neighbour; the CPU
// test that forbids it is kept as a tripwire, not as a known hazard), lds_commit / lds_keep wait for a wave's LDS stores with the
// stored values pinned, lds_read_sync / lds_write_sync issue one LDS access at a time (-DLAV_PLAN_LDS_SYNC=1 builds of the old plan
// kernel: "immune" in round 4 because the changed code no longer contained the packed instruction).
__device__ __forceinline__ void lds_store_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void lds_commit() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ void lds_keep(T &v) { asm volatile("" : "+v"(v)); }
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

// LDS claims (round 4's fence, an opt-in since round 5): LAV_LDS_EXCLUSIVE=1 makes the split-operand convolutions and the ERFNet pair
// kernels ask for all of their CU's LDS (half each where two workgroups share a CU), so that no kernel that uses LDS runs beside them;
// 2: the whole CU also for kernels that could run two per CU; default 0: exact sizes.  needs_two_per_cu: a persistent launch whose
// workgroups must all be resident at once and outnumber the CUs keeps the half claim in every mode.
inline size_t lds_claim(size_t need, size_t static_bytes = 0, bool needs_two_per_cu = false) {
    static const int mode = [] { const char *e = getenv("LAV_LDS_EXCLUSIVE"); return e ? atoi(e) : 0; }();
    const size_t total = 160 * 1024;
    if (mode == 0) return need;
    if (need + static_bytes > total / 2 || (mode == 2 && !needs_two_per_cu && need + static_bytes > total / 3)) return (total - static_bytes) / 16 * 16;   // one workgroup per CU
    if (need + static_bytes > total / 3) return (total / 2 - static_bytes) / 16 * 16;   // two
    return need;                                                                          // three or more: left alone
}

// ---- LAV_CONV_F16X3's scale hand-off (conv_f16.hip, conv_wgrad.hip): maxima of the finite |values| of a tensor, in parts
// the largest finite |v| of the wave's lanes, in every lane
__device__ __forceinline__ float wave_finite_absmax(float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}
__device__ __forceinline__ float finite_abs(float v) {
    const float a = fabsf(v);
    return a <= 3.4028235e38f ? a : 0.f;
}
// max over parts[0 .. n) in every lane of the wave.  Every load of the pass is in flight at once: a loop of dependent-looking scalar
// loads paid one memory round trip per 64 parts (round 6, first version: 57 round trips for the feature map's 3648 parts - the head
// convolution got SLOWER than with its measuring launch).
__device__ __forceinline__ float parts_absmax(const float *__restrict__ parts, int n, int lane) {
    float m = 0.f;
    int i0 = 0;
    if ((reinterpret_cast<uintptr_t>(parts) & 15) == 0) {
        const float4 *p4 = reinterpret_cast<const float4 *>(parts);
        const int n4 = n >> 2;
        float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int j = lane; j < n4; j += 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[min(j + 64 * u, n4 - 1)];   // (clamped: a repeated part changes no maximum)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u].x = fmaxf(acc[u].x, v[u].x); acc[u].y = fmaxf(acc[u].y, v[u].y);
                acc[u].z = fmaxf(acc[u].z, v[u].z); acc[u].w = fmaxf(acc[u].w, v[u].w);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fmaxf(acc[u].x, acc[u].y), fmaxf(acc[u].z, acc[u].w)));
        i0 = n4 << 2;
    }
    for (int i = i0 + lane; i < n; i += 64) m = fmaxf(m, parts[i]);
    return wave_finite_absmax(m);
}
// the power of two that puts a tensor's largest finite magnitude m into [16384, 32768) - fp16 overflows at 65504; exponent floored at
// -100 (a tensor whose largest value is below 2^-100 - or subnormal - would give a subnormal scale and an infinite reciprocal)
__device__ __forceinline__ float f16_scale_of(float m) {
    int e = 0;
    (void)frexpf(m, &e);                           // m = f 2^e, f in [0.5, 1): m / 2^(e - 15) in [16384, 32768)
    return ldexpf(1.f, m > 0.f ? max(e, -100) - 15 : 0);
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

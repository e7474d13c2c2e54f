// ABI version / error string / device probe.
#include "common.hpp"

extern "C" int lav_abi_version(void) { return LAV_ABI_VERSION; }

extern "C" const char *lav_last_error(void) { return lav::error_buffer(); }

extern "C" int lav_batch_limit(const int *d_rows) {
    lav::batch_limit() = d_rows;
    return LAV_OK;
}

extern "C" int lav_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return lav::fail(LAV_EHIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    return n;
}

// ------------------------------------------------------------------------------------------------------
// HIP-event timers around selected kernels, for bench.py's live roofline figure.
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Timer {
    std::string name;
    std::vector<hipEvent_t> start, stop;
    long launches = 0;
};
std::mutex g_mu;
std::vector<Timer> g_timers;
int g_slots = 0;
}  // namespace

namespace lav {
int timer_begin(const char *name, hipStream_t st) {
    if (g_slots <= 0) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    int ti = -1;
    for (size_t i = 0; i < g_timers.size(); ++i)
        if (g_timers[i].name == name) ti = (int)i;
    if (ti < 0) {
        Timer t;
        t.name = name;
        t.start.resize(g_slots);
        t.stop.resize(g_slots);
        for (int i = 0; i < g_slots; ++i) {
            if (hipEventCreate(&t.start[i]) != hipSuccess || hipEventCreate(&t.stop[i]) != hipSuccess) return -1;
        }
        g_timers.push_back(t);
        ti = (int)g_timers.size() - 1;
    }
    Timer &t = g_timers[ti];
    if (t.launches >= g_slots) return -1;  // ring full: stop recording (keeps the first `slots` launches)
    const int slot = (int)t.launches;
    if (hipEventRecord(t.start[slot], st) != hipSuccess) return -1;
    return ti * 65536 + slot;
}
void timer_end(int token, hipStream_t st) {
    if (token < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Timer &t = g_timers[token / 65536];
    (void)hipEventRecord(t.stop[token % 65536], st);
    t.launches += 1;
}
}  // namespace lav

extern "C" int lav_profile_enable(int slots) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &t : g_timers) {
        for (auto e : t.start) (void)hipEventDestroy(e);
        for (auto e : t.stop) (void)hipEventDestroy(e);
    }
    g_timers.clear();
    g_slots = slots > 0 ? (slots < 65536 ? slots : 65535) : 0;
    return LAV_OK;
}

extern "C" int lav_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &t : g_timers) t.launches = 0;
    return LAV_OK;
}

extern "C" int lav_profile_read(const char *kernel, double *total_ms, int *launches) {
    LAV_REQUIRE(kernel && total_ms && launches, "lav_profile_read: null");
    std::lock_guard<std::mutex> lk(g_mu);
    *total_ms = 0;
    *launches = 0;
    for (auto &t : g_timers) {
        if (t.name != kernel) continue;
        for (long i = 0; i < t.launches; ++i) {
            LAV_HIP(hipEventSynchronize(t.stop[i]));
            float ms = 0.f;
            LAV_HIP(hipEventElapsedTime(&ms, t.start[i], t.stop[i]));
            *total_ms += ms;
        }
        *launches = (int)t.launches;
        return LAV_OK;
    }
    return LAV_OK;
}

// ------------------------------------------------------------------------------------------------------
// Small dense layer: out[b][o] = act(bias[o] + sum_k w[o][k] x[b][k]); one wave per output (rows of a few hundred to a
// few thousand values: a library GEMM launch costs more than the arithmetic - the brake classifier Linear(1024 -> 1) +
// sigmoid of team_code_v2/models/rgb.py:62,79 showed up as a 55 us hipBLASLt kernel on the frame's side stream).
namespace {
__global__ __launch_bounds__(256) void k_linear_act(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                    int B, int K, int O, int act, float *__restrict__ out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * O) return;
    const int b = wave / O, o = wave - b * O;
    const float *xr = x + (long)b * K, *wr = w + (long)o * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(wr[k], xr[k], acc);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) {
        float v = acc + (bias ? bias[o] : 0.f);
        if (act == 1) v = 1.f / (1.f + expf(-v));
        out[(long)b * O + o] = v;
    }
}
}  // namespace

extern "C" int lav_linear_act(const float *x, int batch, int in_features, const float *weight, const float *bias, int out_features,
                              int act, float *out, void *stream) {
    LAV_REQUIRE(x && weight && out && batch >= 0 && in_features >= 1 && out_features >= 1, "lav_linear_act: bad argument");
    LAV_REQUIRE(act == 0 || act == 1, "lav_linear_act: act %d (0 none, 1 sigmoid)", act);
    if (batch == 0) return LAV_OK;
    const long waves = (long)batch * out_features;
    hipLaunchKernelGGL(k_linear_act, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, weight, bias, batch,
                       in_features, out_features, act, out);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

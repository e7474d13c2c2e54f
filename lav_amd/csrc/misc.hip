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

// ------------------------------------------------------------------------------------------------------
// Glue of the frame graph that was library launches: 3x3 / stride-2 max pooling of the ResNet stems, the per-channel input
// normalisation of the brake net, and the copies of a tick's inputs into the graphs' static buffers (one launch for all).
namespace {
__global__ __launch_bounds__(256) void k_maxpool3s2(const float *__restrict__ x, int planes, int H, int W, int OH, int OW, float *__restrict__ y,
                                                    const int *__restrict__ n_valid, int planes_per_image) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)planes * OH * OW) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), p = (int)(i / ((long)OW * OH));
    if (n_valid && p / planes_per_image >= *n_valid) return;   // lav_batch_limit
    const float *xp = x + (long)p * H * W;
    float m = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int iy = 2 * oy + dy, ix = 2 * ox + dx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const float v = xp[iy * W + ix];
                m = (v > m || v != v) ? v : m;   // NaN-propagating like F.max_pool2d (fmaxf would drop it)
            }
        }
    y[i] = m;
}

__global__ __launch_bounds__(256) void k_channel_affine(const float *__restrict__ x, long plane, int channels, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, long total, float *__restrict__ y) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    // plane is a multiple of 4 (checked by the host): the four values share a channel
    const int c = (int)((i / plane) % channels);
    const float s = scale[c], t = shift[c];
    const float4 v = *reinterpret_cast<const float4 *>(x + i);
    *reinterpret_cast<float4 *>(y + i) = make_float4(v.x * s + t, v.y * s + t, v.z * s + t, v.w * s + t);
}

constexpr int COPY_MAX = 8;
struct CopyArgs {
    const char *src[COPY_MAX];
    char *dst[COPY_MAX];
    long end16[COPY_MAX];   // running total of 16-byte words (the last one of a copy may be partial) after copy i
    int tail[COPY_MAX];     // bytes in the last word of copy i (4, 8, 12 or 16)
    int n;
};
__global__ __launch_bounds__(256) void k_copy_many(CopyArgs a) {
    const long w = (long)blockIdx.x * 256 + threadIdx.x;
    int k = 0;
    while (k < a.n && w >= a.end16[k]) ++k;
    if (k >= a.n) return;
    const long off = (w - (k ? a.end16[k - 1] : 0)) * 16;
    if (w + 1 < a.end16[k] || a.tail[k] == 16) {
        *reinterpret_cast<float4 *>(a.dst[k] + off) = *reinterpret_cast<const float4 *>(a.src[k] + off);
    } else {
        for (int b = 0; b < a.tail[k]; b += 4) *reinterpret_cast<float *>(a.dst[k] + off + b) = *reinterpret_cast<const float *>(a.src[k] + off + b);
    }
}
// Strided (<= 4-D) float32 / uint8 sources -> contiguous float32 destinations, several tensors in one launch: the camera tensors
// the agent hands over are channels-last views (lav_agent_fast.py:252-277: stack / permute / float), the frame graphs read
// contiguous NCHW buffers.
constexpr int STAGE_MAX = 8;
constexpr int STAGE_BLOCK_WORDS = 64;
struct StageArgs {
    const void *src[STAGE_MAX];
    float *dst[STAGE_MAX];
    long end[STAGE_MAX];          // running total of elements after tensor i
    int dims[STAGE_MAX][4];       // sizes, outermost first (leading dims padded with 1)
    long strides[STAGE_MAX][4];   // source strides in elements
    int is_u8[STAGE_MAX];
    int n;
    // lav_stage_many_block: a small block of host data riding in the kernel arguments (the frame's 176 bytes of sweep indices and
    // poses: an upload of its own is one more packet in front of the frame's first graph)
    int block_words;
    unsigned *block_dst;
    unsigned block[STAGE_BLOCK_WORDS];
};
__global__ __launch_bounds__(256) void k_stage_many(StageArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && (int)threadIdx.x < a.block_words) a.block_dst[threadIdx.x] = a.block[threadIdx.x];
    int k = 0;
    while (k < a.n && i >= a.end[k]) ++k;
    if (k >= a.n) return;
    // (32-bit index arithmetic: a tensor has fewer than 2^31 elements - checked by the host -, and three 64-bit divisions per element
    // were most of this kernel's 12 us on the frame's 1.6 M camera values)
    unsigned r = (unsigned)(i - (k ? a.end[k - 1] : 0));
    const unsigned d3 = (unsigned)a.dims[k][3], d2 = (unsigned)a.dims[k][2], d1 = (unsigned)a.dims[k][1];
    const unsigned i3 = r % d3; r /= d3;
    const unsigned i2 = r % d2; r /= d2;
    const unsigned i1 = r % d1;
    const unsigned i0 = r / d1;
    const long off = (long)i0 * a.strides[k][0] + (long)i1 * a.strides[k][1] + (long)i2 * a.strides[k][2] + (long)i3 * a.strides[k][3];
    const float v = a.is_u8[k] ? (float)static_cast<const unsigned char *>(a.src[k])[off] : static_cast<const float *>(a.src[k])[off];
    a.dst[k][i - (k ? a.end[k - 1] : 0)] = v;
}
}  // namespace

extern "C" int lav_stage_many_block(int n, const void *const *src, float *const *dst, const int *dims, const long *strides, const int *src_is_u8,
                                    const void *block, int block_bytes, void *block_dst, void *stream) {
    LAV_REQUIRE(n >= 0 && n <= STAGE_MAX && (n == 0 || (src && dst && dims && strides && src_is_u8)), "lav_stage_many: at most %d tensors", STAGE_MAX);
    LAV_REQUIRE(block_bytes >= 0 && block_bytes <= 4 * STAGE_BLOCK_WORDS && block_bytes % 4 == 0 && (block_bytes == 0 || (block && block_dst)),
                "lav_stage_many_block: a block of at most %d bytes, a multiple of 4", 4 * STAGE_BLOCK_WORDS);
    if (n == 0 && block_bytes == 0) return LAV_OK;
    StageArgs a;
    a.block_words = block_bytes / 4;
    a.block_dst = static_cast<unsigned *>(block_dst);
    memset(a.block, 0, sizeof(a.block));
    if (block_bytes) memcpy(a.block, block, (size_t)block_bytes);
    long total = 0;
    for (int i = 0; i < STAGE_MAX; ++i) {
        if (i < n) {
            LAV_REQUIRE(src[i] && dst[i], "lav_stage_many: null tensor %d", i);
            long cnt = 1;
            for (int d = 0; d < 4; ++d) {
                LAV_REQUIRE(dims[4 * i + d] >= 1, "lav_stage_many: tensor %d has an empty dimension", i);
                a.dims[i][d] = dims[4 * i + d];
                a.strides[i][d] = strides[4 * i + d];
                cnt *= dims[4 * i + d];
            }
            LAV_REQUIRE(cnt < (1l << 31), "lav_stage_many: tensor %d has 2^31 elements or more", i);
            a.src[i] = src[i]; a.dst[i] = dst[i]; a.is_u8[i] = src_is_u8[i] ? 1 : 0;
            total += cnt;
        } else {
            a.src[i] = nullptr; a.dst[i] = nullptr; a.is_u8[i] = 0;
            for (int d = 0; d < 4; ++d) { a.dims[i][d] = 1; a.strides[i][d] = 0; }
        }
        a.end[i] = total;
    }
    a.n = n;
    hipLaunchKernelGGL(k_stage_many, dim3((unsigned)((total + 255) / 256 > 0 ? (total + 255) / 256 : 1)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_stage_many(int n, const void *const *src, float *const *dst, const int *dims, const long *strides, const int *src_is_u8,
                              void *stream) {
    return lav_stage_many_block(n, src, dst, dims, strides, src_is_u8, nullptr, 0, nullptr, stream);
}

extern "C" int lav_maxpool3x3s2(const float *x, int batch, int channels, int h, int w, float *y, void *stream) {
    LAV_REQUIRE(x && y && batch >= 0 && channels >= 1 && h >= 1 && w >= 1, "lav_maxpool3x3s2: bad argument");
    if (batch == 0) return LAV_OK;
    const int OH = (h - 1) / 2 + 1, OW = (w - 1) / 2 + 1;   // kernel 3, stride 2, padding 1
    const long total = (long)batch * channels * OH * OW;
    hipLaunchKernelGGL(k_maxpool3s2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, batch * channels, h, w,
                       OH, OW, y, lav::batch_limit(), channels);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_channel_affine(const float *x, int batch, int channels, long plane, const float *scale, const float *shift, float *y,
                                  void *stream) {
    LAV_REQUIRE(x && y && scale && shift && batch >= 0 && channels >= 1 && plane >= 4 && plane % 4 == 0, "lav_channel_affine: bad argument (plane must be a multiple of 4)");
    const long total = (long)batch * channels * plane;
    if (total == 0) return LAV_OK;
    hipLaunchKernelGGL(k_channel_affine, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, plane, channels,
                       scale, shift, total, y);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_copy_many(int n, const void *const *src, void *const *dst, const size_t *bytes, void *stream) {
    LAV_REQUIRE(n >= 0 && n <= COPY_MAX && (n == 0 || (src && dst && bytes)), "lav_copy_many: at most %d copies", COPY_MAX);
    if (n == 0) return LAV_OK;
    CopyArgs a;
    long total = 0;
    for (int i = 0; i < COPY_MAX; ++i) {
        if (i < n) {
            LAV_REQUIRE(src[i] && dst[i] && bytes[i] % 4 == 0 && ((size_t)src[i] | (size_t)dst[i]) % 16 == 0, "lav_copy_many: copy %d is not 16-byte aligned / a multiple of 4 bytes", i);
            a.src[i] = static_cast<const char *>(src[i]); a.dst[i] = static_cast<char *>(dst[i]);
            total += (long)((bytes[i] + 15) / 16);
            a.tail[i] = bytes[i] % 16 ? (int)(bytes[i] % 16) : 16;
        } else {
            a.src[i] = nullptr; a.dst[i] = nullptr; a.tail[i] = 16;
        }
        a.end16[i] = total;
    }
    a.n = n;
    if (total == 0) return LAV_OK;
    hipLaunchKernelGGL(k_copy_many, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

// ------------------------------------------------------------------------------------------------ health counter
// counter[0] += the number of the given tensors that hold a NaN / Inf, counter[1] += 1 (launches): a sticky, device-resident
// record that a frame's outputs were finite, enqueued at the end of the frame graphs - nothing waits for it, the host reads it
// whenever it likes (bench.py: after the timed region; the reference agent's own NaN rule, lav_agent_fast.py:325-328, is applied
// by the caller per frame as before).
namespace {
constexpr int FINITE_MAX = 8;
struct FiniteArgs {
    const float *p[FINITE_MAX];
    long n[FINITE_MAX];
    int count;
};
__global__ __launch_bounds__(256) void k_nonfinite_count(FiniteArgs a, int *__restrict__ counter) {
    __shared__ int bad_s;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    int bad = 0;
    for (int t = 0; t < a.count; ++t) {
        bool b = false;
        for (long i = threadIdx.x; i < a.n[t]; i += 256) {
            const float v = a.p[t][i];
            b = b || !(fabsf(v) <= 3.4028234664e38f);   // NaN or Inf
        }
        if (__any(b) && (threadIdx.x & 63) == 0) bad |= 1 << t;
    }
    if (bad) atomicOr(&bad_s, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (bad_s) atomicAdd(counter, __popc(bad_s));
        atomicAdd(counter + 1, 1);
    }
}
}  // namespace

extern "C" int lav_nonfinite_count(int n, const float *const *tensors, const long *numel, int *counter2, void *stream) {
    LAV_REQUIRE(n >= 0 && n <= FINITE_MAX && counter2 && (n == 0 || (tensors && numel)), "lav_nonfinite_count: at most %d tensors", FINITE_MAX);
    FiniteArgs a;
    a.count = n;
    for (int i = 0; i < FINITE_MAX; ++i) {
        a.p[i] = i < n ? tensors[i] : nullptr;
        a.n[i] = i < n ? numel[i] : 0;
        if (i < n) LAV_REQUIRE(tensors[i] != nullptr && numel[i] >= 0, "lav_nonfinite_count: tensor %d is null", i);
    }
    hipLaunchKernelGGL(k_nonfinite_count, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a, counter2);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

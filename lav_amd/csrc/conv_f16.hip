// LAV_CONV_F16X3 kernels of the split-operand convolution (round 5: the head convolution; round 6: every split plan) - a
// translation unit of their own (36 + 2 instantiations of split_body<..., F16 = true>: compile time).
//
// x = s (h0 + h1) with two fp16 pieces carries 22 mantissa bits and a . b ~ a0 b0 + a0 b1 + a1 b0 is THREE
// v_mfma_f32_32x32x16_f16 per 16 k-steps instead of six bf16 ones (conv_split_kernel.hpp).  The power-of-two scale s puts the
// tensor's largest finite magnitude into [16384, 32768).  Round 5 measured it with a launch of its own (k_absmax_parts) in front
// of the head convolution; round 6 lets every convolution leave the per-workgroup maxima of what it WRITES (amax_out), so the
// layers of a chain hand the scale on with no extra launch and no extra pass over the activations.
//
// Replaces (with the rest of lav_conv2d) the cuDNN convolutions of team_code_v2/models/lidar.py:48-161 and lav/models/resnet.py.
#include <cstdlib>

#include "common.hpp"
#include "conv_f16.hpp"

namespace {
using namespace lav;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAX_TAPS = 64;
constexpr int MAX_CLASSES = 16;
#include "conv_split_kernel.hpp"

template <int MP, int MC, int WPX, int NT, int G, bool TP>
__global__ __launch_bounds__(512) void k_conv_split_f16(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ks = blockIdx.z % a.ksplit;
    const int cls = (blockIdx.z / a.ksplit) % a.nclasses, n = blockIdx.z / (a.ksplit * a.nclasses);
    const int nchunks_k = TP ? 2 * a.nchunks : a.nchunks;
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    split_body<MP, MC, WPX, NT, G, TP, false, true>(a, smem_raw, blockIdx.x, blockIdx.y, cls, n, gridDim.z / (a.ksplit * a.nclasses), ks * nchunks_k / a.ksplit,
                                                    (ks + 1) * nchunks_k / a.ksplit, a.ksplit > 1 ? ks : -1, wg);
}

// workgroup g writes the largest FINITE |x| of its share of the layer's input channels to parts[g] (no atomics, nothing to zero);
// Inf / NaN inputs do not enter the scale and propagate through the data path as they are
__global__ __launch_bounds__(256) void k_absmax_parts(const float *__restrict__ x, int batch, int in_c_total, int in_c_offset, int cin, long plane,
                                                      float *__restrict__ parts, const int *__restrict__ n_valid) {
    __shared__ float s_m[4];
    if (n_valid) batch = min(batch, max(*n_valid, 0));
    const long per_img = (long)cin * plane, total = (long)batch * per_img;
    float m = 0.f;
    if ((plane & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long total4 = total >> 2, per4 = per_img >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const long n = i / per4, r = i - n * per4;
            const float4 v = *reinterpret_cast<const float4 *>(x + ((long)n * in_c_total + in_c_offset) * plane + 4 * r);
            m = fmaxf(fmaxf(m, finite_abs(v.x)), fmaxf(finite_abs(v.y), fmaxf(finite_abs(v.z), finite_abs(v.w))));
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const long n = i / per_img, r = i - n * per_img;
            m = fmaxf(m, finite_abs(x[((long)n * in_c_total + in_c_offset) * plane + r]));
        }
    }
    m = wave_finite_absmax(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

template <int MP, int MC, int WPX, int NT, int G, bool TP>
int launch_one(const SplitArgs &s, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_f16<MP, MC, WPX, NT, G, TP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((k_conv_split_f16<MP, MC, WPX, NT, G, TP>), grid, dim3(512), lds, st, s);
    return LAV_OK;
}
template <int MP, int MC, int WPX>
int launch_shape(const SplitArgs &s, bool small, int G, dim3 grid, size_t lds, hipStream_t st) {
    if (G == 4) return small ? launch_one<MP, MC, WPX, 2, 4, false>(s, grid, lds, st) : launch_one<MP, MC, WPX, SPLIT_NT, 4, false>(s, grid, lds, st);
    if (G == 2) return small ? launch_one<MP, MC, WPX, 2, 2, false>(s, grid, lds, st) : launch_one<MP, MC, WPX, SPLIT_NT, 2, false>(s, grid, lds, st);
    return small ? launch_one<MP, MC, WPX, 2, 1, false>(s, grid, lds, st) : launch_one<MP, MC, WPX, SPLIT_NT, 1, false>(s, grid, lds, st);
}
}  // namespace

namespace lav {
int launch_absmax_parts(const float *x, int batch, int in_c_total, int in_c_offset, int cin, long plane, float *parts, const int *n_valid, hipStream_t st) {
    hipLaunchKernelGGL(k_absmax_parts, dim3(F16_PARTS), dim3(256), 0, st, x, batch, in_c_total, in_c_offset, cin, plane, parts, n_valid);
    return LAV_OK;
}

int launch_split_f16(const void *split_args, size_t args_bytes, int mp, int mc, int wpx, int tp, unsigned gx, unsigned gy, unsigned gz, size_t lds, hipStream_t st) {
    if (args_bytes != sizeof(SplitArgs)) return fail(LAV_EINVAL, "launch_split_f16: argument block of %zu bytes, expected %zu", args_bytes, sizeof(SplitArgs));
    const SplitArgs &s = *static_cast<const SplitArgs *>(split_args);
    const dim3 grid(gx, gy, gz);
    const bool small = s.plane <= SPLIT_LOADERS * 2;
    const int G = s.tap_group;
    if (tp) {
        if (mp == 1 && mc == 2 && wpx == 4 && G == 4) return launch_one<1, 2, 4, SPLIT_NT_TP, 4, true>(s, grid, lds, st);
        if (mp == 2 && mc == 2 && wpx == 4 && G == 1) return launch_one<2, 2, 4, SPLIT_NT_TP, 1, true>(s, grid, lds, st);
        return fail(LAV_EINVAL, "lav_conv2d: fp16 tap-pair split tile %dx%d/%d G%d not built", mp, mc, wpx, G);
    }
    switch (mp * 100 + mc * 10 + wpx) {
        case 224: return launch_shape<2, 2, 4>(s, small, G, grid, lds, st);
        case 124: return launch_shape<1, 2, 4>(s, small, G, grid, lds, st);
        case 114: return launch_shape<1, 1, 4>(s, small, G, grid, lds, st);
        case 222: return launch_shape<2, 2, 2>(s, small, G, grid, lds, st);
        case 122: return launch_shape<1, 2, 2>(s, small, G, grid, lds, st);
        case 112: return launch_shape<1, 1, 2>(s, small, G, grid, lds, st);
    }
    return fail(LAV_EINVAL, "lav_conv2d: fp16 split tile %dx%d/%d not built", mp, mc, wpx);
}
}  // namespace lav

// Maxima of the finite |x| of a contiguous tensor of n floats in LAV_AMAX_PARTS parts (the measuring launch of LAV_CONV_F16X3 as an
// entry point of its own: a trainer measures an activation once per step and hands the parts to the forward convolution, the data
// gradient and the weight gradient that read it - lav_conv2d_amax, lav_conv_wgrad_amax).
extern "C" int lav_absmax_parts(const float *x, long n, float *parts, void *stream) {
    LAV_REQUIRE(x && parts && n >= 1, "lav_absmax_parts: bad argument");
    const int rc = lav::launch_absmax_parts(x, 1, 1, 0, 1, n, parts, nullptr, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

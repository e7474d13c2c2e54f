// Train-mode BatchNorm2d (batch statistics) with the neighbouring ReLU and the residual add fused, forward and backward,
// NCHW float32.  gfx950 only.
//
// The student networks of train_full_v2 / train_bev_v2 (lav/lav_final_v2.py:140-259) run three patterns:
//     Conv -> ReLU -> BatchNorm             ConvBackbone stages and up-convolutions (team_code_v2/models/lidar.py:57-108)
//     Conv -> BatchNorm -> ReLU             ResNet-18 conv1 / BasicBlock.conv1       (lav/models/resnet.py)
//     Conv -> BatchNorm -> (+ identity) -> ReLU     BasicBlock.conv2
// As torch ops they are 3-5 launches forward and as many backward, each a full pass over the activation; MIOpen's BatchNorm
// runs at 1.6-2 TB/s on the large maps and is launch bound (~130 us per call) on the small ones
// (profiles/r03_train_conv_probe.txt).  Here:
//
//   forward   k_bn_stats   per-channel sum / sum of squares of relu_pre(x), float64 accumulation, S slices per channel
//             k_bn_apply   y = relu_post((relu_pre(x) - mean) * rstd * gamma + beta + residual)      2 reads + 1 write of x
//   backward  k_bn_bwd_sums   g = dy * [y > 0]  (relu_post; also written out as the residual branch's gradient),
//                             per-channel sum g, sum g * xhat
//             k_bn_bwd_apply  dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) * [x > 0]  (relu_pre)
//
// Every kernel is HBM bound: algorithmic bytes = 4 * B*C*HW * (passes listed above).  Sums are order-fixed (slices are
// reduced in index order by every consumer), so the results are bit-reproducible run to run.
#include "common.hpp"

namespace {
using namespace lav;

constexpr int BN_MAX_SLICES = 64;

struct BnGeom {
    int B, C, S;     // batch, channels, slices per channel
    long HW, N;      // plane, B * HW
    long per;        // elements per slice (multiple of 4 when HW % 4 == 0)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// sum of (a, b) over the workgroup's 256 threads, result valid in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b, double *lds) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { lds[2 * wid] = a; lav::lds_store_fence(); lds[2 * wid + 1] = b; }   // (no ds_write2_b64: common.hpp)
    __syncthreads();
    if (threadIdx.x == 0) {
        a = lds[0] + lds[2] + lds[4] + lds[6];
        b = lds[1] + lds[3] + lds[5] + lds[7];
    }
}

// largest finite |v| over the workgroup's 256 threads -> parts[part] (round 6: the tensor's bound for LAV_CONV_F16X3 readers, lav_amd.h)
__device__ __forceinline__ void block_absmax_out(float m, float *__restrict__ parts, long part) {
    __shared__ float s_m[4];
    m = lav::wave_finite_absmax(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[part] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// visit the elements of slice (c, s): f(index into the NCHW tensor, number of valid floats 1..4)
template <bool VEC, typename F>
__device__ __forceinline__ void for_slice(const BnGeom &g, int c, int s, F f) {
    // (N < 2^31, checked by the host: 32-bit index arithmetic)
    const unsigned hw = (unsigned)g.HW, lo = (unsigned)s * (unsigned)g.per, hi = min(lo + (unsigned)g.per, (unsigned)g.N);
    constexpr unsigned STEP = VEC ? 4u : 1u;
    for (unsigned j = lo + threadIdx.x * STEP; j < hi; j += 256u * STEP) {
        const unsigned b = j / hw, i = j - b * hw;
        f(((long)b * g.C + c) * g.HW + i);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_bn_stats(BnGeom g, const float *__restrict__ x, int relu_pre, double *__restrict__ partial) {
    __shared__ double lds[8];
    const int c = blockIdx.x, s = blockIdx.y;
    double sum = 0.0, sq = 0.0;
    for_slice<VEC>(g, c, s, [&](long at) {
        if constexpr (VEC) {
            const float4 v = *reinterpret_cast<const float4 *>(x + at);
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double t = (double)(relu_pre ? fmaxf(e[k], 0.f) : e[k]);
                sum += t; sq += t * t;
            }
        } else {
            const double t = (double)(relu_pre ? fmaxf(x[at], 0.f) : x[at]);
            sum += t; sq += t * t;
        }
    });
    block_sum2(sum, sq, lds);
    if (threadIdx.x == 0) {
        partial[2 * ((long)c * g.S + s)] = sum;
        partial[2 * ((long)c * g.S + s) + 1] = sq;
    }
}

// the channel's two sums, slices added in index order (every workgroup of the channel gets the same bits)
__device__ __forceinline__ void channel_sums(const BnGeom &g, int c, const double *__restrict__ partial, double &a, double &b) {
    a = 0.0; b = 0.0;
    for (int s = 0; s < g.S; ++s) { a += partial[2 * ((long)c * g.S + s)]; b += partial[2 * ((long)c * g.S + s) + 1]; }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_bn_apply(BnGeom g, const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, double eps, int relu_pre,
                                                  int relu_post, const double *__restrict__ partial, float *__restrict__ save_mean,
                                                  float *__restrict__ save_var, float *__restrict__ save_rstd, float *__restrict__ amax) {
    const int c = blockIdx.x, s = blockIdx.y;
    double a, b;
    channel_sums(g, c, partial, a, b);
    const double mean_d = a / (double)g.N;
    const double var_d = fmax(b / (double)g.N - mean_d * mean_d, 0.0);   // biased, as the normalisation uses it
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + eps));
    if (s == 0 && threadIdx.x == 0) { save_mean[c] = mean; save_var[c] = (float)var_d; save_rstd[c] = rstd; }
    const float ga = gamma[c], be = beta[c];
    float am = 0.f;
    auto one = [&](float v, float r) {
        const float t = relu_pre ? fmaxf(v, 0.f) : v;
        float o = fmaf((t - mean) * rstd, ga, be) + r;
        o = relu_post ? fmaxf(o, 0.f) : o;
        am = fmaxf(am, lav::finite_abs(o));
        return o;
    };
    for_slice<VEC>(g, c, s, [&](long at) {
        if constexpr (VEC) {
            const float4 v = *reinterpret_cast<const float4 *>(x + at);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res) r = *reinterpret_cast<const float4 *>(res + at);
            *reinterpret_cast<float4 *>(y + at) = make_float4(one(v.x, r.x), one(v.y, r.y), one(v.z, r.z), one(v.w, r.w));
        } else {
            y[at] = one(x[at], res ? res[at] : 0.f);
        }
    });
    if (amax) block_absmax_out(am, amax, (long)c * g.S + s);   // (kernel-uniform)
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_bn_bwd_sums(BnGeom g, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                                                     const float *__restrict__ save_mean, const float *__restrict__ save_rstd, int relu_pre,
                                                     int relu_post, float *__restrict__ dres, double *__restrict__ partial) {
    __shared__ double lds[8];
    const int c = blockIdx.x, s = blockIdx.y;
    const float mean = save_mean[c], rstd = save_rstd[c];
    double sg = 0.0, sgx = 0.0;
    for_slice<VEC>(g, c, s, [&](long at) {
        if constexpr (VEC) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + at);
            float4 gv = *reinterpret_cast<const float4 *>(dy + at);
            if (relu_post) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + at);
                gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f; gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
            }
            if (dres) *reinterpret_cast<float4 *>(dres + at) = gv;
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = relu_pre ? fmaxf(xe[k], 0.f) : xe[k];
                sg += (double)ge[k];
                sgx += (double)ge[k] * (double)((t - mean) * rstd);
            }
        } else {
            float gg = dy[at];
            if (relu_post) gg = y[at] > 0.f ? gg : 0.f;
            if (dres) dres[at] = gg;
            const float t = relu_pre ? fmaxf(x[at], 0.f) : x[at];
            sg += (double)gg;
            sgx += (double)gg * (double)((t - mean) * rstd);
        }
    });
    block_sum2(sg, sgx, lds);
    if (threadIdx.x == 0) {
        partial[2 * ((long)c * g.S + s)] = sg;
        partial[2 * ((long)c * g.S + s) + 1] = sgx;
    }
}

// g_src: the gradient after the relu_post mask (dres when the forward had a residual, else dy with the mask applied here)
template <bool VEC>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnGeom g, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ g_src,
                                                      int mask_here, const float *__restrict__ gamma, const float *__restrict__ save_mean,
                                                      const float *__restrict__ save_rstd, int relu_pre, const double *__restrict__ partial,
                                                      float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                      float *__restrict__ amax) {
    const int c = blockIdx.x, s = blockIdx.y;
    double a, b;
    channel_sums(g, c, partial, a, b);
    if (s == 0 && threadIdx.x == 0) { dbeta[c] = (float)a; dgamma[c] = (float)b; }
    const float mean = save_mean[c], rstd = save_rstd[c];
    const float k0 = gamma[c] * rstd, mg = (float)(a / (double)g.N), mgx = (float)(b / (double)g.N);
    float am = 0.f;
    auto one = [&](float xv, float gv, float yv) {
        if (mask_here) gv = yv > 0.f ? gv : 0.f;
        const float t = relu_pre ? fmaxf(xv, 0.f) : xv;
        float o = k0 * (gv - mg - (t - mean) * rstd * mgx);
        o = (relu_pre && !(xv > 0.f)) ? 0.f : o;
        am = fmaxf(am, lav::finite_abs(o));
        return o;
    };
    for_slice<VEC>(g, c, s, [&](long at) {
        if constexpr (VEC) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + at);
            const float4 gv = *reinterpret_cast<const float4 *>(g_src + at);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (mask_here) yv = *reinterpret_cast<const float4 *>(y + at);
            *reinterpret_cast<float4 *>(dx + at) = make_float4(one(xv.x, gv.x, yv.x), one(xv.y, gv.y, yv.y), one(xv.z, gv.z, yv.z), one(xv.w, gv.w, yv.w));
        } else {
            dx[at] = one(x[at], g_src[at], mask_here ? y[at] : 1.f);
        }
    });
    if (amax) block_absmax_out(am, amax, (long)c * g.S + s);
}

int geometry(BnGeom &g, int batch, int channels, long hw, const char *who) {
    LAV_REQUIRE(batch >= 1 && channels >= 1 && channels <= 65535 && hw >= 1, "%s: bad shape", who);
    g.B = batch; g.C = channels; g.HW = hw; g.N = (long)batch * hw;
    LAV_REQUIRE(g.N >= 2, "%s: batch statistics need more than one value per channel", who);
    LAV_REQUIRE(g.N < (1l << 31) - 4096, "%s: more than 2^31 values per channel", who);
    // slices: about 2048 workgroups over the tensor, at least 8192 values each
    long want = std::max<long>(1, 2048 / channels);
    want = std::min<long>(want, (g.N + 8191) / 8192);
    g.S = (int)std::min<long>(std::max<long>(want, 1), BN_MAX_SLICES);
    g.per = ((g.N + g.S - 1) / g.S + 3) / 4 * 4;
    g.S = (int)((g.N + g.per - 1) / g.per);
    return LAV_OK;
}

bool vec_ok(long hw, std::initializer_list<const void *> ptrs) {
    if (hw % 4) return false;
    for (const void *p : ptrs) if (p && reinterpret_cast<uintptr_t>(p) % 16) return false;
    return true;
}
}  // namespace

extern "C" size_t lav_bn_train_workspace_bytes(int channels) { return (size_t)std::max(channels, 1) * BN_MAX_SLICES * 2 * sizeof(double); }

extern "C" int lav_bn_train_amax_count(int batch, int channels, long hw) {
    BnGeom g;
    if (batch < 1 || channels < 1 || hw < 1 || (long)batch * hw < 2 || geometry(g, batch, channels, hw, "lav_bn_train_amax_count")) return 0;
    return g.C * g.S;
}

extern "C" int lav_bn_train_forward(const float *x, const float *residual, float *y, int batch, int channels, long hw, const float *gamma,
                                    const float *beta, double eps, int relu_pre, int relu_post, float *save_mean, float *save_var,
                                    float *save_rstd, void *workspace, size_t workspace_bytes, void *stream) {
    return lav_bn_train_forward_amax(x, residual, y, batch, channels, hw, gamma, beta, eps, relu_pre, relu_post, save_mean, save_var, save_rstd,
                                     nullptr, workspace, workspace_bytes, stream);
}

extern "C" int lav_bn_train_forward_amax(const float *x, const float *residual, float *y, int batch, int channels, long hw, const float *gamma,
                                         const float *beta, double eps, int relu_pre, int relu_post, float *save_mean, float *save_var,
                                         float *save_rstd, float *amax_y, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(x && y && gamma && beta && save_mean && save_var && save_rstd && workspace, "lav_bn_train_forward: null pointer");
    LAV_REQUIRE(workspace_bytes >= lav_bn_train_workspace_bytes(channels), "lav_bn_train_forward: workspace smaller than lav_bn_train_workspace_bytes");
    LAV_REQUIRE(!(relu_pre && (relu_post || residual)), "lav_bn_train_forward: relu_pre excludes relu_post / residual");
    BnGeom g;
    if (int rc = geometry(g, batch, channels, hw, "lav_bn_train_forward")) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(workspace);
    const dim3 grid((unsigned)g.C, (unsigned)g.S);
    const int tok = timer_begin("bn_train_fwd", st);
    if (vec_ok(hw, {x, y, residual})) {
        hipLaunchKernelGGL(k_bn_stats<true>, grid, dim3(256), 0, st, g, x, relu_pre, partial);
        hipLaunchKernelGGL(k_bn_apply<true>, grid, dim3(256), 0, st, g, x, residual, y, gamma, beta, eps, relu_pre, relu_post, partial, save_mean, save_var, save_rstd, amax_y);
    } else {
        hipLaunchKernelGGL(k_bn_stats<false>, grid, dim3(256), 0, st, g, x, relu_pre, partial);
        hipLaunchKernelGGL(k_bn_apply<false>, grid, dim3(256), 0, st, g, x, residual, y, gamma, beta, eps, relu_pre, relu_post, partial, save_mean, save_var, save_rstd, amax_y);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_bn_train_backward(const float *x, const float *y, const float *dy, int batch, int channels, long hw, const float *gamma,
                                     const float *save_mean, const float *save_rstd, int relu_pre, int relu_post, float *dx, float *dres,
                                     float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream) {
    return lav_bn_train_backward_amax(x, y, dy, batch, channels, hw, gamma, save_mean, save_rstd, relu_pre, relu_post, dx, dres, dgamma, dbeta, nullptr,
                                      workspace, workspace_bytes, stream);
}

extern "C" int lav_bn_train_backward_amax(const float *x, const float *y, const float *dy, int batch, int channels, long hw, const float *gamma,
                                          const float *save_mean, const float *save_rstd, int relu_pre, int relu_post, float *dx, float *dres,
                                          float *dgamma, float *dbeta, float *amax_dx, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(x && dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta && workspace, "lav_bn_train_backward: null pointer");
    LAV_REQUIRE(!relu_post || y, "lav_bn_train_backward: relu_post needs the forward output y");
    LAV_REQUIRE(workspace_bytes >= lav_bn_train_workspace_bytes(channels), "lav_bn_train_backward: workspace smaller than lav_bn_train_workspace_bytes");
    LAV_REQUIRE(!(relu_pre && (relu_post || dres)), "lav_bn_train_backward: relu_pre excludes relu_post / residual");
    BnGeom g;
    if (int rc = geometry(g, batch, channels, hw, "lav_bn_train_backward")) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(workspace);
    const dim3 grid((unsigned)g.C, (unsigned)g.S);
    const float *g_src = dres ? dres : dy;
    const int mask_here = relu_post && !dres;
    const int tok = timer_begin("bn_train_bwd", st);
    if (vec_ok(hw, {x, y, dy, dx, dres})) {
        hipLaunchKernelGGL(k_bn_bwd_sums<true>, grid, dim3(256), 0, st, g, x, y, dy, save_mean, save_rstd, relu_pre, relu_post, dres, partial);
        hipLaunchKernelGGL(k_bn_bwd_apply<true>, grid, dim3(256), 0, st, g, x, y, g_src, mask_here, gamma, save_mean, save_rstd, relu_pre, partial, dx, dgamma, dbeta, amax_dx);
    } else {
        hipLaunchKernelGGL(k_bn_bwd_sums<false>, grid, dim3(256), 0, st, g, x, y, dy, save_mean, save_rstd, relu_pre, relu_post, dres, partial);
        hipLaunchKernelGGL(k_bn_bwd_apply<false>, grid, dim3(256), 0, st, g, x, y, g_src, mask_here, gamma, save_mean, save_rstd, relu_pre, partial, dx, dgamma, dbeta, amax_dx);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

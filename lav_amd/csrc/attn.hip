// Single-query multi-head attention pooling of the brake net (lav/models/attention.py:21-38, called from
// team_code_v2/models/rgb.py:69-70).  gfx950 only.
//
// The module has ONE learned query per head, so both projections fold around it:
//   dots[h][n] = scale * q_h . (W_k,h x_n + b_k,h + PE[n])  =  u_h . x_n + bias[h][n]
//                u_h = scale * W_k,h^T q_h  (heads x C),   bias[h][n] = scale * q_h . (b_k,h + PE[n])
//   out_h      = sum_n softmax(dots[h])[n] (W_v,h x_n + b_v,h)  =  W_v,h (sum_n p[h][n] x_n) + b_v,h
// i.e. two reductions over the map and one 64 x 512 matrix-vector product per head instead of the (N x C) x (C x 2C)
// key/value GEMM: 2 MMAC instead of 113 at the wide view's 216 tokens.  u and bias depend on the weights only and are
// prepared once on the host (lav_amd/rgb.py).  HBM-light (the 442 KB map is read twice from L2, W_v once); one
// 256-thread workgroup per (head, image): (a) dots, one token per thread; (b) soft-max over the tokens through LDS;
// (c) the pooled map, one channel per wave at a time; (d) the head's 64 outputs, four lanes per output row of W_v.
#include "common.hpp"

namespace {
using namespace lav;

constexpr int ATT_THREADS = 256;
constexpr int ATT_MAX_TOKENS = 4096;   // tokens of one map (LDS: probabilities)
constexpr int ATT_MAX_C = 1024;        // channels (LDS: pooled map of one head)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ __launch_bounds__(ATT_THREADS) void k_attn_pool(const float *__restrict__ x, int C, int N, int heads,
                                                           const float *__restrict__ u, const float *__restrict__ bias,
                                                           const float *__restrict__ w_v, const float *__restrict__ b_v,
                                                           float *__restrict__ out) {
    __shared__ float prob[ATT_MAX_TOKENS];
    __shared__ float xbar[ATT_MAX_C];
    __shared__ float red[8];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *xb = x + (long)b * C * N;
    const float *uh = u + (long)h * C;
    // (a) dots: consecutive threads read consecutive tokens of one channel plane; u_h[c] is wave-uniform (scalar loads)
    float mx = -INFINITY;
    for (int n = tid; n < N; n += ATT_THREADS) {
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) acc = fmaf(uh[c], xb[(long)c * N + n], acc);
        acc += bias[(long)h * N + n];
        prob[n] = acc;
        mx = fmaxf(mx, acc);
    }
    // (b) soft-max over the tokens
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int n = tid; n < N; n += ATT_THREADS) {
        const float e = expf(prob[n] - mx);
        prob[n] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    // (c) pooled map of this head: xbar[c] = sum_n p[n] x[c][n]
    for (int c = wid; c < C; c += ATT_THREADS / 64) {
        float acc = 0.f;
        for (int n = lane; n < N; n += 64) acc = fmaf(prob[n], xb[(long)c * N + n], acc);
        acc = wave_sum(acc);
        if (lane == 0) xbar[c] = acc * inv;
    }
    __syncthreads();
    // (d) the head's outputs: row h*dh + d of W_v against xbar, four lanes per row
    const int dh = C / heads;
    for (int d0 = 0; d0 < dh; d0 += ATT_THREADS / 4) {
        const int d = d0 + (tid >> 2), seg = tid & 3;
        float acc = 0.f;
        if (d < dh) {
            const float *wr = w_v + (long)(h * dh + d) * C;
            for (int c = seg * 4; c < C; c += 16) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wr + c);
                acc = fmaf(w4.x, xbar[c], acc);
                acc = fmaf(w4.y, xbar[c + 1], acc);
                acc = fmaf(w4.z, xbar[c + 2], acc);
                acc = fmaf(w4.w, xbar[c + 3], acc);
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (d < dh && seg == 0) out[(long)b * C + h * dh + d] = acc + b_v[h * dh + d];
    }
}
}  // namespace

extern "C" int lav_attn_pool(const float *x, int batch, int C, int N, int heads, const float *u, const float *dots_bias,
                             const float *w_v, const float *b_v, float *out, void *stream) {
    LAV_REQUIRE(x && u && dots_bias && w_v && b_v && out, "lav_attn_pool: null argument");
    LAV_REQUIRE(batch >= 1 && batch <= 65535 && heads >= 1 && heads <= 65535, "lav_attn_pool: bad batch / heads");
    LAV_REQUIRE(C >= heads && C % heads == 0 && C % 16 == 0 && C <= ATT_MAX_C, "lav_attn_pool: channels %d unsupported (multiple of 16 and of the heads, <= %d)", C, ATT_MAX_C);
    LAV_REQUIRE(N >= 1 && N <= ATT_MAX_TOKENS, "lav_attn_pool: %d tokens unsupported (<= %d)", N, ATT_MAX_TOKENS);
    hipLaunchKernelGGL(k_attn_pool, dim3(heads, batch), dim3(ATT_THREADS), 0, static_cast<hipStream_t>(stream), x, C, N, heads, u,
                       dots_bias, w_v, b_v, out);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

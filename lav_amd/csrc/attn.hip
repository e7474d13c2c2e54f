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
// 1024-thread workgroup per (head, image): (a) dots (waves split channels x tokens); (b) soft-max over the tokens through LDS;
// (c) the pooled map, two threads per channel; (d) the head's 64 outputs, sixteen lanes per output row of W_v.
#include "common.hpp"

namespace {
using namespace lav;

constexpr int ATT_THREADS = 1024;      // 16 waves: every phase is a latency chain of L2 loads, so it is cut 16 ways
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

// One workgroup per (head, image).
__global__ __launch_bounds__(ATT_THREADS) void k_attn_pool(const float *__restrict__ x, int C, int N, int heads,
                                                           const float *__restrict__ u, const float *__restrict__ bias,
                                                           const float *__restrict__ w_v, const float *__restrict__ b_v,
                                                           float *__restrict__ out) {
    __shared__ float prob[ATT_MAX_TOKENS];
    __shared__ float part[4][ATT_MAX_TOKENS];   // (a): partial dots of the four channel quarters
    __shared__ float xbar[ATT_MAX_C];
    __shared__ float red[32];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *xb = x + (long)b * C * N;
    const float *uh = u + (long)h * C;
    // (a) dots.  Wave (cq, tq): channel quarter cq = wid & 3, tokens tq*64 + lane + 256 i.  Consecutive lanes read
    // consecutive tokens of one channel plane; u_h[c] is wave-uniform (scalar loads); 16 loads in flight per lane.
    {
        const int cq = wid & 3, tq = wid >> 2;
        const int cpq = (C + 3) / 4, c_lo = cq * cpq, c_hi = min(C, c_lo + cpq);
        for (int n = tq * 64 + lane; n < N; n += 256) {
            float acc = 0.f;
#pragma unroll 16
            for (int c = c_lo; c < c_hi; ++c) acc = fmaf(uh[c], xb[(long)c * N + n], acc);
            part[cq][n] = acc;
        }
    }
    __syncthreads();
    // (b) soft-max over the tokens
    float mx = -INFINITY;
    for (int n = tid; n < N; n += ATT_THREADS) {
        const float d = ((part[0][n] + part[1][n]) + (part[2][n] + part[3][n])) + bias[(long)h * N + n];
        prob[n] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < ATT_THREADS / 64; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int n = tid; n < N; n += ATT_THREADS) {
        const float e = expf(prob[n] - mx);
        prob[n] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[16 + wid] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_THREADS / 64; ++w) tot += red[16 + w];
    const float inv = 1.f / tot;
    // (c) pooled map of this head: xbar[c] = sum_n p[n] x[c][n]; two threads per channel walk its token row (L1 lines are
    // consumed whole over the iterations), p[n] is an LDS broadcast
    for (int c0 = 0; c0 < C; c0 += ATT_THREADS / 2) {
        const int c = c0 + (tid >> 1), halfn = tid & 1;
        float acc = 0.f;
        if (c < C) {
            const float *xr = xb + (long)c * N;
            const int n_mid = (N / 2) & ~3, n_lo = halfn ? n_mid : 0, n_hi = halfn ? N : n_mid;
            int n = n_lo;
            if ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) {
#pragma unroll 8
                for (; n + 4 <= n_hi; n += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(xr + n);
                    acc = fmaf(prob[n], v.x, acc);
                    acc = fmaf(prob[n + 1], v.y, acc);
                    acc = fmaf(prob[n + 2], v.z, acc);
                    acc = fmaf(prob[n + 3], v.w, acc);
                }
            }
            for (; n < n_hi; ++n) acc = fmaf(prob[n], xr[n], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        if (c < C && halfn == 0) xbar[c] = acc * inv;
    }
    __syncthreads();
    // (d) the head's outputs: row h*dh + d of W_v against xbar, sixteen lanes per row (each a contiguous 1/16 of the row)
    const int dh = C / heads;
    for (int d0 = 0; d0 < dh; d0 += ATT_THREADS / 16) {
        const int d = d0 + (tid >> 4), seg = tid & 15;
        float acc = 0.f;
        if (d < dh) {
            const float *wr = w_v + (long)(h * dh + d) * C;
            for (int c = seg * 4; c < C; c += 64) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wr + c);
                acc = fmaf(w4.x, xbar[c], acc);
                acc = fmaf(w4.y, xbar[c + 1], acc);
                acc = fmaf(w4.z, xbar[c + 2], acc);
                acc = fmaf(w4.w, xbar[c + 3], acc);
            }
        }
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) acc += __shfl_xor(acc, sft, 64);
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

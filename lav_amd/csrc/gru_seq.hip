// GRU over a sequence with a saved tape, forward and backward: the training-side recurrent layer of the planners
// (team_code_v2/models/uniplanner.py:255-308 in train mode - `cast` and `_plan` call nn.GRU, which on ROCm is MIOpen's RNN:
// per call ~100 launches of small GEMMs and tensor ops forward and as many backward).  gfx950 only.
//
// The layer is split where the sequential dependency is:
//   * the input projection x_t = W_ih u_t + b_ih has none - it is one GEMM over all (row, t) and stays on the caller's
//     side (torch/rocBLAS, differentiable there); likewise the weight gradients, which are two GEMMs over the tape;
//   * the recurrence is what this file runs: ONE launch per time step, the recurrent GEMM fused with the gates.
//
// forward, step t   : gh = h_{t-1} W_hh^T + b_hh;  r = s(x_r + gh_r), z = s(x_z + gh_z), n = tanh(x_n + r gh_n);
//                     h_t = (1 - z) n + z h_{t-1};  tape_t = (r, z, n, gh_n)
// backward, step t-1: dh_{t-1} = dOut_{t-1} + dh_t z_t + dgh_t W_hh;  with the tape of t-1:
//                     da_n = dh (1 - z)(1 - n^2), da_z = dh (h_{t-2} - n) z (1 - z), da_r = da_n gh_n r (1 - r)
//                     dx_{t-1} = (da_r, da_z, da_n),  dgh_{t-1} = (da_r, da_z, da_n r)   [T + 1 launches: first = no GEMM,
//                     last = only dh_0]
//
// Tiling (both directions): a workgroup owns 16 rows x 16 hidden units; its four waves split the reduction dimension
// (H forward, 3H backward) and meet in LDS, so a step is R/16 x H/16 workgroups of ~100 v_mfma_f32_16x16x4_f32 per wave -
// at the trainer's sizes (R = 192-400 rows, H = 512) 384+ workgroups, a few microseconds.  Operands come straight from
// L2 (W_hh is 3 MB, the state a few hundred KB): sixteen-byte loads per lane, k assigned to lanes so that both fragments
// are contiguous (lane (m, kg) holds k = kb + 4 kg + i for the i-th of four consecutive matrix instructions).
// fp32 throughout; precise expf / tanhf.
#include "common.hpp"
#include "gru_seq.hpp"

namespace {
using namespace lav;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// acc[i] += A[16 rows][k_lo, k_hi) . B_i[16 cols][k_lo, k_hi)^T for NB matrices B_i that share A.  a_row / b_row[i]: this
// lane's row of A (m = lane & 15) and of B_i (n = lane & 15), both k-contiguous; the k range is a multiple of 16.
// The operands of BATCH 16-blocks are requested together (L2 latency is the cost of this loop, not the matrix pipe).
template <int NB, int BATCH>
__device__ __forceinline__ void dot_tiles(const float *__restrict__ a_row, const float *const (&b_row)[NB], int k_lo, int k_hi, int kg,
                                          f32x4 (&acc)[NB]) {
    for (int kb = k_lo; kb < k_hi; kb += 16 * BATCH) {
        float4 a4[BATCH], b4[NB][BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int k = min(kb + 16 * i, k_hi - 16) + 4 * kg;   // blocks past the end re-read the last one and are zeroed below
            a4[i] = *reinterpret_cast<const float4 *>(a_row + k);
#pragma unroll
            for (int q = 0; q < NB; ++q) b4[q][i] = *reinterpret_cast<const float4 *>(b_row[q] + k);
        }
        __builtin_amdgcn_sched_barrier(0);   // all requests go out before the first matrix instruction (the scheduler would sink them)
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (kb + 16 * i >= k_hi) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].x, b4[q][i].x, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].y, b4[q][i].y, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].z, b4[q][i].z, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].w, b4[q][i].w, acc[q], 0, 0, 0);
            }
        }
    }
}

using FwdArgs = lav::GruFwdArgs;   // gru_seq.hpp

__global__ __launch_bounds__(256) void k_gru_fwd_step(FwdArgs a) {
    __shared__ float part[4][3][256];   // [wave][gate][lane * 4 + reg]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
    const int m = lane & 15, kg = lane >> 4;
    const int H = a.H;
    const float *a_row = a.h_prev + (long)min(r0 + m, a.R - 1) * a.h_prev_stride;
    // this wave's quarter of the reduction, in whole 16-blocks
    const int blocks = H / 16, per = (blocks + 3) / 4;
    const int k_lo = min(wid * per, blocks) * 16, k_hi = min((wid + 1) * per, blocks) * 16;
    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const float *const b_row[3] = {a.w_hh + ((long)0 * H + j0 + m) * H, a.w_hh + ((long)1 * H + j0 + m) * H, a.w_hh + ((long)2 * H + j0 + m) * H};
    if (k_lo < k_hi) dot_tiles<3, 4>(a_row, b_row, k_lo, k_hi, kg, acc);
#pragma unroll
    for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4 *>(&part[wid][g][lane * 4]) = acc[g];
    __syncthreads();
    // thread (row = tid >> 4, unit = tid & 15): D[row][unit] sits in lane unit + 16 (row >> 2), register row & 3
    const int row = tid >> 4, n = tid & 15, src = (n + 16 * (row >> 2)) * 4 + (row & 3);
    const int r = r0 + row, j = j0 + n;
    if (r >= a.R) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gh[g] = ((part[0][g][src] + part[1][g][src]) + (part[2][g][src] + part[3][g][src])) + a.b_hh[g * H + j];
    float xg[3];
    if (a.u) {   // narrow input projected here: x_g = W_ih[g H + j] . u_r + b_ih[g H + j]
        const float *ur = a.u + (long)r * a.u_stride;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float *wr = a.w_ih + (long)(g * H + j) * a.I;
            float s = 0.f;
            for (int k = 0; k < a.I; ++k) s = fmaf(wr[k], ur[k], s);
            xg[g] = s + a.b_ih[g * H + j];
        }
    } else {
        const float *xr = a.x + (long)r * a.x_stride;
#pragma unroll
        for (int g = 0; g < 3; ++g) xg[g] = xr[g * H + j];
    }
    const float hp = a.h_prev[(long)r * a.h_prev_stride + j];
    const float rg = sigm(xg[0] + gh[0]);
    const float zg = sigm(xg[1] + gh[1]);
    const float ng = tanhf(xg[2] + rg * gh[2]);
    a.h_out[(long)r * a.h_out_stride + j] = (1.f - zg) * ng + zg * hp;
    if (a.tape) {
        float *tp = a.tape + (long)r * a.tape_stride;
        tp[j] = rg;
        tp[H + j] = zg;
        tp[2 * H + j] = ng;
        tp[3 * H + j] = gh[2];
    }
}

struct BwdArgs {
    const float *dgh;      // dgh_t: row r at dgh + r * dgh_stride, [3H]                (null: first launch, no GEMM)
    const float *w_hh_t;   // W_hh transposed: [H][3H]
    const float *dhz;      // dh_t * z_t  [R][H]                                         (null on the first launch)
    const float *dout;     // dOut[:, t-1]                                               (null on the last launch)
    const float *tape;     // tape of step t-1                                           (null on the last launch)
    const float *h_prev2;  // h_{t-2} (h0 for t-1 = 0)
    float *dx;             // dx_{t-1}: row r at dx + r * dx_stride, [3H]
    float *dgh_out;        // dgh_{t-1}: same layout as dgh
    float *dhz_out;        // dh_{t-1} * z_{t-1}  [R][H]
    float *dh0;            // last launch: gradient of the initial state [R][H]
    long dgh_stride, dout_stride, tape_stride, h_prev2_stride, dx_stride;
    int R, H;
};

__global__ __launch_bounds__(256) void k_gru_bwd_step(BwdArgs a) {
    __shared__ float part[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
    const int m = lane & 15, kg = lane >> 4;
    const int H = a.H, K = 3 * H;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (a.dgh) {
        const int blocks = K / 16, per = (blocks + 3) / 4;
        const int k_lo = min(wid * per, blocks) * 16, k_hi = min((wid + 1) * per, blocks) * 16;
        const float *const b_row[1] = {a.w_hh_t + (long)(j0 + m) * K};
        if (k_lo < k_hi) dot_tiles<1, 8>(a.dgh + (long)min(r0 + m, a.R - 1) * a.dgh_stride, b_row, k_lo, k_hi, kg, acc);
    }
    *reinterpret_cast<f32x4 *>(&part[wid][lane * 4]) = acc[0];
    __syncthreads();
    const int row = tid >> 4, n = tid & 15, src = (n + 16 * (row >> 2)) * 4 + (row & 3);
    const int r = r0 + row, j = j0 + n;
    if (r >= a.R) return;
    float dh = (part[0][src] + part[1][src]) + (part[2][src] + part[3][src]);
    if (a.dhz) dh += a.dhz[(long)r * H + j];
    if (!a.tape) {   // last launch
        a.dh0[(long)r * H + j] = dh;
        return;
    }
    dh += a.dout[(long)r * a.dout_stride + j];
    const float *tp = a.tape + (long)r * a.tape_stride;
    const float rg = tp[j], zg = tp[H + j], ng = tp[2 * H + j], ghn = tp[3 * H + j];
    const float hp = a.h_prev2[(long)r * a.h_prev2_stride + j];
    const float da_n = dh * (1.f - zg) * (1.f - ng * ng);
    const float da_z = dh * (hp - ng) * zg * (1.f - zg);
    const float da_r = da_n * ghn * rg * (1.f - rg);
    float *dxr = a.dx + (long)r * a.dx_stride;
    dxr[j] = da_r;
    dxr[H + j] = da_z;
    dxr[2 * H + j] = da_n;
    float *dg = a.dgh_out + (long)r * a.dgh_stride;
    dg[j] = da_r;
    dg[H + j] = da_z;
    dg[2 * H + j] = da_n * rg;
    a.dhz_out[(long)r * H + j] = dh * zg;
}
}  // namespace

void lav::launch_gru_fwd_step(const GruFwdArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_gru_fwd_step, dim3(a.H / 16, (a.R + 15) / 16), dim3(256), 0, st, a);
}

extern "C" int lav_gru_seq_forward(const float *x, int x_per_step, const float *h0, const float *w_hh, const float *b_hh, int R, int T,
                                   int H, float *out, float *tape, void *stream) {
    LAV_REQUIRE(R >= 0 && T >= 1 && H >= 16 && H % 16 == 0, "lav_gru_seq_forward: bad sizes (H must be a multiple of 16)");
    if (R == 0) return LAV_OK;
    LAV_REQUIRE(x && h0 && w_hh && b_hh && out, "lav_gru_seq_forward: null argument");
    LAV_REQUIRE((R + 15) / 16 <= 65535, "lav_gru_seq_forward: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("gru_seq_forward", st);
    for (int t = 0; t < T; ++t) {
        FwdArgs a{};
        a.h_prev = t == 0 ? h0 : out + (long)(t - 1) * H;
        a.h_prev_stride = t == 0 ? H : (long)T * H;
        a.x = x_per_step ? x + (long)t * 3 * H : x;
        a.x_stride = x_per_step ? (long)T * 3 * H : 3l * H;
        a.w_hh = w_hh; a.b_hh = b_hh;
        a.h_out = out + (long)t * H; a.h_out_stride = (long)T * H;
        a.tape = tape ? tape + (long)t * 4 * H : nullptr; a.tape_stride = (long)T * 4 * H;
        a.R = R; a.H = H;
        launch_gru_fwd_step(a, st);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" size_t lav_gru_seq_backward_workspace_bytes(int R, int H) { return lav::align_up(2 * (size_t)R * H * sizeof(float), 256); }

extern "C" int lav_gru_seq_backward(const float *dout, const float *tape, const float *out, const float *h0, const float *w_hh_t, int R,
                                    int T, int H, float *dx, float *dgh, float *dh0, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    LAV_REQUIRE(R >= 0 && T >= 1 && H >= 16 && H % 16 == 0, "lav_gru_seq_backward: bad sizes (H must be a multiple of 16)");
    if (R == 0) return LAV_OK;
    LAV_REQUIRE(dout && tape && out && h0 && w_hh_t && dx && dgh && dh0, "lav_gru_seq_backward: null argument");
    LAV_REQUIRE((R + 15) / 16 <= 65535, "lav_gru_seq_backward: too many rows");
    if (!workspace || workspace_bytes < lav_gru_seq_backward_workspace_bytes(R, H))
        return lav::fail(LAV_EWORKSPACE, "lav_gru_seq_backward: workspace %zu < %zu bytes", workspace_bytes, lav_gru_seq_backward_workspace_bytes(R, H));
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *dhz[2] = {static_cast<float *>(workspace), static_cast<float *>(workspace) + (size_t)R * H};
    const int tok = timer_begin("gru_seq_backward", st);
    // launch s handles the elementwise part of step t1 = T - 1 - s (s = T: only dh0)
    for (int s = 0; s <= T; ++s) {
        const int t1 = T - 1 - s;   // step whose tape is consumed; -1 on the last launch
        BwdArgs a;
        a.R = R; a.H = H;
        a.w_hh_t = w_hh_t;
        a.dgh_stride = (long)T * 3 * H;
        a.dgh = s == 0 ? nullptr : dgh + (long)(t1 + 1) * 3 * H;
        a.dhz = s == 0 ? nullptr : dhz[(s - 1) & 1];
        a.dout = t1 >= 0 ? dout + (long)t1 * H : nullptr; a.dout_stride = (long)T * H;
        a.tape = t1 >= 0 ? tape + (long)t1 * 4 * H : nullptr; a.tape_stride = (long)T * 4 * H;
        a.h_prev2 = t1 >= 1 ? out + (long)(t1 - 1) * H : h0; a.h_prev2_stride = t1 >= 1 ? (long)T * H : H;
        a.dx = t1 >= 0 ? dx + (long)t1 * 3 * H : nullptr; a.dx_stride = (long)T * 3 * H;
        a.dgh_out = t1 >= 0 ? dgh + (long)t1 * 3 * H : nullptr;
        a.dhz_out = dhz[s & 1];
        a.dh0 = dh0;
        hipLaunchKernelGGL(k_gru_bwd_step, dim3(H / 16, (R + 15) / 16), dim3(256), 0, st, a);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

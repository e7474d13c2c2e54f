// Two stacked 1-D convolutions in one launch:  y = epilogue( conv1x3_d( relu( conv3x1_d(x) + bA ) ) + bB )
//
// This is one half of ERFNet's non_bottleneck_1d block (lav/models/erfnet.py:45-56: conv3x1 -> ReLU -> conv1x3 -> BN
// (-> + input) -> ReLU), which makes up 66 of the 74 convolutions of the segmentation network.  As separate
// lav_conv2d launches each of them costs ~20 us, almost all of it per-launch fixed cost (scalar set-up, staging,
// epilogue, launch gap) - the matrix work is 1-3 us.  Fusing the pair removes a launch, a round trip of the
// intermediate through HBM and one staging pass.
//
// Decomposition: one workgroup = ONE image row (all W pixels) x ALL output channels.  The horizontal convolution only
// needs the intermediate of its own row, so the intermediate never leaves the CU: it is written to LDS in the layout
// the second convolution reads it from ([channel][W + 2 dB], zero halo).  The 4 waves split the row's 32-pixel groups
// and the 32-channel groups of the output (W/32 x 128/W = 4 for W in {32, 64, 128}).
//
//   stage   all C input channels of the three rows y-dA, y, y+dA: one asynchronous global->LDS DMA batch (16 B per
//           lane), one barrier - there is no staged pipeline to drain
//   phase A per 16-channel chunk and tap: 8 v_mfma_f32_32x32x2_f32 (exact fp32); the weight fragments come straight
//           from L2 in an MFMA-ready packing (8 consecutive k per lane = two 16-byte loads per tap and chunk,
//           prefetched one chunk ahead in registers), so the loops have no barriers at all
//   mid     relu(acc + bA) -> LDS; one barrier
//   phase B same loop over the intermediate's channels, taps = horizontal offsets 0, dB, 2dB into the padded row
//   epilogue +bB -> x scale + shift (eval BatchNorm) -> + residual -> ReLU -> NCHW store
//
// Bound: launch/latency (a pair is 0.06-0.45 GFLOP); the design target is the fixed cost, not the matrix pipes.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <type_traits>

#include "common.hpp"

namespace {
using namespace lav;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PairArgs {
    const float *x, *wA, *bA, *wB, *bB, *scale, *shift, *res, *zero_page;
    float *y;
    unsigned long long *trace;   // debug (LAV_PAIR_TRACE): [workgroup][8] wall-clock stamps
    int B, C, H, W, dA, dB, relu_post;
    int CP;  // output channels padded to a multiple of 32 (packed-weight stride)
};

// packed weights: [tap 3][chunk C/16][half 2][co CP][cp 8]  with k = chunk*16 + 2*cp + half
__device__ __forceinline__ void load_w(const float *__restrict__ wp, int tap, int chunk, int nchunk, int half, int co, int CP, float (&a)[8]) {
    const float4 *p = reinterpret_cast<const float4 *>(wp + ((((long)tap * nchunk + chunk) * 2 + half) * CP + co) * 8);
    const float4 lo = p[0], hi = p[1];
    a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w; a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
}

// KS = 2: 8 waves, the two halves of the channel loop on different waves, partial sums combined through LDS.
// R = weight ring: a wave keeps the fragments of R chunks (R x 3 taps x 8 registers) in flight; its chunk count is a
// multiple of R.  With R = all of a phase's chunks (ERFNet: 4 at 128 channels, 2 at 64, 1 at 16) the phase-A weights are
// requested once at kernel start, and each slot is refilled with the phase-B fragment of the same index the moment
// phase A has consumed it - the MFMA loops then never wait for L2 (before: one dependent round trip per chunk, 42 % of
// the matrix pipe's rate inside a workgroup).
template <int KS, int R>
__global__ __launch_bounds__(256 * KS) void k_conv1d_pair(PairArgs a) {
#define PAIR_STAMP(i) do { if (a.trace && threadIdx.x == 0) a.trace[(long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    PAIR_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid8 = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wid = wid8 & 3, kpart = wid8 >> 2;   // kpart: which half of the channel loop this wave walks (KS = 2)
    const int C = a.C, W = a.W, H = a.H, CP = a.CP;
    const int n = blockIdx.x / H, y = blockIdx.x - n * H;
    const int NPG = W >> 5;                       // 32-pixel groups per row: 1, 2 or 4
    const int npg_sh = NPG >> 1;                  // log2(NPG): W is 32, 64 or 128, so the index maths below is shifts
    const int pg = wid & (NPG - 1), cg = wid >> npg_sh;   // this wave's pixel group / 32-channel output group
    const int px = pg * 32 + l31;
    const int co_lane = cg * 32 + l31;            // A-operand row (output channel) of this lane
    const bool co_ok = cg * 32 < CP;              // wave has output channels at all (C = 16 -> one group)
    const int WM = W + 2 * a.dB;                  // padded intermediate row
    float *s_in = smem;                           // [C][3][W]
    float *s_mid = smem + C * 3 * W;              // [C][WM]
    const int nchunk = C >> 4;
    const int ch_lo = kpart * (nchunk / KS), ch_hi = kpart == KS - 1 ? nchunk : ch_lo + nchunk / KS;
    float *s_red = smem;                          // [4 waves][16][64] partial accumulators (aliases s_in once it is consumed)
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;

    // ---- stage the three input rows of every channel (16 B per lane; rows outside the image read the zero page)
    {
        const int W4 = W >> 2, F4 = C * 3 * W4, w4_sh = 3 + npg_sh;   // W4 = 8 << npg_sh
        const float *xn = a.x + (long)n * C * H * W;
        for (int f0 = 64 * wid8; f0 < F4; f0 += 256 * KS) {   // wave-uniform: this wave's 64 float4 slots f0 .. f0+63
            const int f = f0 + lane;
            const int q = f >> w4_sh, x4 = f & (W4 - 1);
            const int c = (int)(((unsigned)q * 43691u) >> 17), t = q - 3 * c;   // q / 3, exact below 2^16
            const int yy = y + (t - 1) * a.dA;
            const bool ok = f < F4 && yy >= 0 && yy < H;
            const float *src = ok ? xn + ((long)c * H + yy) * W + 4 * x4 : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + 4 * f0), 16, 0, 0);
        }
    }
    // first weight fragments and the epilogue vectors travel while the DMA is in flight
    float wr[R][3][8];
    const int nch = ch_hi - ch_lo;   // multiple of R
    if (co_ok) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) load_w(a.wA, t, ch_lo + i, nchunk, half, co_lane, CP, wr[i][t]);
    }
    float bAv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bAv[r] = a.bA[min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1)];
    // zero halo of the intermediate row
    if (const int j = tid & 31; j < 2 * a.dB) {   // dB <= 16: 32 lanes per channel cover both halos
        for (int c = tid >> 5; c < C; c += 8 * KS) s_mid[c * WM + (j < a.dB ? j : W + j)] = 0.f;
    }
    PAIR_STAMP(1);
    __syncthreads();  // DMA landed (hipcc drains vmcnt before the barrier), halo written
    PAIR_STAMP(2);

    // ---- phase A: vertical taps
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (co_ok) {
        for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ch = ch_lo + j0 + i;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float *b = s_in + ((ch * 16 + half) * 3 + t) * W + px;
#pragma unroll
                    for (int cp = 0; cp < 8; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[i][t][cp], b[cp * 2 * 3 * W], acc, 0, 0, 0);
                }
                // refill the slot: the next phase-A chunk that maps to it, else the phase-B chunk of the same slot
                const int nxt = j0 + i + R;
                const float *wsrc = nxt < nch ? a.wA : a.wB;
                const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, nchunk, half, co_lane, CP, wr[i][t]);
            }
        }
    }
    PAIR_STAMP(3);
    if constexpr (KS == 2) {   // combine the two halves of K: upper waves park their accumulators in LDS (s_in is consumed)
        __syncthreads();
        if (kpart == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kpart == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
        }
    }
    if (co_ok && kpart == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v = acc[r] + bAv[r];
            if (co < C) s_mid[co * WM + a.dB + px] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();

    PAIR_STAMP(4);
    // residual and epilogue vectors: all loads in flight together (one round trip, not sixteen); with a shallow weight
    // ring there are registers to spare and they travel during phase B
    const long plane = (long)H * W;
    const long base = (long)n * C * plane + (long)y * W + px;
    float rv[16], bBv[16], sv[16], tv[16];
    auto load_epilogue = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1);
            rv[r] = a.res ? a.res[base + co * plane] : 0.f;
            bBv[r] = a.bB[co];
            sv[r] = a.scale ? a.scale[co] : 1.f;
            tv[r] = a.scale ? a.shift[co] : 0.f;
        }
    };
    if constexpr (R <= 2) load_epilogue();
    // ---- phase B: horizontal taps over the intermediate
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (co_ok) {
        for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ch = ch_lo + j0 + i;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float *b = s_mid + (ch * 16 + half) * WM + px + t * a.dB;
#pragma unroll
                    for (int cp = 0; cp < 8; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[i][t][cp], b[cp * 2 * WM], acc, 0, 0, 0);
                }
                if (j0 + i + R < nch) {   // wave-uniform
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(a.wB, t, ch + R, nchunk, half, co_lane, CP, wr[i][t]);
                }
            }
        }
    }
    PAIR_STAMP(5);
    if constexpr (KS == 2) {
        if (kpart == 1) {   // s_red aliases s_in, which nobody reads in phase B
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kpart == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
    }
    if (!co_ok) return;
    // ---- epilogue
    if constexpr (R > 2) {
        asm volatile("" ::: "memory");   // deep weight ring: keep these loads down here, hoisted above the phases they only cost registers
        load_epilogue();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = fmaf(acc[r] + bBv[r], sv[r], tv[r]) + rv[r];
        if (a.relu_post) v = v > 0.f ? v : 0.f;
        if (co < C) a.y[base + co * plane] = v;
    }
    PAIR_STAMP(6);
#undef PAIR_STAMP
}

// ------------------------------------------------------------------------------------------------ bf16x6 variant
// The same pair on the bf16 matrix cores with every fp32 operand split exactly into three bf16 pieces (conv_split.hpp has
// the arithmetic and its error analysis): six v_mfma_f32_32x32x16_bf16 per 16 channels and tap instead of eight fp32 ones
// of twice the length - the phases of a 128-channel pair are matrix bound INSIDE the workgroup (1536 fp32 MFMAs on one CU).
//   input   all C channels of rows y-dA, y, y+dA travel HBM -> registers -> three bf16 pieces -> LDS as
//           [piece][16-channel chunk][k half][row][pixel] entries of 16 bytes (8 channels = one lane's B operand)
//   weights pre-split on the host, [tap][chunk][cout block][piece][lane][8 bf16]: three 16-byte loads per tap and chunk
//           straight from L2 into registers, a ring of chunks ahead
//   mid     relu(acc + bA) is split in registers and written to LDS in the same entry layout (a lane holds channels
//           4h..4h+3 of four 8-channel groups of its pixel: one 8-byte store per group and piece), zero halo of dB
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pair_split3(float x, unsigned &p0, unsigned &p1, unsigned &p2) {   // = split3 of conv_split.hpp
    const unsigned u = __float_as_uint(x), r = u + 0x8000u;
    p0 = ((r & 0x7f800000u) == 0x7f800000u ? u : r) & 0xffff0000u;
    const float r1 = x - __uint_as_float(p0);
    p1 = (__float_as_uint(r1) + 0x8000u) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p1);
    p2 = __float_as_uint(r2) + 0x8000u;
}
__device__ __forceinline__ unsigned pair_pack(unsigned even, unsigned odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }
// two values at once on v_cvt_pk_bf16_f32 (= split3_pair of conv_split.hpp: 13 instructions per pair instead of ~25)
typedef __bf16 pair_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pair_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pair_split3x2(float x0, float x1, unsigned &q0, unsigned &q1, unsigned &q2) {
    constexpr float M = 3.3895313892515355e38f;   // the largest finite bf16
    const float c0 = __builtin_amdgcn_fmed3f(x0, -M, M), c1 = __builtin_amdgcn_fmed3f(x1, -M, M);
    q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(pair_f32x2{c0, c1}, pair_bf16x2));
    const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
    q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(pair_f32x2{r0, r1}, pair_bf16x2));
    const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
    q2 = __builtin_bit_cast(unsigned, __builtin_convertvector(pair_f32x2{s0, s1}, pair_bf16x2));
}

struct PairSplitArgs {
    const float *x, *bA, *bB, *scale, *shift, *res;
    const unsigned char *wA, *wB;   // split packed weights
    float *y;
    int B, C, H, W, dA, dB, relu_post, CP;
};

template <int KS, int R>
__global__ __launch_bounds__(256 * KS) void k_conv1d_pair_split(PairSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid8 = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wid = wid8 & 3, kpart = wid8 >> 2;
    const int C = a.C, W = a.W, H = a.H, CP = a.CP;
    const int n = blockIdx.x / H, y = blockIdx.x - n * H;
    const int NPG = W >> 5, npg_sh = NPG >> 1;
    const int pg = wid & (NPG - 1), cg = wid >> npg_sh;
    const int px = pg * 32 + l31;
    const bool co_ok = cg * 32 < CP;
    const int WM = W + 2 * a.dB;
    const int nchunk = C >> 4, nblk = CP >> 5;
    // LDS: s_in [3 pieces][nchunk][2][3 rows][W] x 16 B, s_mid [3 pieces][nchunk][2][WM] x 16 B
    const int in_piece = nchunk * 2 * 3 * W * 16, mid_piece = nchunk * 2 * WM * 16;
    unsigned char *s_in = smem_raw, *s_mid = smem_raw + 3 * in_piece;
    float *s_red = reinterpret_cast<float *>(smem_raw);   // [4 waves][16][64] partial accumulators (aliases s_in once it is consumed)
    const int ch_lo = kpart * (nchunk / KS), ch_hi = kpart == KS - 1 ? nchunk : ch_lo + nchunk / KS;
    const int nch = ch_hi - ch_lo;   // multiple of R

    // first weight fragments travel while the input is staged
    u32x4 wr[R][3][3];
    auto load_w = [&](const unsigned char *wp, int t, int chunk, u32x4 (&dst)[3]) {
        const unsigned char *p = wp + ((((long)t * nchunk + chunk) * nblk + cg) * 3) * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const u32x4 *>(p + pl * 1024);
    };
    if (co_ok) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) load_w(a.wA, t, ch_lo + i, wr[i][t]);
    }
    // ---- stage the three input rows of every channel: task = (chunk, row, pixel), 16 channel loads each
    {
        const int ntask = nchunk * 3 * W;
        const long plane = (long)H * W;
        const float *xn = a.x + (long)n * C * plane;
        constexpr int NTK = 3;   // tasks per thread: at most 768 tasks (32 channels x 128 pixels) over 256 threads
        float v[NTK][16];
        int task[NTK];
        bool ok[NTK];
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            task[u] = tid + u * 256 * KS;
            const int tk = min(task[u], ntask - 1);
            const int pxs = tk & (W - 1), q = tk >> (5 + npg_sh), c = q / 3, t = q - 3 * c;
            const int yy = y + (t - 1) * a.dA;
            ok[u] = task[u] < ntask && yy >= 0 && yy < H;
            const float *src = xn + ((long)c * 16 * H + (ok[u] ? yy : 0)) * W + pxs;
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) v[u][ch] = src[ch * plane];
        }
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            if (task[u] < ntask) {
                const int pxs = task[u] & (W - 1), q = task[u] >> (5 + npg_sh), c = q / 3, t = q - 3 * c;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 q3[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = ok[u] ? v[u][8 * h + 2 * e] : 0.f, x1 = ok[u] ? v[u][8 * h + 2 * e + 1] : 0.f;
                        unsigned p0, p1, p2;
                        pair_split3x2(x0, x1, p0, p1, p2);
                        q3[0][e] = p0; q3[1][e] = p1; q3[2][e] = p2;
                    }
                    const int entry = ((c * 2 + h) * 3 + t) * W + pxs;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(s_in + pl * in_piece + entry * 16) = q3[pl];
                }
            }
        }
    }
    float bAv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bAv[r] = a.bA[min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1)];
    // zero halo of the intermediate row (all pieces, all 8-channel groups)
    {
        const int ngrp = nchunk * 2, nh = 2 * a.dB;
        for (int i = tid; i < 3 * ngrp * nh; i += 256 * KS) {
            const int j = i % nh, g = (i / nh) % ngrp, pl = i / (nh * ngrp);
            *reinterpret_cast<u32x4 *>(s_mid + pl * mid_piece + (g * WM + (j < a.dB ? j : W + j)) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();

    // ---- phase A: vertical taps
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mma6 = [&](const u32x4 (&w)[3], const u32x4 (&b)[3]) {
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[PA[k]]), __builtin_bit_cast(bf16x8, b[PB[k]]), acc, 0, 0, 0);
    };
    if (co_ok) {
        for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ch = ch_lo + j0 + i;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    u32x4 b[3];
                    const int entry = ((ch * 2 + half) * 3 + t) * W + px;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(s_in + pl * in_piece + entry * 16);
                    mma6(wr[i][t], b);
                }
                // refill the slot: the next phase-A chunk that maps to it, else the phase-B chunk of the same slot
                const int nxt = j0 + i + R;
                const unsigned char *wsrc = nxt < nch ? a.wA : a.wB;
                const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, wr[i][t]);
            }
        }
    }
    if constexpr (KS == 2) {   // combine the two halves of K (s_in is consumed)
        __syncthreads();
        if (kpart == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kpart == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
        }
    }
    if (co_ok && kpart == 0) {
        // this lane: pixel px, channels cg*32 + 8*g8 + 4*half + e (g8 = 0..3, e = 0..3) = acc[4*g8 + e]
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) {
            unsigned pk[3][2];   // pieces of channels (0, 1) and (2, 3) of the lane's four
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v0 = acc[4 * g8 + 2 * e] + bAv[4 * g8 + 2 * e], v1 = acc[4 * g8 + 2 * e + 1] + bAv[4 * g8 + 2 * e + 1];
                pair_split3x2(v0 > 0.f ? v0 : 0.f, v1 > 0.f ? v1 : 0.f, pk[0][e], pk[1][e], pk[2][e]);
            }
            const int grp = cg * 4 + g8;   // 8-channel group = chunk * 2 + k half
            if (grp * 8 < C) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x2 *>(s_mid + pl * mid_piece + (grp * WM + a.dB + px) * 16 + half * 8) =
                        u32x2{pk[pl][0], pk[pl][1]};
            }
        }
    }
    __syncthreads();

    // residual and epilogue vectors: all loads in flight together, travelling during phase B
    const long plane = (long)H * W;
    const long base = (long)n * C * plane + (long)y * W + px;
    float rv[16], bBv[16], sv[16], tv[16];
    const float *res_p = a.res ? a.res : a.x, *scale_p = a.scale ? a.scale : a.bB, *shift_p = a.scale ? a.shift : a.bB;
    auto load_epilogue = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1);
            rv[r] = res_p[base + co * plane];
            bBv[r] = a.bB[co];
            sv[r] = scale_p[co];
            tv[r] = shift_p[co];
        }
    };
    if constexpr (R <= 2) load_epilogue();   // a shallow weight ring leaves registers for them during phase B
    // ---- phase B: horizontal taps over the intermediate
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (co_ok) {
        for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ch = ch_lo + j0 + i;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    u32x4 b[3];
                    const int entry = (ch * 2 + half) * WM + px + t * a.dB;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(s_mid + pl * mid_piece + entry * 16);
                    mma6(wr[i][t], b);
                }
                if (j0 + i + R < nch) {   // wave-uniform
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(a.wB, t, ch + R, wr[i][t]);
                }
            }
        }
    }
    if constexpr (KS == 2) {
        if (kpart == 1) {   // s_red aliases s_in, which nobody reads in phase B
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kpart == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
    }
    if (!co_ok) return;
    if constexpr (R > 2) {
        asm volatile("" ::: "memory");   // deep weight ring: keep these loads down here
        load_epilogue();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r] + bBv[r];
        if (a.scale) v = fmaf(v, sv[r], tv[r]);
        if (a.res) v += rv[r];
        if (a.relu_post) v = v > 0.f ? v : 0.f;
        if (co < C) a.y[base + co * plane] = v;
    }
}

// ------------------------------------------------------------------------------------------------ persistent chain
// A RUN of pairs (the non_bottleneck_1d blocks of one ERFNet stage: same channels and map, dilations per pair) as ONE launch
// (round 4).  One workgroup per image row, as above, walks the whole run.  What that removes per pair: the kernel boundary,
// the argument fetch, the staging of the workgroup's OWN row (it is converted into the next pair's LDS slot in the epilogue,
// never re-read), the residual's round trip (a block's input row stays in registers until its second pair adds it), and the
// exposed latency of the first weight fragments (requested while the previous pair's last chunks are multiplied).  What is left
// between two pairs is the hand-off of the two NEIGHBOUR rows y -+ dA: every row is published with write-through (sc1) stores,
// `s_waitcnt vmcnt(0)`, barrier, then a relaxed agent-scope counter flags[row] = pairs done (MI355X guide, Guideline 16 write-
// through form); a consumer polls its two neighbours' counters with one lane and reads their rows with sc1 loads.  Every pair
// writes its own buffer, so there is no write-after-read hazard whatever the dilations.  Progress: the row with the fewest pairs
// done never waits (its neighbours are at least as far), so the run completes as soon as every workgroup is resident
// (B*H <= the CU count: checked by the host); spins are bounded and raise a status word instead of hanging.
constexpr int CHAIN_MAX = 16;
struct PairChainArgs {
    const float *x0;                  // input of the run [B][C][H][W]
    float *out[CHAIN_MAX];            // output of pair i (the last one is the run's result)
    const unsigned char *wA[CHAIN_MAX], *wB[CHAIN_MAX];
    const float *bA[CHAIN_MAX], *bB[CHAIN_MAX], *scale[CHAIN_MAX], *shift[CHAIN_MAX];
    unsigned char dA[CHAIN_MAX], dB[CHAIN_MAX], res[CHAIN_MAX], relu[CHAIN_MAX];   // res: add the input of the pair before (block residual)
    int *flags;                       // [B*H] pairs completed by the row, zeroed at every launch
    int *sticky;                      // [0] workgroups that gave up waiting, [1] launches - since the caller zero-filled the workspace
    int B, C, H, W, CP, npairs;
    long long spin_limit;
    long long *trace;                 // debug (LAV_PAIR_CHAIN_TRACE), else null
};

// TRACE (LAV_PAIR_CHAIN_TRACE): thread 0 of every workgroup accumulates the shader-clock cycles of each phase over the run into
// a.trace[workgroup][8]: 0 whole run, 1 counter waits, 2 neighbour rows (loads, conversion, LDS), 3 halo + barrier, 4 phase A,
// 5 combine + intermediate row, 6 phase B, 7 epilogue + publication
constexpr size_t CHAIN_STATIC_LDS = 2064;   // s_abort + s_epi of k_conv1d_pair_chain (hipcc: "LDS Size [bytes/block]: 2064")
template <int KS, int R, bool TRACE = false>
__global__ __launch_bounds__(256 * KS) void k_conv1d_pair_chain(PairChainArgs a) {
    long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0, tr_t0 = 0;
    if constexpr (TRACE) { tr_t0 = tr_t = clock64(); }
#define CH_MARK(i) do { if constexpr (TRACE) { const long long now_ = clock64(); tr_acc[i] += now_ - tr_t; tr_t = now_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid8 = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wid = wid8 & 3, kpart = wid8 >> 2;
    const int C = a.C, W = a.W, H = a.H, CP = a.CP;
    const int n = blockIdx.x / H, y = blockIdx.x - n * H;
    const int NPG = W >> 5, npg_sh = NPG >> 1;
    const int pg = wid & (NPG - 1), cg = wid >> npg_sh;
    const int px = pg * 32 + l31;
    const bool co_ok = cg * 32 < CP;
    const int nchunk = C >> 4, nblk = CP >> 5;
    // LDS: s_red [4 waves][16][64] floats (its own region here: the next pair's rows are staged while nothing else is read),
    //      s_in [3 pieces][nchunk][2][3 rows][W] x 16 B, s_mid [3 pieces][nchunk][2][W + 2 dB] x 16 B (sized for the run's largest dB)
    const int in_piece = nchunk * 2 * 3 * W * 16;
    float *s_red = reinterpret_cast<float *>(smem_raw);
    unsigned char *s_in = smem_raw + 16384, *s_mid = s_in + 3 * in_piece;
    __shared__ int s_abort;
    __shared__ float s_epi[4][128];   // per pair: bias of the 3x1 convolution, bias of the 1x3 one, BatchNorm scale and shift
    const int ch_lo = kpart * (nchunk / KS), ch_hi = kpart == KS - 1 ? nchunk : ch_lo + nchunk / KS;
    const int nch = ch_hi - ch_lo;   // multiple of R
    const long plane = (long)H * W;
    const long base = (long)n * C * plane + (long)y * W + px;
    if (tid == 0) s_abort = 0;
    if (blockIdx.x == 0 && tid == 0) atomicAdd(a.sticky + 1, 1);   // launches on this workspace (k_zero_ints did this until round 6: it no longer runs in front of every launch)

    u32x4 wr[R][3][3];
    auto load_w = [&](const unsigned char *wp, int t, int chunk, u32x4 (&dst)[3]) {
        const unsigned char *p = wp + ((((long)t * nchunk + chunk) * nblk + cg) * 3) * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const u32x4 *>(p + pl * 1024);
    };
    // rows of `src` into s_in: all three (first pair) or only the neighbours' (t = 0, 2); task = (chunk, row, pixel), 16 channel loads
    auto stage_rows = [&](const float *src, int dA, auto ALL_ROWS_, bool coherent) __attribute__((always_inline)) {
        constexpr bool all_rows = decltype(ALL_ROWS_)::value;
        constexpr int nrow = all_rows ? 3 : 2;
        const int ntask = nchunk * nrow * W;
        const float *xn = src + (long)n * C * plane;
        // tasks per thread: C W <= 4096, so three rows are at most 768 tasks and two rows 512, over 256 KS threads
        constexpr int NTK = all_rows ? (KS == 2 ? 2 : 3) : (KS == 2 ? 1 : 2);
        float v[NTK][16];
        int task[NTK];
        bool ok[NTK];
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            task[u] = tid + u * 256 * KS;
            const int tk = min(task[u], ntask - 1);
            const int pxs = tk & (W - 1), q = tk >> (5 + npg_sh), c = q / nrow, tt = q - nrow * c;
            const int t = all_rows ? tt : 2 * tt;
            const int yy = y + (t - 1) * dA;
            ok[u] = task[u] < ntask && yy >= 0 && yy < H;
            const float *sp = xn + ((long)c * 16 * H + (ok[u] ? yy : 0)) * W + pxs;
#pragma unroll
            for (int ch = 0; ch < 16; ++ch)
                v[u][ch] = coherent ? __hip_atomic_load(sp + ch * plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sp[ch * plane];
        }
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            if (task[u] < ntask) {
                const int pxs = task[u] & (W - 1), q = task[u] >> (5 + npg_sh), c = q / nrow, tt = q - nrow * c;
                const int t = all_rows ? tt : 2 * tt;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 q3[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = ok[u] ? v[u][8 * h + 2 * e] : 0.f, x1 = ok[u] ? v[u][8 * h + 2 * e + 1] : 0.f;
                        unsigned p0, p1, p2;
                        pair_split3x2(x0, x1, p0, p1, p2);
                        q3[0][e] = p0; q3[1][e] = p1; q3[2][e] = p2;
                    }
                    const int entry = ((c * 2 + h) * 3 + t) * W + pxs;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(s_in + pl * in_piece + entry * 16) = q3[pl];
                }
            }
        }
    };
    stage_rows(a.x0, a.dA[0], std::true_type{}, false);
    if (co_ok) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) load_w(a.wA[0], t, ch_lo + i, wr[i][t]);
    }
    // the block's input row in registers (what the block's second pair adds back): this lane's 16 channels of pixel px
    float blk_in[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) blk_in[r] = a.x0[base + (long)min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1) * plane];

    auto mma6 = [&](f32x16 &acc, const u32x4 (&w)[3], const u32x4 (&b)[3]) {
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[PA[k]]), __builtin_bit_cast(bf16x8, b[PB[k]]), acc, 0, 0, 0);
    };

    for (int p = 0; p < a.npairs; ++p) {
        const int dA = a.dA[p], dB = a.dB[p];
        const int WM = W + 2 * dB, mid_piece = nchunk * 2 * WM * 16;
        const unsigned char *wA = a.wA[p], *wB = a.wB[p];
        const bool last = p + 1 == a.npairs;
        if (p > 0) {
            // ---- the two neighbour rows of the previous pair's output: wait for their counters, then stage them
            if (wid8 == 0) {
                const int yu = y - dA, yd = y + dA;
                long long spins = 0;
                bool ok = false;
                while (!ok) {
                    const int fu = yu >= 0 ? __hip_atomic_load(a.flags + n * H + max(yu, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p;
                    const int fd = yd < H ? __hip_atomic_load(a.flags + n * H + min(yd, H - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p;
                    ok = fu >= p && fd >= p;
                    if (!ok) {
                        // never hang the device: a row that gives up raises the launch's abort word (sticky[2], cleared by k_zero_ints), every
                        // other row that still waits sees it within 64 polls and leaves too; each of them voids its row of the output (below)
                        ++spins;
                        const bool timed_out = spins > a.spin_limit;
                        const bool peer_gone = !timed_out && (spins & 63) == 0 && __hip_atomic_load(a.sticky + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (timed_out || peer_gone) {
                            if (lane == 0) {
                                s_abort = 1;
                                if (timed_out) { atomicAdd(a.sticky, 1); __hip_atomic_store(a.sticky + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            __syncthreads();
            if (*(volatile int *)&s_abort) {
                // (uniform after the barrier.)  This row stops HERE - it never computes on a neighbour row that was not published and never
                // raises its own counter again - and voids its row of the run's result: the output then holds rows that completed every
                // pair (correct) and NaN rows, nothing computed from stale data.  NaN spreads through the layers behind the run (the
                // frame's non-finite check counts it; the reference agent's rule for NaN waypoints applies).
                float *o = a.out[a.npairs - 1] + (long)n * C * plane + (long)y * W;
                for (int i = tid; i < C * W; i += 256 * KS) o[(long)(i / W) * plane + (i % W)] = __uint_as_float(0x7fc00000u);
                return;
            }
            CH_MARK(1);
            stage_rows(a.out[p - 1], dA, std::false_type{}, true);
            CH_MARK(2);
        }
        if (tid < C) {   // the pair's epilogue vectors: through LDS, so that they hold no registers while the phases run
            s_epi[0][tid] = a.bA[p][tid];
            s_epi[1][tid] = a.bB[p][tid];
            s_epi[2][tid] = a.scale[p][tid];
            s_epi[3][tid] = a.shift[p][tid];
        }
        {   // zero halo of the intermediate row (all pieces, all 8-channel groups)
            const int ngrp = nchunk * 2, nh = 2 * dB;
            for (int i = tid; i < 3 * ngrp * nh; i += 256 * KS) {
                const int j = i % nh, g = (i / nh) % ngrp, pl = i / (nh * ngrp);
                *reinterpret_cast<u32x4 *>(s_mid + pl * mid_piece + (g * WM + (j < dB ? j : W + j)) * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        CH_MARK(3);

        // ---- phase A: vertical taps
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (co_ok) {
            for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int ch = ch_lo + j0 + i;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        u32x4 b[3];
                        const int entry = ((ch * 2 + half) * 3 + t) * W + px;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(s_in + pl * in_piece + entry * 16);
                        mma6(acc, wr[i][t], b);
                    }
                    const int nxt = j0 + i + R;
                    const unsigned char *wsrc = nxt < nch ? wA : wB;
                    const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, wr[i][t]);
                }
            }
        }
        CH_MARK(4);
        if constexpr (KS == 2) {
            if (kpart == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
            }
        }
        if (co_ok && kpart == 0) {
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                unsigned pk[3][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cch = min(cg * 32 + 8 * g8 + 4 * half + 2 * e, C - 2);
                    const float v0 = acc[4 * g8 + 2 * e] + s_epi[0][cch], v1 = acc[4 * g8 + 2 * e + 1] + s_epi[0][cch + 1];
                    pair_split3x2(v0 > 0.f ? v0 : 0.f, v1 > 0.f ? v1 : 0.f, pk[0][e], pk[1][e], pk[2][e]);
                }
                const int grp = cg * 4 + g8;
                if (grp * 8 < C) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        *reinterpret_cast<u32x2 *>(s_mid + pl * mid_piece + (grp * WM + dB + px) * 16 + half * 8) = u32x2{pk[pl][0], pk[pl][1]};
                }
            }
        }
        __syncthreads();
        CH_MARK(5);

        // ---- phase B: horizontal taps over the intermediate; the ring is refilled with the NEXT pair's first vertical fragments
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *wNext = a.wA[last ? p : p + 1];
        if (co_ok) {
            for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int ch = ch_lo + j0 + i;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        u32x4 b[3];
                        const int entry = (ch * 2 + half) * WM + px + t * dB;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(s_mid + pl * mid_piece + entry * 16);
                        mma6(acc, wr[i][t], b);
                    }
                    const int nxt = j0 + i + R;
                    const unsigned char *wsrc = nxt < nch ? wB : wNext;
                    const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, wr[i][t]);
                }
            }
        }
        CH_MARK(6);
        if constexpr (KS == 2) {   // (phase A's partials were read before the barrier that followed the intermediate row's stores)
            if (kpart == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
            }
        }
        // ---- epilogue: bias, BatchNorm, block residual, ReLU; the row goes out write-through and into the next pair's LDS slot
        if (co_ok && kpart == 0) {
            float vout[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co_ = min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1);
                float v = fmaf(acc[r] + s_epi[1][co_], s_epi[2][co_], s_epi[3][co_]);   // the single-pair kernel's arithmetic, bit for bit
                if (a.res[p]) v += blk_in[r];
                if (a.relu[p]) v = v > 0.f ? v : 0.f;
                vout[r] = v;
            }
            float *yo = a.out[p];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < C) {
                    if (last) yo[base + co * plane] = vout[r];
                    else __hip_atomic_store(yo + base + co * plane, vout[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!last) {
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    unsigned pk[3][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) pair_split3x2(vout[4 * g8 + 2 * e], vout[4 * g8 + 2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
                    const int grp = cg * 4 + g8;
                    if (grp * 8 < C) {
                        const int entry = (grp * 3 + 1) * W + px;   // (chunk * 2 + k half) = grp, row slot t = 1
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            *reinterpret_cast<u32x2 *>(s_in + pl * in_piece + entry * 16 + half * 8) = u32x2{pk[pl][0], pk[pl][1]};
                    }
                }
                if (a.res[p]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) blk_in[r] = vout[r];
                }
            }
        }
        if (last) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's row stores have left (write-through): the counter may follow
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + n * H + y, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_MARK(7);
        if (*(volatile int *)&s_abort) return;
    }
    if constexpr (TRACE) {
        CH_MARK(7);
        tr_acc[0] = clock64() - tr_t0;
        if (tid == 0)
            for (int i = 0; i < 8; ++i) a.trace[(long)blockIdx.x * 8 + i] = tr_acc[i];
    }
#undef CH_MARK
}

// ------------------------------------------------------------------------------------------------ persistent chain, fp16 pieces
// Round 6 (LAV_CONV_F16X3 for ERFNet's runs): the same run on TWO fp16 pieces per operand and THREE v_mfma_f32_32x32x16_f16 per 16
// k-steps instead of three bf16 pieces and six products - half the matrix instructions, 2/3 of the weight bytes a row workgroup
// streams per pair (393 instead of 590 KB at 128 channels), 2/3 of the LDS operand traffic.  fp16 has five exponent bits, so every
// operand is scaled by a power of two first (exact): the weights by their convolution's largest magnitude at packing time (the
// scale rides behind the packed pieces), the activations by a scale the WORKGROUP derives per pair from the largest finite
// magnitude of the three rows it multiplies - its own row's maximum (a wave reduction in the epilogue that produced the row) and
// the two neighbour rows' maxima, which travel with the hand-off IN its flag: a row publishes ONE 64-bit word per pair, {pairs done |
// maximum of the row it just wrote} (rowmax[pair][row], zero at the start of a pass), and its neighbours poll that word - count and
// maximum arrive in one load, the plain progress counters of the bf16 run are not used.  The intermediate row (ReLU of the vertical convolution) is scaled by a BOUND instead of its maximum - (largest input
// magnitude) x (largest L1 norm of a filter, from the packing) + largest |bias| - so that no barrier is added between the two
// phases: two fp16 pieces keep 22 significant bits over 18 binades below the scale's top, a bound that is loose by a few binades
// costs nothing (tests/test_gpu_conv.py holds the run to the fp32 dot product's error against float64).
// Differences to the bf16 run above, otherwise the same kernel: the workgroup's own row stays in registers until the pair's scale
// is known (it was converted into the next pair's LDS slot in the epilogue); the intermediate row's zero halo is laid out for the
// run's largest horizontal dilation and written once.
struct PairChainF16Args {
    PairChainArgs c;
    const float *tA[CHAIN_MAX], *tB[CHAIN_MAX];   // tails of the fp16 sections: {weight scale, largest L1 norm of a filter}
    // [CHAIN_MAX][rm_stride] 64-bit words, the run's rows from its region's first: {pairs done = i + 1 | largest finite magnitude of pair i's output row},
    // zero at the start of a pass.  The word IS the hand-off's flag: count and maximum arrive in one load (a counter followed by a separate
    // maximum cost the producer a second drained store and the consumer a second round trip per pair - the first version of this kernel).
    unsigned long long *rowmax;
    int rm_stride;
    int dbmax;
};
typedef _Float16 pair_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pair_f16x8 __attribute__((ext_vector_type(8)));
// two values -> two fp16 pieces each (= split2h_pair of conv_split_kernel.hpp): u = h0 + h1 + O(2^-22 |u|), |u| <= 32768 by the caller's scale
__device__ __forceinline__ void pair_split2h(float u0, float u1, unsigned &q0, unsigned &q1) {
    const pair_f16x2 h0 = __builtin_convertvector(pair_f32x2{u0, u1}, pair_f16x2);
    const pair_f32x2 f0 = __builtin_convertvector(h0, pair_f32x2);
    const pair_f16x2 h1 = __builtin_convertvector(pair_f32x2{u0 - f0[0], u1 - f0[1]}, pair_f16x2);
    q0 = __builtin_bit_cast(unsigned, h0);
    q1 = __builtin_bit_cast(unsigned, h1);
}

// TRACE (LAV_PAIR_CHAIN_TRACE): as in the bf16 run - thread 0 of every workgroup accumulates the shader-clock cycles of each phase into a.trace[workgroup][8]:
// 0 whole run, 1 hand-off waits, 2 neighbour rows + own row (loads, conversion, LDS), 3 barrier, 4 phase A, 5 combine + intermediate row, 6 phase B,
// 7 epilogue + publication
template <int KS, int R, bool TRACE = false>
__global__ __launch_bounds__(256 * KS) void k_conv1d_pair_chain_f16(PairChainF16Args fa) {
    const PairChainArgs &a = fa.c;
    long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0, tr_t0 = 0;
    if constexpr (TRACE) { tr_t0 = tr_t = clock64(); }
#define CH_MARK(i) do { if constexpr (TRACE) { const long long now_ = clock64(); tr_acc[i] += now_ - tr_t; tr_t = now_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid8 = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wid = wid8 & 3, kpart = wid8 >> 2;
    const int C = a.C, W = a.W, H = a.H, CP = a.CP;
    const int n = blockIdx.x / H, y = blockIdx.x - n * H;
    const int NPG = W >> 5, npg_sh = NPG >> 1;
    const int pg = wid & (NPG - 1), cg = wid >> npg_sh;
    const int px = pg * 32 + l31;
    const bool co_ok = cg * 32 < CP;
    const int nchunk = C >> 4, nblk = CP >> 5;
    const long rm_stride = fa.rm_stride;
    // LDS: s_red [4 waves][16][64] floats, s_in [2 pieces][nchunk][2][3 rows][W] x 16 B, s_mid [2 pieces][nchunk][2][W + 2 dbmax] x 16 B
    const int in_piece = nchunk * 2 * 3 * W * 16;
    const int dbmax = fa.dbmax, WM = W + 2 * dbmax, mid_piece = nchunk * 2 * WM * 16;
    float *s_red = reinterpret_cast<float *>(smem_raw);
    unsigned char *s_in = smem_raw + 16384, *s_mid = s_in + 2 * in_piece;
    __shared__ int s_abort;
    __shared__ float s_epi[4][128];
    __shared__ float s_wmax[8];   // per wave: largest finite magnitude of what it holds (first pair: the staged rows; then: the output row)
    __shared__ float s_bmax[2];   // per wave: largest |bias| of the vertical convolution
    __shared__ float s_m3;        // largest finite magnitude of the pair's three input rows
    const int ch_lo = kpart * (nchunk / KS), ch_hi = kpart == KS - 1 ? nchunk : ch_lo + nchunk / KS;
    const int nch = ch_hi - ch_lo;   // multiple of R
    const long plane = (long)H * W;
    const long base = (long)n * C * plane + (long)y * W + px;
    if (tid == 0) s_abort = 0;
    if (blockIdx.x == 0 && tid == 0) atomicAdd(a.sticky + 1, 1);

    u32x4 wr[R][3][2];
    auto load_w = [&](const unsigned char *wp, int t, int chunk, u32x4 (&dst)[2]) {
        const unsigned char *p = wp + ((((long)t * nchunk + chunk) * nblk + cg) * 2) * 1024 + lane * 16;
        dst[0] = *reinterpret_cast<const u32x4 *>(p);
        dst[1] = *reinterpret_cast<const u32x4 *>(p + 1024);
    };
    // rows of `src` for s_in - all three (first pair) or only the neighbours' (t = 0, 2) - in two steps: the loads (returning the largest
    // finite magnitude this thread saw), and the conversion once the scale is known.  task = (chunk, row, pixel), 16 channel loads
    constexpr int NTKMAX = KS == 2 ? 2 : 3;
    float v[NTKMAX][16];
    bool okr[NTKMAX];
    auto stage_load = [&](const float *src, int dA, auto ALL_ROWS_, bool coherent) __attribute__((always_inline)) -> float {
        constexpr bool all_rows = decltype(ALL_ROWS_)::value;
        constexpr int nrow = all_rows ? 3 : 2;
        constexpr int NTK = all_rows ? (KS == 2 ? 2 : 3) : (KS == 2 ? 1 : 2);
        const int ntask = nchunk * nrow * W;
        const float *xn = src + (long)n * C * plane;
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            const int task = tid + u * 256 * KS;
            const int tk = min(task, ntask - 1);
            const int pxs = tk & (W - 1), q = tk >> (5 + npg_sh), c = q / nrow, tt = q - nrow * c;
            const int t = all_rows ? tt : 2 * tt;
            const int yy = y + (t - 1) * dA;
            okr[u] = task < ntask && yy >= 0 && yy < H;
            const float *sp = xn + ((long)c * 16 * H + (okr[u] ? yy : 0)) * W + pxs;
#pragma unroll
            for (int ch = 0; ch < 16; ++ch)
                v[u][ch] = coherent ? __hip_atomic_load(sp + ch * plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sp[ch * plane];
        }
#pragma unroll
        for (int u = 0; u < NTK; ++u)
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) {
                v[u][ch] = okr[u] ? v[u][ch] : 0.f;
                m = fmaxf(m, finite_abs(v[u][ch]));
            }
        return m;
    };
    auto stage_store = [&](auto ALL_ROWS_, float inv) __attribute__((always_inline)) {
        constexpr bool all_rows = decltype(ALL_ROWS_)::value;
        constexpr int nrow = all_rows ? 3 : 2;
        constexpr int NTK = all_rows ? (KS == 2 ? 2 : 3) : (KS == 2 ? 1 : 2);
        const int ntask = nchunk * nrow * W;
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            const int task = tid + u * 256 * KS;
            if (task < ntask) {
                const int pxs = task & (W - 1), q = task >> (5 + npg_sh), c = q / nrow, tt = q - nrow * c;
                const int t = all_rows ? tt : 2 * tt;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 q2[2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned p0, p1;
                        pair_split2h(v[u][8 * h + 2 * e] * inv, v[u][8 * h + 2 * e + 1] * inv, p0, p1);
                        q2[0][e] = p0; q2[1][e] = p1;
                    }
                    const int entry = ((c * 2 + h) * 3 + t) * W + pxs;
                    *reinterpret_cast<u32x4 *>(s_in + entry * 16) = q2[0];
                    *reinterpret_cast<u32x4 *>(s_in + in_piece + entry * 16) = q2[1];
                }
            }
        }
    };
    // ---- first pair: the three rows of the run's input, scaled by their own largest magnitude (one extra barrier per RUN)
    float m3;
    {
        const float lm = wave_finite_absmax(stage_load(a.x0, a.dA[0], std::true_type{}, false));
        if (lane == 0) s_wmax[wid8] = lm;
    }
    if (co_ok) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) load_w(a.wA[0], t, ch_lo + i, wr[i][t]);
    }
    // the block's input row in registers (what the block's second pair adds back): this lane's 16 channels of pixel px
    float blk_in[16], own[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { blk_in[r] = a.x0[base + (long)min(cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, C - 1) * plane]; own[r] = 0.f; }
    {   // zero halo of the intermediate row, once for the run (both pieces, all 8-channel groups)
        const int ngrp = nchunk * 2, nh = 2 * dbmax;
        for (int i = tid; i < 2 * ngrp * nh; i += 256 * KS) {
            const int j = i % nh, g = (i / nh) % ngrp, pl = i / (nh * ngrp);
            *reinterpret_cast<u32x4 *>(s_mid + pl * mid_piece + (g * WM + (j < dbmax ? j : W + j)) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    {
        m3 = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) m3 = fmaxf(m3, s_wmax[i]);
        stage_store(std::true_type{}, 1.f / f16_scale_of(m3));
    }

    auto mma3 = [&](f32x16 &acc, const u32x4 (&w)[2], const u32x4 (&b)[2]) {
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};   // smallest terms first: w1 b0, w0 b1, w0 b0
#pragma unroll
        for (int k = 0; k < 3; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pair_f16x8, w[HA[k]]), __builtin_bit_cast(pair_f16x8, b[HB[k]]), acc, 0, 0, 0);
    };
    float m_own = 0.f;   // (wave 0) largest finite magnitude of this row's last output

    for (int p = 0; p < a.npairs; ++p) {
        const int dA = a.dA[p], dB = a.dB[p];
        const unsigned char *wA = a.wA[p], *wB = a.wB[p];
        const bool last = p + 1 == a.npairs;
        const float swA = fa.tA[p][0], l1A = fa.tA[p][1], swB = fa.tB[p][0];
        {   // the pair's epilogue vectors through LDS, and the largest |bias| of the vertical convolution (for the intermediate row's bound):
            // fetched by waves 1-2 while wave 0 polls its neighbours (everybody is past the previous pair's last read of them: the barrier
            // that closed it); behind the hand-off their round trip stood in front of every pair's first matrix instruction
            float bm = 0.f;
            const int et = tid - 64;
            if (et >= 0 && et < C) {
                const float b0 = a.bA[p][et];
                s_epi[0][et] = b0;
                s_epi[1][et] = a.bB[p][et];
                s_epi[2][et] = a.scale[p][et];
                s_epi[3][et] = a.shift[p][et];
                bm = finite_abs(b0);
            }
            if (wid8 == 1 || wid8 == 2) {
                bm = wave_finite_absmax(bm);
                if (lane == 0) s_bmax[wid8 - 1] = bm;
            }
        }
        if (p > 0) {
            // ---- the two neighbour rows of the previous pair's output: wait for their counters, take their maxima, then stage them
            if (wid8 == 0) {
                const int yu = y - dA, yd = y + dA;
                long long spins = 0;
                bool ok = false;
                const unsigned long long *rm = fa.rowmax + (long)(p - 1) * rm_stride + n * H;
                unsigned long long wu = (unsigned long long)p << 32, wd = (unsigned long long)p << 32;   // rows outside the image: done, maximum 0
                while (!ok) {
                    if (yu >= 0) wu = __hip_atomic_load(rm + max(yu, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (yd < H) wd = __hip_atomic_load(rm + min(yd, H - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (int)(wu >> 32) == p && (int)(wd >> 32) == p;
                    if (!ok) {
                        ++spins;
                        const bool timed_out = spins > a.spin_limit;
                        const bool peer_gone = !timed_out && (spins & 63) == 0 && __hip_atomic_load(a.sticky + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (timed_out || peer_gone) {
                            if (lane == 0) {
                                s_abort = 1;
                                if (timed_out) { atomicAdd(a.sticky, 1); __hip_atomic_store(a.sticky + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (ok && lane == 0) s_m3 = fmaxf(m_own, fmaxf(finite_abs(__uint_as_float((unsigned)wu)), finite_abs(__uint_as_float((unsigned)wd))));
            }
            __syncthreads();
            if (*(volatile int *)&s_abort) {
                // (uniform after the barrier.)  As in the bf16 run: this row stops here and voids its row of the run's result
                float *o = a.out[a.npairs - 1] + (long)n * C * plane + (long)y * W;
                for (int i = tid; i < C * W; i += 256 * KS) o[(long)(i / W) * plane + (i % W)] = __uint_as_float(0x7fc00000u);
                return;
            }
            CH_MARK(1);
            m3 = s_m3;
            const float inv = 1.f / f16_scale_of(m3);
            (void)stage_load(a.out[p - 1], dA, std::false_type{}, true);
            stage_store(std::false_type{}, inv);
            if (co_ok && kpart == 0) {   // the own row, kept in registers since the previous pair's epilogue
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) pair_split2h(own[4 * g8 + 2 * e] * inv, own[4 * g8 + 2 * e + 1] * inv, pk[0][e], pk[1][e]);
                    const int grp = cg * 4 + g8;
                    if (grp * 8 < C) {
                        const int entry = (grp * 3 + 1) * W + px;   // (chunk * 2 + k half) = grp, row slot t = 1
                        *reinterpret_cast<u32x2 *>(s_in + entry * 16 + half * 8) = u32x2{pk[0][0], pk[0][1]};
                        *reinterpret_cast<u32x2 *>(s_in + in_piece + entry * 16 + half * 8) = u32x2{pk[1][0], pk[1][1]};
                    }
                }
            }
        }
        CH_MARK(2);
        __syncthreads();
        CH_MARK(3);
        const float sx = f16_scale_of(m3);
        // |relu(conv + bias)| <= (largest input) x (largest L1 norm of a filter) + largest |bias|: the intermediate row's scale without a reduction
        const float smid = f16_scale_of(fminf(fmaf(m3, l1A, fmaxf(s_bmax[0], s_bmax[1])), 3.0e38f));
        const float fA = sx * swA, inv_mid = 1.f / smid, fB = smid * swB;

        // ---- phase A: vertical taps
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (co_ok) {
            for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int ch = ch_lo + j0 + i;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        u32x4 b[2];
                        const int entry = ((ch * 2 + half) * 3 + t) * W + px;
                        b[0] = *reinterpret_cast<const u32x4 *>(s_in + entry * 16);
                        b[1] = *reinterpret_cast<const u32x4 *>(s_in + in_piece + entry * 16);
                        mma3(acc, wr[i][t], b);
                    }
                    const int nxt = j0 + i + R;
                    const unsigned char *wsrc = nxt < nch ? wA : wB;
                    const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, wr[i][t]);
                }
            }
        }
        CH_MARK(4);
        if constexpr (KS == 2) {
            if (kpart == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
            }
        }
        if (co_ok && kpart == 0) {
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                unsigned pk[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cch = min(cg * 32 + 8 * g8 + 4 * half + 2 * e, C - 2);
                    const float v0 = fmaf(acc[4 * g8 + 2 * e], fA, s_epi[0][cch]), v1 = fmaf(acc[4 * g8 + 2 * e + 1], fA, s_epi[0][cch + 1]);
                    pair_split2h((v0 > 0.f ? v0 : 0.f) * inv_mid, (v1 > 0.f ? v1 : 0.f) * inv_mid, pk[0][e], pk[1][e]);
                }
                const int grp = cg * 4 + g8;
                if (grp * 8 < C) {
                    *reinterpret_cast<u32x2 *>(s_mid + (grp * WM + dbmax + px) * 16 + half * 8) = u32x2{pk[0][0], pk[0][1]};
                    *reinterpret_cast<u32x2 *>(s_mid + mid_piece + (grp * WM + dbmax + px) * 16 + half * 8) = u32x2{pk[1][0], pk[1][1]};
                }
            }
        }
        __syncthreads();

        CH_MARK(5);
        // ---- phase B: horizontal taps over the intermediate; the ring is refilled with the NEXT pair's first vertical fragments
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *wNext = a.wA[last ? p : p + 1];
        if (co_ok) {
            for (int j0 = 0; j0 < nch; j0 += R) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int ch = ch_lo + j0 + i;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        u32x4 b[2];
                        const int entry = (ch * 2 + half) * WM + dbmax + px + (t - 1) * dB;
                        b[0] = *reinterpret_cast<const u32x4 *>(s_mid + entry * 16);
                        b[1] = *reinterpret_cast<const u32x4 *>(s_mid + mid_piece + entry * 16);
                        mma3(acc, wr[i][t], b);
                    }
                    const int nxt = j0 + i + R;
                    const unsigned char *wsrc = nxt < nch ? wB : wNext;
                    const int nch_src = nxt < nch ? ch_lo + nxt : ch_lo + nxt - nch;
#pragma unroll
                    for (int t = 0; t < 3; ++t) load_w(wsrc, t, nch_src, wr[i][t]);
                }
            }
        }
        if constexpr (KS == 2) {
            if (kpart == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[(wid * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += s_red[(wid * 16 + r) * 64 + lane];
            }
        }
        // ---- epilogue: bias, BatchNorm, block residual, ReLU; the row goes out write-through, stays in registers for the next pair, and
        //      leaves its largest finite magnitude
        CH_MARK(6);
        float lm = 0.f;
        if (co_ok && kpart == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, co_ = min(co, C - 1);
                float vv = fmaf(fmaf(acc[r], fB, s_epi[1][co_]), s_epi[2][co_], s_epi[3][co_]);
                if (a.res[p]) vv += blk_in[r];
                if (a.relu[p]) vv = vv > 0.f ? vv : 0.f;
                own[r] = co < C ? vv : 0.f;
                lm = fmaxf(lm, finite_abs(own[r]));
            }
            float *yo = a.out[p];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < C) {
                    if (last) yo[base + co * plane] = own[r];
                    else __hip_atomic_store(yo + base + co * plane, own[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!last && a.res[p]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) blk_in[r] = own[r];
            }
        }
        if (last) break;
        if (kpart == 0) {
            lm = wave_finite_absmax(lm);
            if (lane == 0) s_wmax[wid] = lm;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's row stores have left (write-through): maximum and counter may follow
        __syncthreads();
        if (wid8 == 0) {
            m_own = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
            if (lane == 0)
                __hip_atomic_store(fa.rowmax + (long)p * rm_stride + n * H + y, ((unsigned long long)(p + 1) << 32) | __float_as_uint(m_own), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        CH_MARK(7);
        if (*(volatile int *)&s_abort) return;
    }
    if constexpr (TRACE) {
        CH_MARK(7);
        tr_acc[0] = clock64() - tr_t0;
        if (tid == 0)
            for (int i = 0; i < 8; ++i) a.trace[(long)blockIdx.x * 8 + i] = tr_acc[i];
    }
#undef CH_MARK
}

// fp32 floats of the exact packing of one convolution of a pair
inline size_t pair_f32_floats(int channels) {
    const int CP = (channels + 31) / 32 * 32;
    return (size_t)3 * (channels / 16) * 2 * CP * 8;
}
inline size_t pair_split_bytes(int channels) {
    const int CP = (channels + 31) / 32 * 32;
    return (size_t)3 * (channels / 16) * (CP / 32) * 3 * 1024;
}
inline size_t pair_f16_bytes(int channels) {   // two fp16 pieces instead of three bf16 ones
    const int CP = (channels + 31) / 32 * 32;
    return (size_t)3 * (channels / 16) * (CP / 32) * 2 * 1024;
}
inline bool pair_use_split() {
    static const bool v = [] {
        const char *e = getenv("LAV_CONV_PRECISION");
        return !(e && (!strcmp(e, "f32") || !strcmp(e, "fp32")));
    }();
    return v;
}
}  // namespace

extern "C" size_t lav_conv1d_pair_packed_weight_floats(int channels) {
    if (channels < 16 || channels % 16) return 0;
    // the exact fp32 packing, the three-piece bf16 packing, the two-piece fp16 packing (round 6) and its tail {scale, largest L1 norm of a filter, 0, 0}
    return pair_f32_floats(channels) + pair_split_bytes(channels) / 4 + pair_f16_bytes(channels) / 4 + 4;
}

extern "C" int lav_conv1d_pair_pack_weights(int channels, const float *h_weight, float *h_packed) {
    LAV_REQUIRE(h_weight && h_packed, "lav_conv1d_pair_pack_weights: null");
    LAV_REQUIRE(channels >= 16 && channels % 16 == 0, "lav_conv1d_pair_pack_weights: channels must be a multiple of 16");
    const int C = channels, CP = (C + 31) / 32 * 32, nchunk = C / 16;
    // h_weight: PyTorch layout [cout][cin][3] (the singleton kernel dimension squeezed out)
    for (int t = 0; t < 3; ++t)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int half = 0; half < 2; ++half)
                for (int co = 0; co < CP; ++co)
                    for (int cp = 0; cp < 8; ++cp) {
                        const int k = ch * 16 + 2 * cp + half;
                        h_packed[((((size_t)t * nchunk + ch) * 2 + half) * CP + co) * 8 + cp] = co < C ? h_weight[((size_t)co * C + k) * 3 + t] : 0.f;
                    }
    // split packing: [tap][chunk][cout block][piece][lane = khalf*32 + cout%32][8 channels] bf16
    unsigned short *o = reinterpret_cast<unsigned short *>(h_packed + pair_f32_floats(C));
    auto bf = [](float x, float &rest) {
        unsigned u;
        memcpy(&u, &x, 4);
        const unsigned r = u + 0x8000u;
        u = ((r & 0x7f800000u) == 0x7f800000u ? u : r) & 0xffff0000u;
        float b;
        memcpy(&b, &u, 4);
        rest = x - b;
        return (unsigned short)(u >> 16);
    };
    for (int t = 0; t < 3; ++t)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int blk = 0; blk < CP / 32; ++blk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + e;
                        const float w = co < C ? h_weight[((size_t)co * C + ci) * 3 + t] : 0.f;
                        float r1, r2, r3;
                        const unsigned short p0 = bf(w, r1), p1 = bf(r1, r2), p2 = bf(r2, r3);
                        const size_t frag = ((((size_t)t * nchunk + ch) * (CP / 32) + blk) * 3) * 512;
                        o[frag + lane * 8 + e] = p0; o[frag + 512 + lane * 8 + e] = p1; o[frag + 1024 + lane * 8 + e] = p2;
                    }
    // fp16 packing (lav_conv1d_pair_chain_f16): [tap][chunk][cout block][piece 2][lane][8 channels] fp16 of w / s, s = the power of two that
    // puts the largest |w| into [16384, 32768); behind it {s, max over cout of sum |w[cout]|, 0, 0}
    {
        const size_t nw = (size_t)C * C * 3;
        float m = 0.f, l1 = 0.f;
        for (size_t i = 0; i < nw; ++i) { const float v = fabsf(h_weight[i]); if (v <= 3.4028235e38f && v > m) m = v; }
        for (int co = 0; co < C; ++co) {
            double acc = 0.0;
            for (int k = 0; k < C * 3; ++k) { const float v = fabsf(h_weight[(size_t)co * C * 3 + k]); if (v <= 3.4028235e38f) acc += v; }
            l1 = std::max(l1, (float)(acc * (1.0 + 1e-6)));
        }
        int ex = 0;
        (void)frexpf(m, &ex);
        const float sw = ldexpf(1.f, m > 0.f ? std::max(ex, -100) - 15 : 0), inv = 1.f / sw;
        _Float16 *h = reinterpret_cast<_Float16 *>(h_packed + pair_f32_floats(C) + pair_split_bytes(C) / 4);
        for (int t = 0; t < 3; ++t)
            for (int ch = 0; ch < nchunk; ++ch)
                for (int blk = 0; blk < CP / 32; ++blk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + e;
                            const float w = co < C ? h_weight[((size_t)co * C + ci) * 3 + t] * inv : 0.f;
                            const _Float16 h0 = (_Float16)w, h1 = (_Float16)(w - (float)h0);
                            const size_t frag = ((((size_t)t * nchunk + ch) * (CP / 32) + blk) * 2) * 512;
                            h[frag + lane * 8 + e] = h0; h[frag + 512 + lane * 8 + e] = h1;
                        }
        float *tail = h_packed + pair_f32_floats(C) + pair_split_bytes(C) / 4 + pair_f16_bytes(C) / 4;
        tail[0] = sw; tail[1] = l1; tail[2] = 0.f; tail[3] = 0.f;
    }
    return LAV_OK;
}

extern "C" size_t lav_conv1d_pair_lds_bytes(int channels, int w, int d_b) {
    if (pair_use_split()) return (size_t)channels * w * 18 + (size_t)channels * (w + 2 * d_b) * 6;   // 16-byte entries of 8 channels, 3 pieces
    return ((size_t)channels * 3 * w + (size_t)channels * (w + 2 * d_b)) * sizeof(float);
}

extern "C" int lav_conv1d_pair(int batch, int channels, int h, int w, int d_a, int d_b, const float *x, const float *wa_packed,
                               const float *bias_a, const float *wb_packed, const float *bias_b, const float *scale,
                               const float *shift, const float *residual, int relu_post, float *y, void *stream) {
    LAV_REQUIRE(batch >= 1 && h >= 1 && d_a >= 1 && d_b >= 1 && d_b <= 16, "lav_conv1d_pair: bad sizes (dilation of the 1x3 convolution at most 16)");
    LAV_REQUIRE(w == 32 || w == 64 || w == 128, "lav_conv1d_pair: row width %d not in {32, 64, 128}", w);
    LAV_REQUIRE(channels >= 16 && channels % 16 == 0 && (channels + 31) / 32 <= 128 / w,
                "lav_conv1d_pair: %d channels do not fit a %d-pixel row tile (at most %d)", channels, w, 32 * (128 / w));
    LAV_REQUIRE(x && wa_packed && bias_a && wb_packed && bias_b && y, "lav_conv1d_pair: null argument");
    LAV_REQUIRE((scale == nullptr) == (shift == nullptr), "lav_conv1d_pair: scale and shift come together");
    const size_t lds = lav_conv1d_pair_lds_bytes(channels, w, d_b);
    LAV_REQUIRE(lds <= 160 * 1024, "lav_conv1d_pair: %zu bytes of LDS needed", lds);
    static float *zero_page = nullptr;
    if (!zero_page) {
        LAV_HIP(hipMalloc(reinterpret_cast<void **>(&zero_page), 256));
        LAV_HIP(hipMemset(zero_page, 0, 256));
    }
    if (pair_use_split()) {
        PairSplitArgs sa;
        sa.x = x; sa.bA = bias_a; sa.bB = bias_b; sa.scale = scale; sa.shift = shift; sa.res = residual; sa.y = y;
        sa.wA = reinterpret_cast<const unsigned char *>(wa_packed + pair_f32_floats(channels));
        sa.wB = reinterpret_cast<const unsigned char *>(wb_packed + pair_f32_floats(channels));
        sa.B = batch; sa.C = channels; sa.H = h; sa.W = w; sa.dA = d_a; sa.dB = d_b; sa.relu_post = relu_post;
        sa.CP = (channels + 31) / 32 * 32;
        hipStream_t sst = static_cast<hipStream_t>(stream);
        const int ks2 = channels >= 64 && (channels / 16) % 2 == 0 ? 2 : 1;   // K halves of equal whole chunk counts only (80 channels: one wave group)
        const int nch2 = channels / 16 / ks2;
        const int ring2 = nch2 % 4 == 0 ? 4 : nch2 % 2 == 0 ? 2 : 1;
        const int tok2 = timer_begin("conv1d_pair", sst);
#define LAV_PAIRS_CASE(KS_, R_) if (ks2 == KS_ && ring2 == R_) { \
        static bool attr = false; \
        if (!attr) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_split<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
        hipLaunchKernelGGL((k_conv1d_pair_split<KS_, R_>), dim3(batch * h), dim3(256 * KS_), lav::lds_claim(lds), sst, sa); }
        LAV_PAIRS_CASE(1, 1) LAV_PAIRS_CASE(1, 2) LAV_PAIRS_CASE(1, 4) LAV_PAIRS_CASE(2, 1) LAV_PAIRS_CASE(2, 2) LAV_PAIRS_CASE(2, 4)
#undef LAV_PAIRS_CASE
        timer_end(tok2, sst);
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    }
    PairArgs a;
    a.x = x; a.wA = wa_packed; a.bA = bias_a; a.wB = wb_packed; a.bB = bias_b; a.scale = scale; a.shift = shift; a.res = residual;
    a.zero_page = zero_page; a.y = y;
    a.trace = nullptr;
    static const bool want_trace = getenv("LAV_PAIR_TRACE") != nullptr;
    static unsigned long long *d_trace = nullptr;
    static int runs = 0;
    if (want_trace && batch * h <= 8192) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)8192 * 8 * sizeof(unsigned long long)));
        a.trace = d_trace;
    }
    a.B = batch; a.C = channels; a.H = h; a.W = w; a.dA = d_a; a.dB = d_b; a.relu_post = relu_post;
    a.CP = (channels + 31) / 32 * 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("conv1d_pair", st);
    // rows are few (one workgroup each, at most one per CU for ERFNet's shapes): put 8 waves on the channel loop when it
    // is long enough to halve (the partial sums need 16 KB of the staging area)
    static const bool no_split = getenv("LAV_PAIR_KSPLIT") && getenv("LAV_PAIR_KSPLIT")[0] == '0';
    const int ks = (channels >= 64 && (channels / 16) % 2 == 0 && !no_split && (size_t)channels * 3 * w * sizeof(float) >= 16 * 1024) ? 2 : 1;
    const int nch = channels / 16 / ks;              // chunks per wave
    const int ring = nch % 4 == 0 ? 4 : nch % 2 == 0 ? 2 : 1;
#define LAV_PAIR_CASE(KS_, R_) if (ks == KS_ && ring == R_) { \
        static bool attr = false; \
        if (!attr) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
        hipLaunchKernelGGL((k_conv1d_pair<KS_, R_>), dim3(batch * h), dim3(256 * KS_), lav::lds_claim(lds), st, a); }
    LAV_PAIR_CASE(1, 1) LAV_PAIR_CASE(1, 2) LAV_PAIR_CASE(1, 4) LAV_PAIR_CASE(2, 1) LAV_PAIR_CASE(2, 2) LAV_PAIR_CASE(2, 4)
#undef LAV_PAIR_CASE
    if (a.trace && ++runs % 10 == 0) {   // debug: per-workgroup phase times of every 10th launch
        const size_t nwg = (size_t)batch * h;
        std::vector<unsigned long long> hst(nwg * 8);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hst.data(), d_trace, hst.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull, t1 = 0;
            double ph[6] = {0, 0, 0, 0, 0, 0}, last_start = 0;
            for (size_t i = 0; i < nwg; ++i) {
                t0 = std::min(t0, hst[i * 8]); t1 = std::max(t1, hst[i * 8 + 6]);
            }
            for (size_t i = 0; i < nwg; ++i) {
                for (int k = 0; k < 6; ++k) ph[k] += (double)(hst[i * 8 + k + 1] - hst[i * 8 + k]) / 100.0;
                last_start = std::max(last_start, (double)(hst[i * 8] - t0) / 100.0);
            }
            fprintf(stderr, "[pair trace] C %d W %d d %d/%d ks %d ring %d, %zu wgs: span %.2f us, last start %.2f | mean us: issue %.2f | dma wait %.2f | phase A %.2f | combine+mid %.2f | phase B %.2f | combine+epilogue %.2f\n",
                    channels, w, d_a, d_b, ks, ring, nwg, (double)(t1 - t0) / 100.0, last_start, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, ph[4] / nwg, ph[5] / nwg);
        }
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_zero_ints(int *p, int n, int *q, int nq, int *abort_word) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
    else if (i - n < nq) q[i - n] = 0;   // (the fp16 runs' count-and-maximum words)
    if (i == 0) *abort_word = 0;   // sticky[2]: the abort word of the launches that follow
}

// lav_conv1d_pair_chain_region: where this thread's next runs keep their counters, and what they clean first
thread_local int g_chain_row_offset = 0, g_chain_clean_rows = -1;
inline size_t chain_cap_rows(size_t workspace_bytes) { return workspace_bytes > 256 ? (workspace_bytes - 256) / 132 / 64 * 64 : 0; }
}  // namespace

extern "C" size_t lav_conv1d_pair_chain_workspace_bytes(int batch, int h) {
    // sticky counters | per-row progress counters (capacity rounded up to 64 rows) | (fp16 run) every pair's per-row count-and-maximum words
    if (batch <= 0 || h <= 0) return 0;
    const size_t cap = ((size_t)batch * h + 63) / 64 * 64;
    return 256 + cap * 132;
}

extern "C" int lav_conv1d_pair_chain_region(int row_offset, int clean_rows) {
    LAV_REQUIRE(row_offset >= 0 && clean_rows >= -1 && (clean_rows >= 0 || row_offset == 0), "lav_conv1d_pair_chain_region: offset >= 0, clean_rows >= 0 (or -1 at offset 0: the default)");
    g_chain_row_offset = row_offset; g_chain_clean_rows = clean_rows;
    return LAV_OK;
}

extern "C" size_t lav_conv1d_pair_chain_lds_bytes(int channels, int w, int d_b_max) {
    return 16384 + (size_t)channels * w * 18 + (size_t)channels * (w + 2 * d_b_max) * 6;
}

static int pair_chain_launch(bool f16, int batch, int channels, int h, int w, int npairs, const int *d_a, const int *d_b, const int *residual,
                             const int *relu_post, const float *x, const float *const *wa_packed, const float *const *bias_a,
                             const float *const *wb_packed, const float *const *bias_b, const float *const *scale,
                             const float *const *shift, float *const *out, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(pair_use_split(), "lav_conv1d_pair_chain: the persistent run exists for the bf16x6 kernels only (LAV_CONV_PRECISION=f32 runs the pairs one launch each)");
    LAV_REQUIRE(batch >= 1 && h >= 1 && npairs >= 1 && npairs <= CHAIN_MAX, "lav_conv1d_pair_chain: 1..%d pairs", CHAIN_MAX);
    LAV_REQUIRE(w == 32 || w == 64 || w == 128, "lav_conv1d_pair_chain: row width %d not in {32, 64, 128}", w);
    LAV_REQUIRE(channels >= 16 && channels % 16 == 0 && (channels + 31) / 32 <= 128 / w,
                "lav_conv1d_pair_chain: %d channels do not fit a %d-pixel row tile (at most %d)", channels, w, 32 * (128 / w));
    LAV_REQUIRE(d_a && d_b && residual && relu_post && x && wa_packed && bias_a && wb_packed && bias_b && scale && shift && out, "lav_conv1d_pair_chain: null argument");
    LAV_REQUIRE(workspace && workspace_bytes >= lav_conv1d_pair_chain_workspace_bytes(batch, h), "lav_conv1d_pair_chain: workspace too small");
    // every row's workgroup must be resident at once (they wait for each other): one workgroup per CU at these LDS sizes
    static const int cus = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    {   // a CU takes two of the run's workgroups when they are 256 threads with at most 76 KB of LDS each (16-channel stages)
        int dbm = 1;
        for (int i = 0; i < npairs; ++i) dbm = std::max(dbm, d_b[i]);
        const bool two = lav_conv1d_pair_chain_lds_bytes(channels, w, dbm) <= 76 * 1024 && !(channels >= 64 && (channels / 16) % 2 == 0);
        LAV_REQUIRE(batch * h <= cus * (two ? 2 : 1), "lav_conv1d_pair_chain: %d rows exceed the %d workgroups the chip holds at once (run the pairs one launch each)",
                    batch * h, cus * (two ? 2 : 1));
    }
    PairChainArgs a;
    int dbmax = 1;
    for (int i = 0; i < CHAIN_MAX; ++i) {
        const int j = i < npairs ? i : npairs - 1;
        LAV_REQUIRE(d_a[j] >= 1 && d_b[j] >= 1 && d_b[j] <= 16 && d_a[j] < 256, "lav_conv1d_pair_chain: bad dilation");
        LAV_REQUIRE(wa_packed[j] && bias_a[j] && wb_packed[j] && bias_b[j] && scale[j] && shift[j] && out[j], "lav_conv1d_pair_chain: null argument of pair %d", j);
        a.out[i] = out[j];
        a.wA[i] = reinterpret_cast<const unsigned char *>(wa_packed[j] + pair_f32_floats(channels));
        a.wB[i] = reinterpret_cast<const unsigned char *>(wb_packed[j] + pair_f32_floats(channels));
        a.bA[i] = bias_a[j]; a.bB[i] = bias_b[j]; a.scale[i] = scale[j]; a.shift[i] = shift[j];
        a.dA[i] = (unsigned char)d_a[j]; a.dB[i] = (unsigned char)d_b[j]; a.res[i] = residual[j] ? 1 : 0; a.relu[i] = relu_post[j] ? 1 : 0;
        dbmax = std::max(dbmax, d_b[j]);
    }
    LAV_REQUIRE(!residual[0], "lav_conv1d_pair_chain: the run starts at a block boundary (its first pair has no residual)");
    const size_t cap_rows = chain_cap_rows(workspace_bytes);
    const int row_off = g_chain_row_offset, clean_rows = g_chain_clean_rows;
    LAV_REQUIRE((size_t)row_off + (size_t)batch * h <= cap_rows && (clean_rows < 0 || (size_t)clean_rows <= cap_rows),
                "lav_conv1d_pair_chain: rows %d + %d exceed the workspace's %zu counters (lav_conv1d_pair_chain_region)", row_off, batch * h, cap_rows);
    a.x0 = x; a.sticky = static_cast<int *>(workspace); a.flags = a.sticky + 64 + row_off;
    a.B = batch; a.C = channels; a.H = h; a.W = w; a.CP = (channels + 31) / 32 * 32; a.npairs = npairs;
    const char *lim = getenv("LAV_CHAIN_SPIN_LIMIT");   // test knob: 0 makes every wait that is not satisfied at once a time-out
    a.spin_limit = lim ? std::max(0ll, atoll(lim)) : (1ll << 21);
    const size_t lds = lav_conv1d_pair_chain_lds_bytes(channels, w, dbmax);
    LAV_REQUIRE(lds <= 152 * 1024, "lav_conv1d_pair_chain: %zu bytes of LDS needed", lds);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nflag = batch * h;
    // counters at zero: this run's own (default), the first `clean_rows` of the array (the first of several runs that share one
    // cleaning), or nothing (clean_rows = 0: a launch earlier on the stream cleaned this run's region)
    {
        // (the fp16 runs hand off through their count-and-maximum words - 16 pairs x capacity rows x 8 bytes behind the counters, cleaned as a whole)
        int *words = reinterpret_cast<int *>(static_cast<char *>(workspace) + 256 + cap_rows * sizeof(int));
        const int nwords = f16 ? (int)(CHAIN_MAX * cap_rows * 2) : 0;
        if (clean_rows < 0) hipLaunchKernelGGL(k_zero_ints, dim3((nflag + nwords + 255) / 256), dim3(256), 0, st, a.flags, nflag, words, nwords, a.sticky + 2);
        else if (clean_rows > 0) hipLaunchKernelGGL(k_zero_ints, dim3((clean_rows + nwords + 255) / 256), dim3(256), 0, st, a.sticky + 64, clean_rows, words, nwords, a.sticky + 2);
    }
    const int ks2 = channels >= 64 && (channels / 16) % 2 == 0 ? 2 : 1;
    const int nch2 = channels / 16 / ks2;
    // weight ring of at most two chunks: the four-chunk ring of the single-pair kernel does not fit the registers next to the
    // block's input row and the next pair's prefetch (364 bytes of scratch per lane)
    const int ring2 = nch2 % 2 == 0 ? 2 : 1;
    const int tok = timer_begin("conv1d_pair", st);
    if (f16) {
        PairChainF16Args fa;
        fa.c = a;
        fa.c.trace = nullptr;
        const size_t sec = pair_f32_floats(channels) + pair_split_bytes(channels) / 4;   // floats in front of the fp16 section
        for (int i = 0; i < CHAIN_MAX; ++i) {
            const int j = i < npairs ? i : npairs - 1;
            fa.c.wA[i] = reinterpret_cast<const unsigned char *>(wa_packed[j] + sec);
            fa.c.wB[i] = reinterpret_cast<const unsigned char *>(wb_packed[j] + sec);
            fa.tA[i] = wa_packed[j] + sec + pair_f16_bytes(channels) / 4;
            fa.tB[i] = wb_packed[j] + sec + pair_f16_bytes(channels) / 4;
        }
        fa.rowmax = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + 256 + cap_rows * sizeof(int)) + row_off;
        fa.rm_stride = (int)cap_rows;
        fa.dbmax = dbmax;
        const size_t lds16 = 16384 + (size_t)channels * w * 12 + (size_t)channels * (w + 2 * dbmax) * 4;
        constexpr size_t F16_STATIC_LDS = 2064 + 64;   // s_abort, s_epi, s_wmax, s_bmax, s_m3
#define LAV_CHAIN16_CASE(KS_, R_) if (ks2 == KS_ && ring2 == R_) { \
        static size_t cap = 0; \
        if (!cap) { \
            cap = (160 * 1024 - F16_STATIC_LDS) / 16 * 16; \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain_f16<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess) { \
                (void)hipGetLastError(); \
                cap = 152 * 1024; \
                LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain_f16<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap)); \
            } \
        } \
        /* (the residency rule above was checked with the bf16 run's LDS size: this run claims what THAT one would, so the same rows per CU) */ \
        hipLaunchKernelGGL((k_conv1d_pair_chain_f16<KS_, R_>), dim3(batch * h), dim3(256 * KS_), \
                           std::min(cap, std::max(lds16, lav::lds_claim(lds, F16_STATIC_LDS, batch * h > cus))), st, fa); }
        static const bool want_trace16 = getenv("LAV_PAIR_CHAIN_TRACE") != nullptr;
        if (want_trace16 && ks2 == 2 && ring2 == 2 && batch * h <= 512) {
            static long long *d_trace16 = nullptr;
            if (!d_trace16) LAV_HIP(hipMalloc(&d_trace16, (size_t)512 * 8 * sizeof(long long)));
            fa.c.trace = d_trace16;
            static bool attr_t = false;
            if (!attr_t) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain_f16<2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_t = true; }
            hipLaunchKernelGGL((k_conv1d_pair_chain_f16<2, 2, true>), dim3(batch * h), dim3(512), std::max(lds16, lav::lds_claim(lds, F16_STATIC_LDS, batch * h > cus)), st, fa);
            static int runs16 = 0;
            if (++runs16 % 10 == 0) {
                std::vector<long long> hst((size_t)batch * h * 8);
                if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hst.data(), d_trace16, hst.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int i = 0; i < batch * h; ++i) for (int k = 0; k < 8; ++k) m[k] += (double)hst[(size_t)i * 8 + k];
                    const double f = 1.0 / (batch * h) / npairs;
                    fprintf(stderr, "[pair chain f16 trace] C %d W %d, %d rows, %d pairs: cycles per pair and workgroup: all %.0f | hand-off wait %.0f | neighbour + own rows %.0f | barrier %.0f | phase A %.0f | combine+mid %.0f | phase B %.0f | epilogue+publish %.0f\n",
                            channels, w, batch * h, npairs, m[0] * f, m[1] * f, m[2] * f, m[3] * f, m[4] * f, m[5] * f, m[6] * f, m[7] * f);
                }
            }
        } else {
        LAV_CHAIN16_CASE(1, 1) LAV_CHAIN16_CASE(1, 2) LAV_CHAIN16_CASE(2, 1) LAV_CHAIN16_CASE(2, 2)
        }
#undef LAV_CHAIN16_CASE
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    }
#define LAV_CHAIN_CASE(KS_, R_) if (ks2 == KS_ && ring2 == R_) { \
        static size_t cap = 0;   /* (the kernel also holds CHAIN_STATIC_LDS bytes of static LDS: the dynamic part may claim 160 KB minus that) */ \
        if (!cap) { \
            cap = (160 * 1024 - CHAIN_STATIC_LDS) / 16 * 16; \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess) { \
                (void)hipGetLastError(); \
                cap = 152 * 1024; \
                LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain<KS_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap)); \
            } \
            if (getenv("LAV_PAIR_CHAIN_DEBUG")) fprintf(stderr, "[pair chain] dynamic LDS cap %zu bytes, this launch %zu\n", cap, std::min(cap, lav::lds_claim(lds, CHAIN_STATIC_LDS, batch * h > cus))); \
        } \
        hipLaunchKernelGGL((k_conv1d_pair_chain<KS_, R_>), dim3(batch * h), dim3(256 * KS_), std::min(cap, lav::lds_claim(lds, CHAIN_STATIC_LDS, batch * h > cus)), st, a); }
    static const bool want_trace = getenv("LAV_PAIR_CHAIN_TRACE") != nullptr;
    static long long *d_trace = nullptr;
    a.trace = nullptr;
    if (want_trace && ks2 == 2 && ring2 == 2) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)512 * 8 * sizeof(long long)));
        a.trace = d_trace;
        static bool attr_t = false;
        if (!attr_t) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv1d_pair_chain<2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_t = true; }
        hipLaunchKernelGGL((k_conv1d_pair_chain<2, 2, true>), dim3(batch * h), dim3(512), lds, st, a);   // (trace build: exact size)
        static int runs = 0;
        if (++runs % 10 == 0 && batch * h <= 512) {
            std::vector<long long> hst((size_t)batch * h * 8);
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hst.data(), d_trace, hst.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < batch * h; ++i) for (int k = 0; k < 8; ++k) m[k] += (double)hst[(size_t)i * 8 + k];
                const double f = 1.0 / (batch * h) / npairs;
                fprintf(stderr, "[pair chain trace] C %d W %d, %d rows, %d pairs: cycles per pair and workgroup: all %.0f | counter wait %.0f | neighbour rows %.0f | halo+barrier %.0f | phase A %.0f | combine+mid %.0f | phase B %.0f | epilogue+publish %.0f\n",
                        channels, w, batch * h, npairs, m[0] * f, m[1] * f, m[2] * f, m[3] * f, m[4] * f, m[5] * f, m[6] * f, m[7] * f);
            }
        }
    } else {
    LAV_CHAIN_CASE(1, 1) LAV_CHAIN_CASE(1, 2) LAV_CHAIN_CASE(2, 1) LAV_CHAIN_CASE(2, 2)
    }
#undef LAV_CHAIN_CASE
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_conv1d_pair_chain(int batch, int channels, int h, int w, int npairs, const int *d_a, const int *d_b, const int *residual,
                                     const int *relu_post, const float *x, const float *const *wa_packed, const float *const *bias_a,
                                     const float *const *wb_packed, const float *const *bias_b, const float *const *scale,
                                     const float *const *shift, float *const *out, void *workspace, size_t workspace_bytes, void *stream) {
    return pair_chain_launch(false, batch, channels, h, w, npairs, d_a, d_b, residual, relu_post, x, wa_packed, bias_a, wb_packed, bias_b, scale, shift, out,
                             workspace, workspace_bytes, stream);
}

extern "C" int lav_conv1d_pair_chain_f16(int batch, int channels, int h, int w, int npairs, const int *d_a, const int *d_b, const int *residual,
                                         const int *relu_post, const float *x, const float *const *wa_packed, const float *const *bias_a,
                                         const float *const *wb_packed, const float *const *bias_b, const float *const *scale,
                                         const float *const *shift, float *const *out, void *workspace, size_t workspace_bytes, void *stream) {
    return pair_chain_launch(true, batch, channels, h, w, npairs, d_a, d_b, residual, relu_post, x, wa_packed, bias_a, wb_packed, bias_b, scale, shift, out,
                             workspace, workspace_bytes, stream);
}

extern "C" int lav_conv1d_pair_chain_status(const void *workspace, int *h_timeouts_launches2, void *stream) {
    LAV_REQUIRE(workspace && h_timeouts_launches2, "lav_conv1d_pair_chain_status: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LAV_HIP(hipMemcpyAsync(h_timeouts_launches2, workspace, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LAV_HIP(hipStreamSynchronize(st));
    return LAV_OK;
}

// Weight gradient of the training graph's 3x3 convolutions (stride 1 and 2, padding 1) and of its 7x7 stride-2 stems on the bf16 matrix
// cores (round 5).
//
// Replaces the weight-gradient half of torch.autograd's convolution backward (MIOpen igemm_wrw: the largest kernel family of a
// train_full step, profiles/r04_train_full_kernel_top.txt) for the layers of LAV.train_lidar that carry the convolution time:
// ConvBackbone's stage convolutions (team_code_v2/models/lidar.py:57-108, reference lav/lav_final_v2.py:140-259 backward) and the fused
// heads convolution 384 -> 4 x 64 (lidar.py:147-161).
//
//     dW[co][ci][ky][kx] = sum over (n, y, x) of dY[n][co][y][x] * X[n][ci][y + ky - 1][x + kx - 1]          (zero padding)
//
// A GEMM per tap with K = pixels.  fp32 operands are split exactly into three bf16 pieces each and the six leading partial products
// run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation ("bf16x6", conv_split.hpp: not less accurate than an fp32 fmaf chain).
//
//   workgroup   8 waves.  Waves 0-3 own the accumulators of a 64 co x 64 ci tile for all nine taps (wave = (co half, ci half): 9 x 16
//               registers) and do nothing but ds_read_b128 + matrix instructions; waves 4-7 load, split and stage.
//   K walk      a task = (tile, image, block of rows); it walks its rows top to bottom inside one 16-pixel column segment after the
//               other.  One STEP = 16 output pixels of one row = one k-block of the matrix instruction.  Input rows y-1, y, y+1 live in
//               a ring of four row slots: every step stages ONE new input row and ONE row of dY, each input row serves three steps.
//   staging     lane operands are 8 consecutive pixels (16 bytes of bf16).  The kx = 0 / 2 taps read the row shifted by one pixel,
//               which would be a 2-byte misaligned 16-byte LDS read: the loaders write THREE copies of a row, one per kx, each
//               aligned (the shift costs nothing at load time).  LDS: ring 4 x 18 KB + dY 2 x 6 KB = 84 KB.
//   sync        one LDS-only barrier per step (s_waitcnt lgkmcnt(0); s_barrier): the loaders' global loads for the step after next
//               stay in flight across it.
//   reduction   every task writes its 64 x 64 x 9 partial sums; lav_conv_wgrad's second launch adds the partials of a tile in slice
//               order (deterministic, no atomics) and writes dW in PyTorch layout.
//   stride 2    (the first convolution of every backbone stage: dY[n][co][oy][ox] * X[n][ci][2 oy + ky - 1][2 ox + kx - 1])  k_conv_wgrad_s2:
//               the same tile, step and accumulators; a step consumes input rows 2 oy - 1, 2 oy, 2 oy + 1, so the loaders stage TWO new
//               input rows per step into a ring of six slots, and the three kx copies of a row are its odd columns from 2 ox - 1, its even
//               columns, and its odd columns from 2 ox + 1 (the first and third out of the same loaded registers).  LDS 6 x 18 + 12 KB.
//   fp16 pieces (round 6, template parameter F16; lav_conv_wgrad_amax): both operands as TWO fp16 pieces of x / s_x and dY / s_dY - the
//               powers of two that put each tensor's largest finite magnitude into [16384, 32768), from maxima the caller hands in (the
//               measurements the forward / data-gradient convolutions of the same step already made) - and the three leading partial
//               products on v_mfma_f32_32x32x16_f16: half the matrix instructions, 2/3 of the LDS bytes, the partial sums multiplied back
//               by s_x and s_dY when they are written.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace {
using namespace lav;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

constexpr int T_CO = 64, T_CI = 64, PX = 16;            // tile of the weight gradient, pixels per step
constexpr int ENTRY = 16;                               // bytes of one lane operand (8 bf16 / fp16)
// sizes for NPC pieces per operand (3 bf16 / 2 fp16):
constexpr int row_copy(int npc) { return npc * 2 * T_CI * ENTRY; }          // one kx copy of an input row: [piece][k half 2][ci 64] entries = 6 / 4 KB
constexpr int row_slot(int npc) { return 3 * row_copy(npc); }               // three kx copies = 18 / 12 KB
constexpr int dy_buf(int npc) { return npc * 2 * T_CO * ENTRY; }            // [piece][k half 2][co 64] = 6 / 4 KB
constexpr int lds_bytes(int npc) { return 4 * row_slot(npc) + 2 * dy_buf(npc); }   // 84 / 56 KB
constexpr int PF = 3;                                   // steps the loaders' global loads run ahead of their staging

struct WgradArgs {
    const float *x, *dy;
    float *partial;     // [slice][tile][tap 9][co 64][ci 64]
    int B, cin, cout, H, W;
    int nseg;           // 16-pixel column segments of a row
    int rows_per_block, nblocks;   // a slice = (image, block of rows)
    int ntile_ci, ntiles;
    // F16: maxima of the finite |x| / |dY| in parts (lav_conv2d_amax's hand-off, lav_absmax_parts)
    const float *amax_x, *amax_dy;
    int n_amax_x, n_amax_dy;
};

// x = q0 + q1 + q2 exactly (three bf16 pieces of two values at once; conv_split.hpp: split3_pair)
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned &q0, unsigned &q1, unsigned &q2) {
    constexpr float M = 3.3895313892515355e38f;   // 0x7f7f0000: the first piece never rounds into the Inf exponent
    const float c0 = __builtin_amdgcn_fmed3f(x0, -M, M), c1 = __builtin_amdgcn_fmed3f(x1, -M, M);
    q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{c0, c1}, wg_bf16x2));
    const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
    q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{r0, r1}, wg_bf16x2));
    const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
    q2 = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{s0, s1}, wg_bf16x2));
}

typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// the pieces of two values (x0 in the low half): three bf16 pieces of x exactly, or two fp16 pieces of x * inv (conv_split_kernel.hpp: split2h_pair)
template <bool F16>
__device__ __forceinline__ void split_pair(float x0, float x1, float inv, unsigned (&q)[F16 ? 2 : 3]) {
    if constexpr (F16) {
        const float u0 = x0 * inv, u1 = x1 * inv;
        const wg_f16x2 h0 = __builtin_convertvector(wg_f32x2{u0, u1}, wg_f16x2);
        const wg_f32x2 f0 = __builtin_convertvector(h0, wg_f32x2);
        const wg_f16x2 h1 = __builtin_convertvector(wg_f32x2{u0 - f0[0], u1 - f0[1]}, wg_f16x2);
        q[0] = __builtin_bit_cast(unsigned, h0);
        q[1] = __builtin_bit_cast(unsigned, h1);
    } else {
        split3x2(x0, x1, q[0], q[1], q[2]);
    }
}
// acc += A . B over the operands' pieces: the six leading products of three bf16 pieces, or the three of two fp16 pieces
template <bool F16>
__device__ __forceinline__ void mma_pieces(f32x16 &acc, const u32x4 (&av)[F16 ? 2 : 3], const u32x4 (&bv)[F16 ? 2 : 3]) {
    if constexpr (F16) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[PA[k]]), __builtin_bit_cast(f16x8, bv[PB[k]]), acc, 0, 0, 0);
    } else {
        constexpr int PA[6] = {0, 0, 1, 0, 1, 2}, PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[PA[k]]), __builtin_bit_cast(bf16x8, bv[PB[k]]), acc, 0, 0, 0);
    }
}
// F16: the two operand scales of a task (every wave reads the parts itself)
template <bool F16>
__device__ __forceinline__ void task_scales(const WgradArgs &a, int lane, float &sx, float &sdy) {
    sx = 1.f; sdy = 1.f;
    if constexpr (F16) {
        sx = f16_scale_of(parts_absmax(a.amax_x, a.n_amax_x, lane));
        sdy = f16_scale_of(parts_absmax(a.amax_dy, a.n_amax_dy, lane));
    }
}

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool F16>
__global__ __launch_bounds__(512) void k_conv_wgrad(WgradArgs a) {
    constexpr int NPC = F16 ? 2 : 3, ROW_COPY = row_copy(NPC), ROW_SLOT = row_slot(NPC), DY_BUF = dy_buf(NPC);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_x = smem, *s_dy = smem + 4 * ROW_SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int t_co = tile / a.ntile_ci, t_ci = tile - t_co * a.ntile_ci;
    const int n = slice / a.nblocks, blk = slice - n * a.nblocks;
    const int y_lo = blk * a.rows_per_block, y_hi = min(a.H, y_lo + a.rows_per_block);
    const long plane = (long)a.H * a.W;
    const float *xn = a.x + ((long)n * a.cin + (long)t_ci * T_CI) * plane;
    const float *dyn = a.dy + ((long)n * a.cout + (long)t_co * T_CO) * plane;
    const int nrows = y_hi - y_lo;
    // a segment is walked as steps j = -2 .. nrows - 1: step j computes output row y_lo + j (j >= 0) while the loaders stage input row
    // y_lo + j + 2 and dY row y_lo + j + 1 for the steps that follow; j = -2, -1 only stage (rows y_lo - 1, y_lo and dY row y_lo)
    const int T = a.nseg * (nrows + 3), T_pad = (T + PF - 1) / PF * PF;   // iterations of the task; every role runs T_pad barriers
    float sx, sdy;
    task_scales<F16>(a, lane, sx, sdy);
    const float inv_x = 1.f / sx, inv_dy = 1.f / sdy;   // (powers of two: exact)
    if (wid >= 4) {
        // ------------------------------------------------------------------------------------------------ loaders
        const int lt = tid - 256;   // 0 .. 255; two items per thread and step.  Loader wave lw = 0, 1: copy kx = 0 of the input row, then
        //                             copy 2; lw = 2, 3: copy 1, then the dY row.  (ch, kh) = channel and k half of the 8-pixel entry.
        const int lw = wid - 4;                          // scalar
        const int ch = lt & 63, kh = lw & 1;
        const int kx0 = lw >> 1;                         // first item: copy kx0 of the input row
        const bool second_is_x = lw < 2;                 // second item: copy 2 of the input row, or dY
        // Loads run PF steps ahead of their staging (register sets, the loop unrolled by PF): one step of matrix work is ~0.7 us, a
        // load from HBM under this kernel's own traffic takes longer - with one step of distance every step waited for memory
        // (0.45 of the roof; profiles/r05_wgrad_probe.txt).  The sequence of steps runs on across the segments of the task.
        float4 va[PF][3], vb[PF][3];                     // raw loads of the two items (three aligned 16-byte pieces each)
        bool oka[PF][3], okb[PF][3];
        int i_seg = 0, i_j = -3, s_j = -3;               // the step the next issue / the next staging works on
        // window of an X item: pixels w0 .. w0 + 7 with w0 = x0 + 8 kh + kx - 1; aligned base a0 = w0 rounded down to 4
        auto issue = [&](float4 (&va)[3], float4 (&vb)[3], bool (&oka)[3], bool (&okb)[3]) __attribute__((always_inline)) {
            const int seg = min(i_seg, a.nseg - 1), j = i_j;   // (past the last step: a harmless reload, staged where nobody reads)
            const int x0 = seg * PX + 8 * kh;
            {   // first item: input row y_lo + j + 2, copy kx0
                const int row = y_lo + j + 2, a0 = kx0 == 0 ? x0 - 4 : x0;
                const bool rok = row >= 0 && row < a.H;
                const float *p = xn + (long)ch * plane + (long)min(max(row, 0), a.H - 1) * a.W;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = a0 + 4 * q;
                    oka[q] = rok && c >= 0 && c < a.W;
                    va[q] = *reinterpret_cast<const float4 *>(p + min(max(c, 0), a.W - 4));
                }
            }
            {   // second item: copy 2 of the same input row (base x0, window from x0 + 1), or the dY row y_lo + j + 1 (base x0)
                const int row = second_is_x ? y_lo + j + 2 : y_lo + j + 1;
                const bool rok = second_is_x ? (row >= 0 && row < a.H) : (row >= y_lo && row < y_hi);
                const float *p = (second_is_x ? xn : dyn) + (long)ch * plane + (long)min(max(row, 0), a.H - 1) * a.W;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = x0 + 4 * q;
                    okb[q] = rok && c < a.W;
                    vb[q] = *reinterpret_cast<const float4 *>(p + min(c, a.W - 4));
                }
            }
            if (++i_j == nrows) { i_j = -3; ++i_seg; }
        };
        // eight consecutive floats starting SH floats into the 12 loaded ones -> the 16-byte pieces at `dst` (piece stride ps)
        auto emit = [&](auto SH_, const float4 (&v)[3], const bool (&ok)[3], unsigned char *dst, int ps, float inv) __attribute__((always_inline)) {
            constexpr int SH = decltype(SH_)::value;
            float f[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                f[4 * q] = ok[q] ? v[q].x : 0.f; f[4 * q + 1] = ok[q] ? v[q].y : 0.f; f[4 * q + 2] = ok[q] ? v[q].z : 0.f; f[4 * q + 3] = ok[q] ? v[q].w : 0.f;
            }
            u32x4 pc[NPC];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned q[NPC];
                split_pair<F16>(f[SH + 2 * e], f[SH + 2 * e + 1], inv, q);
#pragma unroll
                for (int p = 0; p < NPC; ++p) pc[p][e] = q[p];
            }
#pragma unroll
            for (int p = 0; p < NPC; ++p) *reinterpret_cast<u32x4 *>(dst + p * ps) = pc[p];
        };
        using std::integral_constant;
        auto stage = [&](const float4 (&va)[3], const float4 (&vb)[3], const bool (&oka)[3], const bool (&okb)[3]) __attribute__((always_inline)) {
            const int j = s_j;
            const int slot = (j + 2 + 1) & 3;   // input row y_lo + j + 2 -> ring slot (relative row + 1) mod 4
            unsigned char *xd = s_x + slot * ROW_SLOT + (kh * T_CI + ch) * ENTRY;
            // the first item's window starts 3 floats into its aligned base for kx 0 (a0 = x0 - 4), 0 floats for kx 1
            if (kx0 == 0) emit(integral_constant<int, 3>{}, va, oka, xd, 2 * T_CI * ENTRY, inv_x);
            else emit(integral_constant<int, 0>{}, va, oka, xd + ROW_COPY, 2 * T_CI * ENTRY, inv_x);
            if (second_is_x) emit(integral_constant<int, 1>{}, vb, okb, xd + 2 * ROW_COPY, 2 * T_CI * ENTRY, inv_x);
            else emit(integral_constant<int, 0>{}, vb, okb, s_dy + ((j + 1) & 1) * DY_BUF + (kh * T_CO + ch) * ENTRY, 2 * T_CO * ENTRY, inv_dy);
            if (++s_j == nrows) s_j = -3;
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) issue(va[d], vb[d], oka[d], okb[d]);
        for (int t = 0; t < T_pad; t += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                stage(va[d], vb[d], oka[d], okb[d]);
                issue(va[d], vb[d], oka[d], okb[d]);
                barrier_lds();
            }
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------------ compute waves
    const int h = wid & 1, g = wid >> 1;               // co half, ci half of the tile
    const int l31 = lane & 31, kgrp = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a_off = (kgrp * T_CO + 32 * h + l31) * ENTRY, b_off = (kgrp * T_CI + 32 * g + l31) * ENTRY;
    int j = -3;
    for (int t = 0; t < T_pad; ++t) {
        if (j >= 0 && t < T) {
            u32x4 av[NPC];
            const unsigned char *pa = s_dy + (j & 1) * DY_BUF + a_off;
#pragma unroll
            for (int p = 0; p < NPC; ++p) av[p] = *reinterpret_cast<const u32x4 *>(pa + p * 2 * T_CO * ENTRY);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int ky = t9 / 3, kx = t9 - 3 * ky;
                // input row y + ky - 1 = relative row j + ky - 1 -> slot (j + ky - 1 + 1) & 3
                const unsigned char *pb = s_x + ((j + ky) & 3) * ROW_SLOT + kx * ROW_COPY + b_off;
                u32x4 bv[NPC];
#pragma unroll
                for (int p = 0; p < NPC; ++p) bv[p] = *reinterpret_cast<const u32x4 *>(pb + p * 2 * T_CI * ENTRY);
                mma_pieces<F16>(acc[t9], av, bv);
            }
        }
        barrier_lds();
        if (++j == nrows) j = -3;
    }
    // partial sums of this task: [tap][co 64][ci 64]; acc register i of lane (l31, kgrp): row 8 (i / 4) + 4 kgrp + i % 4, column l31
    float *out = a.partial + (((long)slice * a.ntiles + tile) * 9) * (T_CO * T_CI);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = 32 * h + 8 * (i >> 2) + 4 * kgrp + (i & 3), ci = 32 * g + l31;
            out[((long)t * T_CO + co) * T_CI + ci] = F16 ? (acc[t][i] * sx) * sdy : acc[t][i];
        }
}

// eight floats -> their 16-byte pieces at dst, dst + ps, ...
template <bool F16>
__device__ __forceinline__ void emit8(const float (&f)[8], unsigned char *dst, int ps, float inv) {
    constexpr int NPC = F16 ? 2 : 3;
    u32x4 pc[NPC];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned q[NPC];
        split_pair<F16>(f[2 * e], f[2 * e + 1], inv, q);
#pragma unroll
        for (int p = 0; p < NPC; ++p) pc[p][e] = q[p];
    }
#pragma unroll
    for (int p = 0; p < NPC; ++p) *reinterpret_cast<u32x4 *>(dst + p * ps) = pc[p];
}

constexpr int S2_SLOTS = 6;
constexpr int lds_bytes_s2(int npc) { return S2_SLOTS * row_slot(npc) + 2 * dy_buf(npc); }   // 120 / 80 KB

// Stride 2.  a.H, a.W = the INPUT map (both even), the dY planes are (H/2) x (W/2); rows_per_block / nseg count OUTPUT rows / 16-pixel
// segments of an output row.  Relative input row rr = input row - (2 y_lo - 1) lives in ring slot rr % 6; iteration j (-2 .. nrows - 1)
// multiplies output row y_lo + j (j >= 0: slots 2j, 2j+1, 2j+2) while the loaders stage rr = 2j + 3, 2j + 4 and dY row y_lo + j + 1.
template <bool F16>
__global__ __launch_bounds__(512) void k_conv_wgrad_s2(WgradArgs a) {
    constexpr int NPC = F16 ? 2 : 3, ROW_COPY = row_copy(NPC), ROW_SLOT = row_slot(NPC), DY_BUF = dy_buf(NPC);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_x = smem, *s_dy = smem + S2_SLOTS * ROW_SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int t_co = tile / a.ntile_ci, t_ci = tile - t_co * a.ntile_ci;
    const int n = slice / a.nblocks, blk = slice - n * a.nblocks;
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int y_lo = blk * a.rows_per_block, y_hi = min(OH, y_lo + a.rows_per_block);
    const long plane = (long)a.H * a.W, oplane = (long)OH * OW;
    const float *xn = a.x + ((long)n * a.cin + (long)t_ci * T_CI) * plane;
    const float *dyn = a.dy + ((long)n * a.cout + (long)t_co * T_CO) * oplane;
    const int nrows = y_hi - y_lo;
    const int r_base = 2 * y_lo - 1;   // input row of rr = 0
    const int T = a.nseg * (nrows + 2), T_pad = (T + PF - 1) / PF * PF;
    float sx, sdy;
    task_scales<F16>(a, lane, sx, sdy);
    const float inv_x = 1.f / sx, inv_dy = 1.f / sdy;
    if (wid >= 4) {
        // ------------------------------------------------------------------------------------------------ loaders
        // entry (ch, kh) = 8 output pixels x0 .. x0 + 7 of channel ch = input columns c0 + (kx - 1) + 2 e, c0 = 2 x0 (a multiple of 16).
        // Waves 0, 1 (kh = 0, 1): the ODD columns of both new rows - copies kx = 0 (from c0 - 1: one extra float) and kx = 2;
        // waves 2, 3: the EVEN columns of both rows (copy kx = 1) and the dY row.
        const int lt = tid - 256, lw = wid - 4;
        const int ch = lt & 63, kh = lw & 1;
        const bool odd_role = lw < 2;
        float4 vr[PF][2][4];      // the two input rows: columns c0 .. c0 + 15
        float halo[PF][2];        // column c0 - 1 (odd role)
        float4 vd[PF][2];         // dY: pixels x0 .. x0 + 7 (even role)
        bool okr[PF][2], okc[PF][4], okdq[PF][2], okh[PF];
        int i_seg = 0, i_j = -2, s_j = -2, s_base = 2;   // the step the next issue / staging works on; s_base = (2 s_j) mod 6
        auto issue = [&](float4 (&vr)[2][4], float (&halo)[2], float4 (&vd)[2], bool (&okr)[2], bool (&okc)[4], bool (&okdq)[2], bool &okh) __attribute__((always_inline)) {
            const int seg = min(i_seg, a.nseg - 1), j = i_j;
            const int x0 = seg * PX + 8 * kh, c0 = 2 * x0;
#pragma unroll
            for (int q = 0; q < 4; ++q) okc[q] = c0 + 4 * q < a.W;
            okh = c0 >= 1 && c0 - 1 < a.W;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = r_base + 2 * j + 3 + i;
                okr[i] = row >= 0 && row < a.H;
                const float *p = xn + (long)ch * plane + (long)min(max(row, 0), a.H - 1) * a.W;
#pragma unroll
                for (int q = 0; q < 4; ++q) vr[i][q] = *reinterpret_cast<const float4 *>(p + min(c0 + 4 * q, a.W - 4));
                if (odd_role) halo[i] = p[min(max(c0 - 1, 0), a.W - 1)];
            }
            if (!odd_role) {
                const int row = y_lo + j + 1;
                const bool okd = row >= y_lo && row < y_hi;
                const float *p = dyn + (long)ch * oplane + (long)min(max(row, 0), OH - 1) * OW;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    okdq[q] = okd && x0 + 4 * q < OW;
                    vd[q] = *reinterpret_cast<const float4 *>(p + min(x0 + 4 * q, OW - 4));
                }
            }
            if (++i_j == nrows) { i_j = -2; ++i_seg; }
        };
        auto stage = [&](const float4 (&vr)[2][4], const float (&halo)[2], const float4 (&vd)[2], const bool (&okr)[2], const bool (&okc)[4], const bool (&okdq)[2],
                         const bool &okh) __attribute__((always_inline)) {
            const int j = s_j, base = s_base;
            const int ps = 2 * T_CI * ENTRY;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int slot = base + 3 + i;
                slot = slot >= S2_SLOTS ? slot - S2_SLOTS : slot;
                unsigned char *xd = s_x + slot * ROW_SLOT + (kh * T_CI + ch) * ENTRY;
                float fl[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ok = okr[i] && okc[q];
                    fl[4 * q] = ok ? vr[i][q].x : 0.f; fl[4 * q + 1] = ok ? vr[i][q].y : 0.f; fl[4 * q + 2] = ok ? vr[i][q].z : 0.f; fl[4 * q + 3] = ok ? vr[i][q].w : 0.f;
                }
                float f[8];
                if (odd_role) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fl[2 * e + 1];
                    emit8<F16>(f, xd + 2 * ROW_COPY, ps, inv_x);                // kx = 2: columns c0 + 1 + 2 e
#pragma unroll
                    for (int e = 7; e > 0; --e) f[e] = f[e - 1];
                    f[0] = okr[i] && okh && okc[0] ? halo[i] : 0.f;            // kx = 0: columns c0 - 1 + 2 e
                    emit8<F16>(f, xd, ps, inv_x);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fl[2 * e];
                    emit8<F16>(f, xd + ROW_COPY, ps, inv_x);                    // kx = 1: columns c0 + 2 e
                }
            }
            if (!odd_role) {
                float f[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f[4 * q] = okdq[q] ? vd[q].x : 0.f; f[4 * q + 1] = okdq[q] ? vd[q].y : 0.f; f[4 * q + 2] = okdq[q] ? vd[q].z : 0.f; f[4 * q + 3] = okdq[q] ? vd[q].w : 0.f;
                }
                emit8<F16>(f, s_dy + ((j + 1) & 1) * DY_BUF + (kh * T_CO + ch) * ENTRY, 2 * T_CO * ENTRY, inv_dy);
            }
            s_base = s_base == 4 ? 0 : s_base + 2;
            if (++s_j == nrows) { s_j = -2; s_base = 2; }
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) issue(vr[d], halo[d], vd[d], okr[d], okc[d], okdq[d], okh[d]);
        for (int t = 0; t < T_pad; t += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                stage(vr[d], halo[d], vd[d], okr[d], okc[d], okdq[d], okh[d]);
                issue(vr[d], halo[d], vd[d], okr[d], okc[d], okdq[d], okh[d]);
                barrier_lds();
            }
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------------ compute waves
    const int h = wid & 1, g = wid >> 1;
    const int l31 = lane & 31, kgrp = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a_off = (kgrp * T_CO + 32 * h + l31) * ENTRY, b_off = (kgrp * T_CI + 32 * g + l31) * ENTRY;
    int j = -2, base = 2;
    for (int t = 0; t < T_pad; ++t) {
        if (j >= 0 && t < T) {
            u32x4 av[NPC];
            const unsigned char *pa = s_dy + (j & 1) * DY_BUF + a_off;
#pragma unroll
            for (int p = 0; p < NPC; ++p) av[p] = *reinterpret_cast<const u32x4 *>(pa + p * 2 * T_CO * ENTRY);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int ky = t9 / 3, kx = t9 - 3 * ky;
                int slot = base + ky;
                slot = slot >= S2_SLOTS ? slot - S2_SLOTS : slot;
                const unsigned char *pb = s_x + slot * ROW_SLOT + kx * ROW_COPY + b_off;
                u32x4 bv[NPC];
#pragma unroll
                for (int p = 0; p < NPC; ++p) bv[p] = *reinterpret_cast<const u32x4 *>(pb + p * 2 * T_CI * ENTRY);
                mma_pieces<F16>(acc[t9], av, bv);
            }
        }
        barrier_lds();
        base = base == 4 ? 0 : base + 2;
        if (++j == nrows) { j = -2; base = 2; }
    }
    float *out = a.partial + (((long)slice * a.ntiles + tile) * 9) * (T_CO * T_CI);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = 32 * h + 8 * (i >> 2) + 4 * kgrp + (i & 3), ci = 32 * g + l31;
            out[((long)t * T_CO + co) * T_CI + ci] = F16 ? (acc[t][i] * sx) * sdy : acc[t][i];
        }
}

// 7x7, stride 2, padding 3 (the ResNet-18 stem of uniplanner's lidar_conv_emb on the 384-channel BEV crops: MIOpen's igemm_wrw spent
// 3.7 ms on each of its two calls per train_full step).  49 taps x 16 accumulator registers do not fit a wave: a task takes ONE ky (seven
// taps, 112 registers) of a 64 x 64 tile.  A step = 16 output pixels of one output row = ONE input row (2 oy + ky - 3), staged as seven
// aligned copies (copy kx = columns 2 ox + kx - 3), double buffered: LDS 2 x 42 + 12 KB.  The four copies of the even kx are
// shifts of the row's odd columns and the three of the odd kx shifts of its even columns, so a loader thread splits every value into
// its three bf16 pieces ONCE per pairing (10 or 9 pair conversions for 4 or 3 copies) and only re-packs.
constexpr int k7_row(int npc) { return 7 * row_copy(npc); }                          // 42 / 28 KB
constexpr int lds_bytes_k7(int npc) { return 2 * k7_row(npc) + 2 * dy_buf(npc); }    // 96 / 64 KB

template <bool F16>
__global__ __launch_bounds__(512) void k_conv_wgrad_k7(WgradArgs a) {
    constexpr int NPC = F16 ? 2 : 3, ROW_COPY = row_copy(NPC), DY_BUF = dy_buf(NPC), K7_ROW = k7_row(NPC);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_x = smem, *s_dy = smem + 2 * K7_ROW;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x / 7, ky = blockIdx.x - 7 * tile, slice = blockIdx.y;
    const int t_co = tile / a.ntile_ci, t_ci = tile - t_co * a.ntile_ci;
    const int n = slice / a.nblocks, blk = slice - n * a.nblocks;
    const int OH = a.H >> 1, OW = a.W >> 1;
    const int y_lo = blk * a.rows_per_block, y_hi = min(OH, y_lo + a.rows_per_block);
    const long plane = (long)a.H * a.W, oplane = (long)OH * OW;
    const float *xn = a.x + ((long)n * a.cin + (long)t_ci * T_CI) * plane;
    const float *dyn = a.dy + ((long)n * a.cout + (long)t_co * T_CO) * oplane;
    const int nrows = y_hi - y_lo;
    // iteration j = -1 .. nrows - 1: the compute waves multiply output row y_lo + j out of buffer j & 1 while the loaders stage row
    // y_lo + j + 1 (its input row and its dY row) into the other one
    const int T = a.nseg * (nrows + 1), T_pad = (T + PF - 1) / PF * PF;
    float sx, sdy;
    task_scales<F16>(a, lane, sx, sdy);
    const float inv_x = 1.f / sx, inv_dy = 1.f / sdy;
    if (wid >= 4) {
        // ------------------------------------------------------------------------------------------------ loaders
        // entry (ch, kh) = 8 output pixels from x0 = 16 seg + 8 kh; fl[i] = input column c0 - 4 + i, c0 = 2 x0; copy kx holds
        // fl[2 e + kx + 1], e = 0 .. 7.  Waves 0, 1: kx = 0, 2, 4, 6 (the odd fl); waves 2, 3: kx = 1, 3, 5 (the even fl) and dY.
        const int lt = tid - 256, lw = wid - 4;
        const int ch = lt & 63, kh = lw & 1;
        const bool odd_role = lw < 2;
        float4 vr[PF][6], vd[PF][2];
        bool okq[PF][6], okdq[PF][2];
        int i_seg = 0, i_j = -1, s_j = -1;
        auto issue = [&](float4 (&vr)[6], float4 (&vd)[2], bool (&okq)[6], bool (&okdq)[2]) __attribute__((always_inline)) {
            const int seg = min(i_seg, a.nseg - 1), j = i_j;
            const int x0 = seg * PX + 8 * kh, c0 = 2 * x0;
            const int row = 2 * (y_lo + j + 1) + ky - 3;
            const bool rok = row >= 0 && row < a.H;
            const float *p = xn + (long)ch * plane + (long)min(max(row, 0), a.H - 1) * a.W;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int c = c0 - 4 + 4 * q;
                okq[q] = rok && c >= 0 && c < a.W;
                vr[q] = *reinterpret_cast<const float4 *>(p + min(max(c, 0), a.W - 4));
            }
            if (!odd_role) {
                const int orow = y_lo + j + 1;
                const bool okd = orow < y_hi;
                const float *pd = dyn + (long)ch * oplane + (long)min(orow, OH - 1) * OW;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    okdq[q] = okd && x0 + 4 * q < OW;
                    vd[q] = *reinterpret_cast<const float4 *>(pd + min(x0 + 4 * q, OW - 4));
                }
            }
            if (++i_j == nrows) { i_j = -1; ++i_seg; }
        };
        auto stage = [&](const float4 (&vr)[6], const float4 (&vd)[2], const bool (&okq)[6], const bool (&okdq)[2]) __attribute__((always_inline)) {
            const int j = s_j;
            const int ps = 2 * T_CI * ENTRY;
            unsigned char *xd = s_x + ((j + 1) & 1) * K7_ROW + (kh * T_CI + ch) * ENTRY;
            float fl[24];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                fl[4 * q] = okq[q] ? vr[q].x : 0.f; fl[4 * q + 1] = okq[q] ? vr[q].y : 0.f; fl[4 * q + 2] = okq[q] ? vr[q].z : 0.f; fl[4 * q + 3] = okq[q] ? vr[q].w : 0.f;
            }
            // v[m]: the role's columns in order (odd role: fl[1 + 2 m], m = 0 .. 10; even role: fl[2 + 2 m], m = 0 .. 9); pair conversions
            // pe[i] = (v[2i], v[2i+1]) and po[i] = (v[2i+1], v[2i+2]); copy number t of the role = v[e + t], i.e. pe[t/2 ..] or po[(t-1)/2 ..]
            auto copies = [&](auto ODD_) __attribute__((always_inline)) {
                constexpr bool ODD = decltype(ODD_)::value;
                constexpr int b = ODD ? 1 : 2, NT = ODD ? 4 : 3;
                unsigned pe[5][NPC], po[5][NPC];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    split_pair<F16>(fl[b + 4 * i], fl[b + 4 * i + 2], inv_x, pe[i]);
                    if (ODD || i < 4) split_pair<F16>(fl[b + 4 * i + 2], fl[b + 4 * i + 4], inv_x, po[i]);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    unsigned char *dst = xd + (ODD ? 2 * t : 2 * t + 1) * ROW_COPY;
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) {
                        u32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (t & 1) ? po[e + (t >> 1)][pc] : pe[e + (t >> 1)][pc];
                        *reinterpret_cast<u32x4 *>(dst + pc * ps) = v;
                    }
                }
            };
            if (odd_role) copies(std::true_type{});
            else copies(std::false_type{});
            if (!odd_role) {
                float f[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f[4 * q] = okdq[q] ? vd[q].x : 0.f; f[4 * q + 1] = okdq[q] ? vd[q].y : 0.f; f[4 * q + 2] = okdq[q] ? vd[q].z : 0.f; f[4 * q + 3] = okdq[q] ? vd[q].w : 0.f;
                }
                emit8<F16>(f, s_dy + ((j + 1) & 1) * DY_BUF + (kh * T_CO + ch) * ENTRY, 2 * T_CO * ENTRY, inv_dy);
            }
            if (++s_j == nrows) s_j = -1;
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) issue(vr[d], vd[d], okq[d], okdq[d]);
        for (int t = 0; t < T_pad; t += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                stage(vr[d], vd[d], okq[d], okdq[d]);
                issue(vr[d], vd[d], okq[d], okdq[d]);
                barrier_lds();
            }
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------------ compute waves
    const int h = wid & 1, g = wid >> 1;
    const int l31 = lane & 31, kgrp = lane >> 5;
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a_off = (kgrp * T_CO + 32 * h + l31) * ENTRY, b_off = (kgrp * T_CI + 32 * g + l31) * ENTRY;
    int j = -1;
    for (int t = 0; t < T_pad; ++t) {
        if (j >= 0 && t < T) {
            u32x4 av[NPC];
            const unsigned char *pa = s_dy + (j & 1) * DY_BUF + a_off;
#pragma unroll
            for (int p = 0; p < NPC; ++p) av[p] = *reinterpret_cast<const u32x4 *>(pa + p * 2 * T_CO * ENTRY);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const unsigned char *pb = s_x + (j & 1) * K7_ROW + kx * ROW_COPY + b_off;
                u32x4 bv[NPC];
#pragma unroll
                for (int p = 0; p < NPC; ++p) bv[p] = *reinterpret_cast<const u32x4 *>(pb + p * 2 * T_CI * ENTRY);
                mma_pieces<F16>(acc[kx], av, bv);
            }
        }
        barrier_lds();
        if (++j == nrows) j = -1;
    }
    float *out = a.partial + (((long)slice * a.ntiles + tile) * 49 + ky * 7) * (T_CO * T_CI);
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = 32 * h + 8 * (i >> 2) + 4 * kgrp + (i & 3), ci = 32 * g + l31;
            out[((long)t * T_CO + co) * T_CI + ci] = F16 ? (acc[t][i] * sx) * sdy : acc[t][i];
        }
}

// dW[co][ci][tap] = sum over the slices, in slice order, of partial[slice][tile][tap][co % 64][ci % 64]
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int nslices, int ntiles, int ntile_ci, int cin, int cout,
                                                      int ntaps, float *__restrict__ dw) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // (tile, tap, co_l, ci_l) in the partial's own order
    const long per_tile = (long)ntaps * T_CO * T_CI;
    if (e >= ntiles * per_tile) return;
    const int tile = (int)(e / per_tile);
    const int r = (int)(e - tile * per_tile), t = r / (T_CO * T_CI), co_l = (r / T_CI) % T_CO, ci_l = r % T_CI;
    float s = 0.f;
    for (int sl = 0; sl < nslices; ++sl) s += partial[(long)sl * ntiles * per_tile + e];
    const int co = (tile / ntile_ci) * T_CO + co_l, ci = (tile % ntile_ci) * T_CI + ci_l;
    dw[((long)co * cin + ci) * ntaps + t] = s;
}

// (H, W = the OUTPUT map of the layer = the dY planes; prologue = steps that only stage: 3 for stride 1, 2 for stride 2)
int wgrad_blocks(int B, int cin, int cout, int H, int W, int prologue, int tasks_per_tile = 1) {
    // Blocks of rows per image, by a cost in steps: a task walks nseg x (rows + 3) steps (three prologue steps per segment fill the row
    // ring), the chip runs one task per CU at a time.  tools/wgrad_probe.py: 64 -> 64 @160 x 32 images 738 us at 20 blocks of 8 rows
    // (640 tasks: three rounds, 27 % prologue) against 460 us at 8 blocks of 20 rows (256 tasks: one round).
    static const int cus = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    const long tiles = (long)(cin / T_CI) * (cout / T_CO) * tasks_per_tile;
    const int nseg = (W + PX - 1) / PX;
    int best = 1;
    double best_cost = 1e30;
    for (int nb = 1; nb <= H; ++nb) {
        const int rows = (H + nb - 1) / nb;
        if (rows < 4 && nb > 1) break;
        const long tasks = tiles * B * nb;
        if (tasks > 65535l * tiles) break;
        const double cost = (double)((tasks + cus - 1) / cus) * nseg * (rows + prologue) + 0.5 * nb;   // (+ the reduce launch reads nb partials)
        if (cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}
}  // namespace

namespace {
// row blocks per image of a (kernel size, stride) case; prologue steps and tasks per tile as the three kernels have them
int wgrad_blocks_of(int batch, int cin, int cout, int h, int w, int ksize, int stride) {
    return ksize == 7 ? wgrad_blocks(batch, cin, cout, h / 2, w / 2, 1, 7) : wgrad_blocks(batch, cin, cout, h / stride, w / stride, stride == 1 ? 3 : 2);
}
}  // namespace

extern "C" size_t lav_conv_wgrad_workspace_bytes(int batch, int cin, int cout, int h, int w, int ksize, int stride) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || cin % T_CI || cout % T_CO) return 0;
    if (!((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 7 && stride == 2))) return 0;
    if (w % 4 || (stride == 2 && (h % 2 || w % 8))) return 0;
    const int nb = wgrad_blocks_of(batch, cin, cout, h, w, ksize, stride);
    return (size_t)batch * nb * (cin / T_CI) * (cout / T_CO) * ksize * ksize * T_CO * T_CI * sizeof(float);
}

extern "C" int lav_conv_wgrad(const float *x, const float *dy, int batch, int cin, int cout, int h, int w, int ksize, int stride, float *dw,
                              void *workspace, size_t workspace_bytes, void *stream) {
    return lav_conv_wgrad_amax(x, dy, batch, cin, cout, h, w, ksize, stride, dw, workspace, workspace_bytes, nullptr, 0, nullptr, 0, stream);
}

extern "C" int lav_conv_wgrad_amax(const float *x, const float *dy, int batch, int cin, int cout, int h, int w, int ksize, int stride, float *dw,
                                   void *workspace, size_t workspace_bytes, const float *amax_x, int n_amax_x, const float *amax_dy, int n_amax_dy,
                                   void *stream) {
    LAV_REQUIRE(x && dy && dw, "lav_conv_wgrad: null argument");
    LAV_REQUIRE((amax_x == nullptr) == (amax_dy == nullptr) && (!amax_x || (n_amax_x >= 1 && n_amax_dy >= 1)), "lav_conv_wgrad_amax: the maxima of x and dY come together");
    const bool f16 = amax_x != nullptr;
    LAV_REQUIRE((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 7 && stride == 2), "lav_conv_wgrad: %dx%d kernel of stride %d (3x3 of stride 1 / 2, 7x7 of stride 2)", ksize, ksize, stride);
    LAV_REQUIRE(batch >= 1 && h >= 1 && w >= 4 && w % 4 == 0, "lav_conv_wgrad: batch %d, map %dx%d (rows of whole 16-byte pieces)", batch, h, w);
    LAV_REQUIRE(stride == 1 || (h % 2 == 0 && w % 8 == 0), "lav_conv_wgrad: stride 2 takes even heights and widths that are multiples of 8 (%dx%d)", h, w);
    LAV_REQUIRE(cin >= T_CI && cout >= T_CO && cin % T_CI == 0 && cout % T_CO == 0, "lav_conv_wgrad: %d -> %d channels (multiples of 64)", cin, cout);
    LAV_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0, "lav_conv_wgrad: x and dy must be 16-byte aligned");
    const size_t need = lav_conv_wgrad_workspace_bytes(batch, cin, cout, h, w, ksize, stride);
    if (!workspace || workspace_bytes < need) return fail(LAV_EWORKSPACE, "lav_conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr = false;
    if (!attr) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(3)));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad_s2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_s2(3)));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad_k7<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_k7(3)));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2)));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad_s2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_s2(2)));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad_k7<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_k7(2)));
        attr = true;
    }
    const int oh = h / stride, ow = w / stride;
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = static_cast<float *>(workspace);
    a.B = batch; a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.nseg = (ow + PX - 1) / PX;
    a.nblocks = wgrad_blocks_of(batch, cin, cout, h, w, ksize, stride);
    a.rows_per_block = (oh + a.nblocks - 1) / a.nblocks;
    a.ntile_ci = cin / T_CI; a.ntiles = a.ntile_ci * (cout / T_CO);
    a.amax_x = amax_x; a.amax_dy = amax_dy; a.n_amax_x = n_amax_x; a.n_amax_dy = n_amax_dy;
    const int nslices = batch * a.nblocks;
    LAV_REQUIRE(nslices <= 65535, "lav_conv_wgrad: %d slices", nslices);
    const int tok = timer_begin("conv_wgrad", st);
    if (f16) {
        if (ksize == 7) hipLaunchKernelGGL(k_conv_wgrad_k7<true>, dim3(a.ntiles * 7, nslices), dim3(512), lds_bytes_k7(2), st, a);
        else if (stride == 1) hipLaunchKernelGGL(k_conv_wgrad<true>, dim3(a.ntiles, nslices), dim3(512), lds_bytes(2), st, a);
        else hipLaunchKernelGGL(k_conv_wgrad_s2<true>, dim3(a.ntiles, nslices), dim3(512), lds_bytes_s2(2), st, a);
    } else {
        if (ksize == 7) hipLaunchKernelGGL(k_conv_wgrad_k7<false>, dim3(a.ntiles * 7, nslices), dim3(512), lds_bytes_k7(3), st, a);
        else if (stride == 1) hipLaunchKernelGGL(k_conv_wgrad<false>, dim3(a.ntiles, nslices), dim3(512), lds_bytes(3), st, a);
        else hipLaunchKernelGGL(k_conv_wgrad_s2<false>, dim3(a.ntiles, nslices), dim3(512), lds_bytes_s2(3), st, a);
    }
    const int ntaps = ksize * ksize;
    const long total = (long)a.ntiles * ntaps * T_CO * T_CI;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.partial, nslices, a.ntiles, a.ntile_ci, cin, cout, ntaps, dw);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

// Weight gradient of the training graph's 3x3 stride-1 convolutions on the bf16 matrix cores (round 5).
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
constexpr int ENTRY = 16;                               // bytes of one lane operand (8 bf16)
constexpr int ROW_COPY = 3 * 2 * T_CI * ENTRY;          // one kx copy of an input row: [piece 3][k half 2][ci 64] entries = 6 KB
constexpr int ROW_SLOT = 3 * ROW_COPY;                  // three kx copies = 18 KB
constexpr int DY_BUF = 3 * 2 * T_CO * ENTRY;            // [piece 3][k half 2][co 64] = 6 KB
constexpr int LDS_BYTES = 4 * ROW_SLOT + 2 * DY_BUF;    // 84 KB

struct WgradArgs {
    const float *x, *dy;
    float *partial;     // [slice][tile][tap 9][co 64][ci 64]
    int B, cin, cout, H, W;
    int nseg;           // 16-pixel column segments of a row
    int rows_per_block, nblocks;   // a slice = (image, block of rows)
    int ntile_ci, ntiles;
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

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) void k_conv_wgrad(WgradArgs a) {
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
    if (wid >= 4) {
        // ------------------------------------------------------------------------------------------------ loaders
        const int lt = tid - 256;   // 0 .. 255; two items per thread and step.  Loader wave lw = 0, 1: copy kx = 0 of the input row, then
        //                             copy 2; lw = 2, 3: copy 1, then the dY row.  (ch, kh) = channel and k half of the 8-pixel entry.
        const int lw = wid - 4;                          // scalar
        const int ch = lt & 63, kh = lw & 1;
        const int kx0 = lw >> 1;                         // first item: copy kx0 of the input row
        const bool second_is_x = lw < 2;                 // second item: copy 2 of the input row, or dY
        float4 va[3], vb[3];                             // raw loads of the two items (three aligned 16-byte pieces each)
        bool oka[3], okb[3];
        // window of an X item: pixels w0 .. w0 + 7 with w0 = x0 + 8 kh + kx - 1; aligned base a0 = w0 rounded down to 4
        auto issue = [&](int seg, int j) {
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
        };
        // eight consecutive floats starting SH floats into the 12 loaded ones -> three 16-byte bf16 pieces at `dst` (piece stride ps)
        auto emit = [&](auto SH_, const float4 (&v)[3], const bool (&ok)[3], unsigned char *dst, int ps) __attribute__((always_inline)) {
            constexpr int SH = decltype(SH_)::value;
            float f[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                f[4 * q] = ok[q] ? v[q].x : 0.f; f[4 * q + 1] = ok[q] ? v[q].y : 0.f; f[4 * q + 2] = ok[q] ? v[q].z : 0.f; f[4 * q + 3] = ok[q] ? v[q].w : 0.f;
            }
            u32x4 p0, p1, p2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned q0, q1, q2;
                split3x2(f[SH + 2 * e], f[SH + 2 * e + 1], q0, q1, q2);
                p0[e] = q0; p1[e] = q1; p2[e] = q2;
            }
            *reinterpret_cast<u32x4 *>(dst) = p0;
            *reinterpret_cast<u32x4 *>(dst + ps) = p1;
            *reinterpret_cast<u32x4 *>(dst + 2 * ps) = p2;
        };
        using std::integral_constant;
        auto stage = [&](int j) {   // what issue(seg, j) loaded
            const int slot = (j + 2 + 1) & 3;   // input row y_lo + j + 2 -> ring slot (relative row + 1) mod 4
            unsigned char *xd = s_x + slot * ROW_SLOT + (kh * T_CI + ch) * ENTRY;
            // the first item's window starts 3 floats into its aligned base for kx 0 (a0 = x0 - 4), 0 floats for kx 1
            if (kx0 == 0) emit(integral_constant<int, 3>{}, va, oka, xd, 2 * T_CI * ENTRY);
            else emit(integral_constant<int, 0>{}, va, oka, xd + ROW_COPY, 2 * T_CI * ENTRY);
            if (second_is_x) emit(integral_constant<int, 1>{}, vb, okb, xd + 2 * ROW_COPY, 2 * T_CI * ENTRY);
            else emit(integral_constant<int, 0>{}, vb, okb, s_dy + ((j + 1) & 1) * DY_BUF + (kh * T_CO + ch) * ENTRY, 2 * T_CO * ENTRY);
        };
        for (int seg = 0; seg < a.nseg; ++seg) {
            issue(seg, -3);
            for (int j = -3; j < nrows; ++j) {
                // loads of "j" are in registers (issued one step ago): stage them, then issue the next step's
                stage(j);
                if (j + 1 < nrows) issue(seg, j + 1);
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
    for (int seg = 0; seg < a.nseg; ++seg) {
        for (int j = -3; j < nrows; ++j) {
            if (j >= 0) {
                u32x4 av[3];
                const unsigned char *pa = s_dy + (j & 1) * DY_BUF + a_off;
#pragma unroll
                for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const u32x4 *>(pa + p * 2 * T_CO * ENTRY);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - 3 * ky;
                    // input row y + ky - 1 = relative row j + ky - 1 -> slot (j + ky - 1 + 1) & 3
                    const unsigned char *pb = s_x + ((j + ky) & 3) * ROW_SLOT + kx * ROW_COPY + b_off;
                    u32x4 bv[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bv[p] = *reinterpret_cast<const u32x4 *>(pb + p * 2 * T_CI * ENTRY);
                    constexpr int PA[6] = {0, 0, 1, 0, 1, 2}, PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[PA[k]]), __builtin_bit_cast(bf16x8, bv[PB[k]]), acc[t], 0, 0, 0);
                }
            }
            barrier_lds();
        }
    }
    // partial sums of this task: [tap][co 64][ci 64]; acc register i of lane (l31, kgrp): row 8 (i / 4) + 4 kgrp + i % 4, column l31
    float *out = a.partial + (((long)slice * a.ntiles + tile) * 9) * (T_CO * T_CI);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = 32 * h + 8 * (i >> 2) + 4 * kgrp + (i & 3), ci = 32 * g + l31;
            out[((long)t * T_CO + co) * T_CI + ci] = acc[t][i];
        }
}

// dW[co][ci][tap] = sum over the slices, in slice order, of partial[slice][tile][tap][co % 64][ci % 64]
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int nslices, int ntiles, int ntile_ci, int cin, int cout,
                                                      float *__restrict__ dw) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // (tile, tap, co_l, ci_l) in the partial's own order
    const long per_tile = 9l * T_CO * T_CI;
    if (e >= ntiles * per_tile) return;
    const int tile = (int)(e / per_tile);
    const int r = (int)(e - tile * per_tile), t = r / (T_CO * T_CI), co_l = (r / T_CI) % T_CO, ci_l = r % T_CI;
    float s = 0.f;
    for (int sl = 0; sl < nslices; ++sl) s += partial[(long)sl * ntiles * per_tile + e];
    const int co = (tile / ntile_ci) * T_CO + co_l, ci = (tile % ntile_ci) * T_CI + ci_l;
    dw[((long)co * cin + ci) * 9 + t] = s;
}

int wgrad_blocks(int B, int cin, int cout, int H, int W) {
    // Blocks of rows per image, by a cost in steps: a task walks nseg x (rows + 3) steps (three prologue steps per segment fill the row
    // ring), the chip runs one task per CU at a time.  tools/wgrad_probe.py: 64 -> 64 @160 x 32 images 738 us at 20 blocks of 8 rows
    // (640 tasks: three rounds, 27 % prologue) against 460 us at 8 blocks of 20 rows (256 tasks: one round).
    static const int cus = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    const long tiles = (long)(cin / T_CI) * (cout / T_CO);
    const int nseg = (W + PX - 1) / PX;
    int best = 1;
    double best_cost = 1e30;
    for (int nb = 1; nb <= H; ++nb) {
        const int rows = (H + nb - 1) / nb;
        if (rows < 4 && nb > 1) break;
        const long tasks = tiles * B * nb;
        if (tasks > 65535l * tiles) break;
        const double cost = (double)((tasks + cus - 1) / cus) * nseg * (rows + 3) + 0.5 * nb;   // (+ the reduce launch reads nb partials)
        if (cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}
}  // namespace

extern "C" size_t lav_conv_wgrad_workspace_bytes(int batch, int cin, int cout, int h, int w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || cin % T_CI || cout % T_CO) return 0;
    const int nb = wgrad_blocks(batch, cin, cout, h, w);
    return (size_t)batch * nb * (cin / T_CI) * (cout / T_CO) * 9 * T_CO * T_CI * sizeof(float);
}

extern "C" int lav_conv_wgrad(const float *x, const float *dy, int batch, int cin, int cout, int h, int w, float *dw, void *workspace,
                              size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(x && dy && dw, "lav_conv_wgrad: null argument");
    LAV_REQUIRE(batch >= 1 && h >= 1 && w >= 4 && w % 4 == 0, "lav_conv_wgrad: batch %d, map %dx%d (rows of whole 16-byte pieces)", batch, h, w);
    LAV_REQUIRE(cin >= T_CI && cout >= T_CO && cin % T_CI == 0 && cout % T_CO == 0, "lav_conv_wgrad: %d -> %d channels (multiples of 64)", cin, cout);
    LAV_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0, "lav_conv_wgrad: x and dy must be 16-byte aligned");
    const size_t need = lav_conv_wgrad_workspace_bytes(batch, cin, cout, h, w);
    if (!workspace || workspace_bytes < need) return fail(LAV_EWORKSPACE, "lav_conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr = false;
    if (!attr) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr = true;
    }
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = static_cast<float *>(workspace);
    a.B = batch; a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.nseg = (w + PX - 1) / PX;
    a.nblocks = wgrad_blocks(batch, cin, cout, h, w);
    a.rows_per_block = (h + a.nblocks - 1) / a.nblocks;
    a.ntile_ci = cin / T_CI; a.ntiles = a.ntile_ci * (cout / T_CO);
    const int nslices = batch * a.nblocks;
    LAV_REQUIRE(nslices <= 65535, "lav_conv_wgrad: %d slices", nslices);
    const int tok = timer_begin("conv_wgrad", st);
    hipLaunchKernelGGL(k_conv_wgrad, dim3(a.ntiles, nslices), dim3(512), LDS_BYTES, st, a);
    const long total = (long)a.ntiles * 9 * T_CO * T_CI;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.partial, nslices, a.ntiles, a.ntile_ci, cin, cout, dw);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

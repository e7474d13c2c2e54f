// Implicit-GEMM 2-D convolution on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s chip peak).
//
// Replaces the cuDNN convolutions behind the reference's ConvBackbone / Head
// (team_code_v2/models/lidar.py:48-161) and ResNet-18 embedder (lav/models/resnet.py).  One kernel covers
// Conv2d (any kernel / stride / padding / dilation) and ConvTranspose2d: a transposed convolution of stride s
// is decomposed into s*s output-parity classes, each an ordinary small-tap convolution over the input grid
// (class (ry,rx): oy + pad = s*qy + ry, taps ky = ry + s*j reading input row qy - j), all classes in one launch.
//
// GEMM view per (image, class):  D[cout][q] = sum_k W[cout][k] * X[k][q],   q = linearised output-grid pixel,
// k = (tap, cin).  MFMA operands: A = weights (lane l: cout l&31, k parity l>>5), B = activations (lane l:
// pixel l&31, k parity l>>5); the accumulator then has pixel = lane&31, i.e. the epilogue's NCHW stores are
// 128-byte contiguous per half-wave.  Activations stay NCHW end to end (the canvas, the 384-channel feature map
// handed to crop_feature and the head outputs are NCHW in the reference API).
//
// Workgroup = 4 waves; tile = (128*MP pixels) x (32*MC couts); per cin chunk of CK channels the input rows the
// tile touches (full width, zero-padded halo) and the weight slab [taps][CK][32*MC] are staged in LDS; every
// (tap, channel-pair) step is MP*MC MFMAs fed by MP+MC conflict-free ds_read_b32.
// Epilogue (fused): +bias -> ReLU -> *scale+shift (eval BatchNorm; the reference puts BN AFTER the ReLU,
// lidar.py:58-60, so it cannot be folded into the weights) -> +residual -> ReLU -> sigmoid, then a channel-
// offset store (fused torch.cat of lidar.py:143).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "conv_f16.hpp"

namespace {
using namespace lav;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_TAPS = 64;
constexpr int MAX_CLASSES = 16;
constexpr int CK = 16;        // input channels per LDS stage (compile time: the pair loop is fully unrolled)
constexpr int TAP_GROUP = 9;  // taps per weight stage (a 7x7 kernel is staged one kernel row at a time)
constexpr int NPOS_MAX = 40;  // bound on staged input positions per thread (LDS capacity is the real limit)

struct ConvArgs {
    const float *x, *w, *bias, *scale, *shift, *res;
    const int *n_valid;   // lav_batch_limit: images >= *n_valid are skipped (null: all)
    unsigned long long *trace;  // debug (LAV_CONV_TRACE): [workgroup][8] wall-clock stamps
    float *y;
    int in_c_total, in_c_offset, cin, H, W;
    int cout, out_c_total, out_c_offset, OH, OW;
    int cin_pad, cout_pad;  // packed-weight strides (multiples of CK / 64), zero filled
    int QH, QW, in_s, out_s;
    int Wst, ROWS, plane_pad, nclasses, taps_per_class, tap_group;
    int cps;      // 16-channel chunks staged per pipeline stage (short layers stage all of K at once)
    int ksplit;   // >1: the cin chunks are split over `ksplit` workgroups writing raw partial sums to `partial`
    float pad_value;   // what out-of-image input positions hold (zero, or a folded normalisation's pre-image of zero)
    float *partial;
    int in_bufs;  // 2: input tile double buffered; 1: tile too large for that (wide 7x7 stems) - loaded at chunk start
    int rowblock, xblocks;  // 1: tiles are PIXW-wide segments of ONE output-grid row (wide images); 0: linearised pixels
    int relu_pre, relu_post, sigmoid;
    int cls_ntaps[MAX_CLASSES], cls_in_oy[MAX_CLASSES], cls_in_ox[MAX_CLASSES];
    int cls_out_oy[MAX_CLASSES], cls_out_ox[MAX_CLASSES], cls_woff[MAX_CLASSES];
    int toff[MAX_TAPS];  // class c, tap t -> toff[c*taps_per_class + t] = dy*Wst + dx
};

// Operands of one tap: 8 channel pairs x (MC weight fragments + MP activation fragments)
template <int MP, int MC>
struct TapOps {
    float a[CK / 2][MC], b[CK / 2][MP];
};

// Weight slab of one tap in LDS (and in HBM, see lav_conv_pack_weights): [cout block of 32][8-channel group][lane][4],
// lane = (channel parity)*32 + cout, element q = channel pair q of the group - i.e. exactly the A operands of four
// consecutive k-steps per lane, fetched with one ds_read_b128.
template <int MP, int MC>
__device__ __forceinline__ void load_tap(TapOps<MP, MC> &o, const float *__restrict__ s_w_tap, const float *__restrict__ s_in,
                                         const int (&base)[MP], int to, int plane, int lane, int half) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int g = 0; g < CK / 8; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(s_w_tap + mc * (CK * 32) + g * 256 + lane * 4);
            o.a[4 * g + 0][mc] = v.x; o.a[4 * g + 1][mc] = v.y; o.a[4 * g + 2][mc] = v.z; o.a[4 * g + 3][mc] = v.w;
        }
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp) {
        const int c = 2 * cp + half;
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) o.b[cp][mp] = s_in[c * plane + base[mp] + to];
    }
}

template <int MP, int MC>
__device__ __forceinline__ void mma_tap(const TapOps<MP, MC> &o, f32x16 (&acc)[MC][MP]) {
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
            for (int mp = 0; mp < MP; ++mp)
                acc[mc][mp] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[cp][mc], o.b[cp][mp], acc[mc][mp], 0, 0, 0);
}

template <int MP, int MC, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void k_conv(ConvArgs a) {
#define CONV_STAMP(i) do { if constexpr (TRACE) { if (threadIdx.x == 0) a.trace[(((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } } while (0)
    CONV_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CO_T = 32 * MC, PIXW = 128 * MP;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: loops over this wave's slices run on the SALU
    const int ks = blockIdx.z % a.ksplit;
    const int cls = (blockIdx.z / a.ksplit) % a.nclasses, n = blockIdx.z / (a.ksplit * a.nclasses);
    if (a.n_valid && n >= *a.n_valid) return;   // workgroup-uniform
    const int cb = blockIdx.y * CO_T;
    const int Q = a.QH * a.QW;
    const int Wst = a.Wst, ROWS = a.ROWS, plane = a.plane_pad;  // channel stride in LDS (ROWS*Wst rounded up to 64)
    const int in_floats = a.cps * CK * plane, w_floats = a.cps * a.tap_group * CK * CO_T;
    float *s_in0 = smem;                               // [2][CK][plane]   double-buffered input tile
    float *s_w0 = smem + a.in_bufs * in_floats;        // [2][tap_group][CK][CO_T]   double-buffered weight slab
    // all scalar arguments the prologue needs, fetched in ONE batch of kernarg loads with a single wait (instead of a
    // dependent s_load + s_waitcnt round trip at each first use)
    asm volatile("" ::"s"(a.x), "s"(a.w), "s"(a.pad_value), "s"(a.in_c_total), "s"(a.in_c_offset), "s"(a.cin), "s"(a.H), "s"(a.W),
                 "s"(a.cin_pad), "s"(a.cout_pad), "s"(a.QH), "s"(a.QW), "s"(a.in_s), "s"(a.Wst), "s"(a.ROWS), "s"(a.plane_pad),
                 "s"(a.taps_per_class), "s"(a.tap_group), "s"(a.cps), "s"(a.in_bufs), "s"(a.rowblock), "s"(a.xblocks));
    const int ntaps = a.cls_ntaps[cls];
    // tile origin: first output-grid row, first staged input column (relative to in_ox)
    const int xb = a.rowblock ? (int)(blockIdx.x % a.xblocks) : 0;
    const int q0 = a.rowblock ? 0 : blockIdx.x * PIXW;
    const int qy0 = a.rowblock ? (int)(blockIdx.x / a.xblocks) : q0 / a.QW;
    const int xs0 = xb * PIXW * a.in_s;
    const int iy_base = qy0 * a.in_s + a.cls_in_oy[cls];
    const int in_ox = a.cls_in_ox[cls] + xs0;
    const float *wbase = a.w + a.cls_woff[cls];
    const int *toff = a.toff + cls * a.taps_per_class;

    // this lane's MP output-grid pixels
    int pqy[MP], pqx[MP];
    bool pvalid[MP];
    int base[MP];
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) {
        const int local = (wid * MP + mp) * 32 + l31;
        if (a.rowblock) {
            pqy[mp] = qy0;
            pqx[mp] = xb * PIXW + local;
            pvalid[mp] = pqx[mp] < a.QW;
            pqx[mp] = min(pqx[mp], a.QW - 1);
        } else {
            const int q = q0 + local;
            pvalid[mp] = q < Q;
            const int qc = min(q, Q - 1);
            pqy[mp] = qc / a.QW;
            pqx[mp] = qc - pqy[mp] * a.QW;
        }
        base[mp] = (pqy[mp] - qy0) * a.in_s * Wst + pqx[mp] * a.in_s - xs0;
    }
    // input staging map: thread owns tile positions tid + 256*i, walked incrementally (no per-position division)
    const int rr0 = tid / Wst, xx0 = tid - rr0 * Wst;
    const int step_q = 256 / Wst, step_r = 256 - step_q * Wst;
    const int live = ROWS * Wst;

    f32x16 acc[MC][MP];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int mp = 0; mp < MP; ++mp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mc][mp][r] = 0.f;

    const float *xin = a.x + ((long)n * a.in_c_total + a.in_c_offset) * a.H * a.W;
    const long cplane = (long)a.H * a.W;
    const int ngroups = (ntaps + a.tap_group - 1) / a.tap_group;
    const int nchunks_all = (a.cin + CK - 1) / CK;
    const int chunk_lo = ks * nchunks_all / a.ksplit, chunk_hi = (ks + 1) * nchunks_all / a.ksplit;
    const int nsuper = (chunk_hi - chunk_lo + a.cps - 1) / a.cps;  // stages stage `cps` chunks at a time
    const int nstages = nsuper * ngroups;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;

    // Asynchronous global -> LDS DMA (global_load_lds): data never passes through VGPRs, so the loads of stage
    // s+1 are in flight while stage s runs on the matrix pipes.  The LDS destination of one wave-instruction is
    // wave-uniform base + lane*size, which is exactly how both tiles are laid out (positions / float4s in thread
    // order).
    // Out-of-image positions (the same ones in every chunk and stage - they only depend on the tile) and the
    // planes of a ragged channel tail are written ONCE, with the pad value; the DMA then runs under an exec mask with the
    // channel plane as a scalar base and the pixel as a 32-bit lane offset: 4 mostly scalar instructions per load
    // (s_mov m0 / s_add / load / 64-bit add).
    {
        const int nplanes = a.in_bufs * a.cps * CK;
        const int tail = a.cin % CK;   // last chunk's live channels when the channel count is ragged (0: none)
        int rr = rr0, xx = xx0;
        for (int pb = 64 * wid; pb < plane; pb += 256) {
            const int iy = iy_base + rr, ix = in_ox + xx;
            const bool valid = pb + lane < live && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            if (!valid) {
                for (int s = 0; s < nplanes; ++s) s_in0[s * plane + pb + lane] = a.pad_value;
            } else if (tail) {   // planes >= tail of every chunk slot may be the last chunk's dead channels
                for (int s = 0; s < nplanes; ++s)
                    if ((s % CK) >= tail) s_in0[s * plane + pb + lane] = 0.f;
            }
            rr += step_q;
            xx += step_r;
            if (xx >= Wst) { xx -= Wst; ++rr; }
        }
        if (tail) __syncthreads();   // the zeroed tail planes of a slot are DMA targets while it holds a full chunk
    }
    auto issue_input = [&](int sc) {  // super-chunk sc: chunks chunk_lo + sc*cps ... (up to cps of them)
        float *dst0 = s_in0 + (a.in_bufs == 2 ? (sc & 1) * in_floats : 0);
        const int c_first = chunk_lo + sc * a.cps, c_last = min(c_first + a.cps, chunk_hi);
        for (int chunk = c_first; chunk < c_last; ++chunk) {
            const int ci0 = chunk * CK;
            float *dst = dst0 + (chunk - c_first) * CK * plane;
            int rr = rr0, xx = xx0;
            for (int pb = 64 * wid; pb < plane; pb += 256) {  // wave-uniform: this wave's 64 positions pb..pb+63
                const int iy = iy_base + rr, ix = in_ox + xx;
                const int g = (pb + lane < live && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? iy * a.W + ix : -1;
                if (g >= 0) {   // exec mask: lanes on padding keep the prefilled value
                    const char *pix = reinterpret_cast<const char *>(xin + (long)ci0 * cplane) + (size_t)(unsigned)(g * 4);
                    float *d = dst + pb;
                    const size_t cstride = (size_t)cplane * 4;
                    const int nc = min(CK, a.cin - ci0);   // wave-uniform
                    if (nc == CK) {
#pragma unroll
                        for (int c = 0; c < CK; ++c) {
                            __builtin_amdgcn_global_load_lds((gptr_t)pix, (lptr_t)(d + c * plane), 4, 0, 0);
                            pix += cstride;
                            asm volatile("" : "+v"(pix));   // keep the pointer a running sum (one 64-bit add per channel)
                        }
                    } else {
                        for (int c = 0; c < nc; ++c) {
                            __builtin_amdgcn_global_load_lds((gptr_t)pix, (lptr_t)(d + c * plane), 4, 0, 0);
                            pix += cstride;
                        }
                    }
                }
                rr += step_q;
                xx += step_r;
                if (xx >= Wst) { xx -= Wst; ++rr; }
            }
        }
    };
    auto issue_weights = [&](int stage) {
        const int sc = stage / ngroups, grp = stage % ngroups;
        const int c_first = chunk_lo + sc * a.cps, c_last = min(c_first + a.cps, chunk_hi);
        const int t0 = grp * a.tap_group;
        const int nt = min(a.tap_group, ntaps - t0);
        // one tap of one 16-channel chunk and cout block is 512 contiguous floats in HBM (2 groups x 64 lanes x 4)
        const long blk_stride = (long)ntaps * a.cin_pad * 32;
        for (int chunk = c_first; chunk < c_last; ++chunk) {
            const int ci0 = chunk * CK;
            float *wdst = s_w0 + (stage & 1) * w_floats + (chunk - c_first) * a.tap_group * CK * CO_T;
            for (int f0 = 64 * wid; f0 < nt * MC * 128; f0 += 256) {  // float4 index; wave-uniform bounds
                const int piece = f0 >> 7;   // (tap, cout block): 128 float4s each, a wave's 64 never straddle two
                const int tap = piece / MC, mc = piece - tap * MC;
                const float *src = wbase + (long)(blockIdx.y * MC + mc) * blk_stride + ((long)(t0 + tap) * a.cin_pad + ci0) * 32 +
                                   4 * ((f0 & 127) + lane);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wdst + 4 * f0), 16, 0, 0);
            }
        }
    };

    if (nstages > 0) {
        issue_weights(0);
        if (a.in_bufs == 2) issue_input(0);
    }
    CONV_STAMP(1);
    for (int stage = 0; stage < nstages; ++stage) {
        const int sc = stage / ngroups, grp = stage % ngroups;
        if (a.in_bufs == 1 && grp == 0) {  // single input buffer: everyone must be done with the previous chunks first
            __syncthreads();
            issue_input(sc);
        }
        // stage's DMA has landed for every wave, and every wave is done computing stage-1 (whose buffers stage+1 reuses)
        __syncthreads();  // hipcc drains vmcnt(0) ahead of the barrier because LDS-DMA is in flight
        if (stage == 0) CONV_STAMP(2);
        if (stage + 1 < nstages) {
            issue_weights(stage + 1);
            if (a.in_bufs == 2 && grp == ngroups - 1) issue_input(sc + 1);
        }
        const int t0 = grp * a.tap_group;
        const int nt = min(a.tap_group, ntaps - t0);
        const int nsub = min(a.cps, chunk_hi - (chunk_lo + sc * a.cps));
        for (int sub = 0; sub < nsub; ++sub) {
            const float *s_in = s_in0 + (a.in_bufs == 2 ? (sc & 1) * in_floats : 0) + sub * CK * plane;
            const float *s_w = s_w0 + (stage & 1) * w_floats + sub * a.tap_group * CK * CO_T;
            // software pipeline over taps: operands of tap t+1 are fetched from LDS while tap t runs on the MFMA pipe
            TapOps<MP, MC> o0, o1;
            load_tap<MP, MC>(o0, s_w, s_in, base, toff[t0], plane, lane, half);
            int t = 0;
            for (; t + 1 < nt; t += 2) {
                load_tap<MP, MC>(o1, s_w + (t + 1) * CK * CO_T, s_in, base, toff[t0 + t + 1], plane, lane, half);
                mma_tap<MP, MC>(o0, acc);
                if (t + 2 < nt) load_tap<MP, MC>(o0, s_w + (t + 2) * CK * CO_T, s_in, base, toff[t0 + t + 2], plane, lane, half);
                mma_tap<MP, MC>(o1, acc);
            }
            if (t < nt) mma_tap<MP, MC>(o0, acc);
        }
    }

    CONV_STAMP(3);
    const int out_oy = a.cls_out_oy[cls], out_ox = a.cls_out_ox[cls];
    if (a.ksplit > 1) {  // raw partial sums [ks][n][cout][OH][OW]; k_conv_reduce adds them up and applies the epilogue
        const long plane_o = (long)a.OH * a.OW;
        const int batch = gridDim.z / (a.ksplit * a.nclasses);
        float *pbase = a.partial + ((long)ks * batch + n) * a.cout * plane_o;
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                const int oy = pqy[mp] * a.out_s + out_oy, ox = pqx[mp] * a.out_s + out_ox;
                const bool pix_ok = pvalid[mp] && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb + mc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pix_ok && co < a.cout) pbase[co * plane_o + (long)oy * a.OW + ox] = acc[mc][mp][r];
                }
            }
        CONV_STAMP(4);
        return;
    }
    const bool has_bias = a.bias != nullptr, has_aff = a.scale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
        // per-channel epilogue vectors of this lane's 16 output channels, loaded once
        float bv[16], sv[16], tv[16];
        int cov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb + mc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            cov[r] = co;
            const int cc = min(co, a.cout - 1);
            bv[r] = has_bias ? a.bias[cc] : 0.f;
            sv[r] = has_aff ? a.scale[cc] : 1.f;
            tv[r] = has_aff ? a.shift[cc] : 0.f;
        }
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) {
            const int oy = pqy[mp] * a.out_s + out_oy, ox = pqx[mp] * a.out_s + out_ox;
            const bool pix_ok = pvalid[mp] && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
            const long pix = (long)oy * a.OW + ox;
            const long cbase = ((long)n * a.out_c_total + a.out_c_offset) * a.OH * a.OW;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long idx = cbase + (long)min(cov[r], a.cout - 1) * a.OH * a.OW + (pix_ok ? pix : 0);
                rv[r] = has_res ? a.res[idx] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mc][mp][r] + bv[r];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                v = fmaf(v, sv[r], tv[r]);
                v += rv[r];
                if (a.relu_post) v = v > 0.f ? v : 0.f;
                if (a.sigmoid && cov[r] >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
                if (pix_ok && cov[r] < a.cout) a.y[cbase + (long)cov[r] * a.OH * a.OW + pix] = v;
            }
        }
    }
    CONV_STAMP(4);
#undef CONV_STAMP
}

// Split-K second pass: y = epilogue( sum_ks partial[ks] ), fixed summation order (deterministic).  amax_out (round 6, LAV_CONV_F16X3's
// scale hand-off): block b leaves the largest finite |y| it wrote in amax_out[b] (0 for blocks beyond lav_batch_limit).
__device__ __forceinline__ float reduce_finite_abs(float v) {
    const float a = fabsf(v);
    return a <= 3.4028235e38f ? a : 0.f;
}
__global__ __launch_bounds__(256) void k_conv_reduce(ConvArgs a, int batch, float *__restrict__ amax_out) {
    const long plane_o = (long)a.OH * a.OW;
    const long total = (long)batch * a.cout * plane_o;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(min(e, total - 1) / (a.cout * plane_o));
    // lav_batch_limit: rows the first pass skipped hold no partial sums
    const bool live = e < total && !(a.n_valid && n >= *a.n_valid);
    float m = 0.f;
    if (live) {
        const int co = (int)((e / plane_o) % a.cout);
        const long pix = e % plane_o;
        float v = 0.f;
        for (int ks = 0; ks < a.ksplit; ++ks) v += a.partial[(long)ks * total + e];
        if (a.bias) v += a.bias[co];
        if (a.relu_pre) v = v > 0.f ? v : 0.f;
        if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
        const long idx = ((long)n * a.out_c_total + a.out_c_offset + co) * plane_o + pix;
        if (a.res) v += a.res[idx];
        if (a.relu_post) v = v > 0.f ? v : 0.f;
        if (a.sigmoid && co >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
        a.y[idx] = v;
        m = reduce_finite_abs(v);
    }
    if (amax_out) {   // (kernel-uniform)
        __shared__ float s_m[4];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) amax_out[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    }
}

// ------------------------------------------------------------------------------------------------ host plan
struct Tap {
    int dy, dx, ky, kx;
};
struct Plan {
    int OH, OW, QH, QW, in_s, out_s, max_dy, max_dx, nclasses, taps_per_class;
    std::vector<std::vector<Tap>> taps;  // per class
    std::vector<int> in_oy, in_ox, out_oy, out_ox;
    std::vector<size_t> woff;
    size_t wfloats;
    int cin_pad, cout_pad;
};

int build_plan(const lav_conv &c, Plan &p) {
    LAV_REQUIRE(c.batch >= 1 && c.cin >= 1 && c.cout >= 1 && c.h >= 1 && c.w >= 1, "lav_conv: bad sizes");
    LAV_REQUIRE(c.kh >= 1 && c.kw >= 1 && c.stride >= 1 && c.dil_h >= 1 && c.dil_w >= 1, "lav_conv: bad kernel");
    LAV_REQUIRE(c.in_c_offset >= 0 && c.in_c_offset + c.cin <= c.in_c_total, "lav_conv: input channel window");
    LAV_REQUIRE(c.out_c_offset >= 0 && c.out_c_offset + c.cout <= c.out_c_total, "lav_conv: output channel window");
    p.taps.clear(); p.in_oy.clear(); p.in_ox.clear(); p.out_oy.clear(); p.out_ox.clear(); p.woff.clear();
    if (!c.transposed) {
        p.OH = (c.h + 2 * c.pad_h - c.dil_h * (c.kh - 1) - 1) / c.stride + 1;
        p.OW = (c.w + 2 * c.pad_w - c.dil_w * (c.kw - 1) - 1) / c.stride + 1;
        LAV_REQUIRE(p.OH >= 1 && p.OW >= 1, "lav_conv: empty output");
        p.QH = p.OH; p.QW = p.OW; p.in_s = c.stride; p.out_s = 1;
        std::vector<Tap> t;
        for (int ky = 0; ky < c.kh; ++ky)
            for (int kx = 0; kx < c.kw; ++kx) t.push_back({ky * c.dil_h, kx * c.dil_w, ky, kx});
        p.taps.push_back(t);
        p.in_oy.push_back(-c.pad_h); p.in_ox.push_back(-c.pad_w); p.out_oy.push_back(0); p.out_ox.push_back(0);
    } else {
        LAV_REQUIRE(c.dil_h == 1 && c.dil_w == 1, "lav_conv: dilated ConvTranspose2d unsupported");
        const int s = c.stride;
        p.OH = (c.h - 1) * s - 2 * c.pad_h + c.kh + c.out_pad;
        p.OW = (c.w - 1) * s - 2 * c.pad_w + c.kw + c.out_pad;
        LAV_REQUIRE(p.OH >= 1 && p.OW >= 1, "lav_conv: empty output");
        p.QH = (p.OH - 1 + c.pad_h) / s + 1; p.QW = (p.OW - 1 + c.pad_w) / s + 1;
        p.in_s = 1; p.out_s = s;
        for (int ry = 0; ry < s; ++ry)
            for (int rx = 0; rx < s; ++rx) {
                const int nty = ry < c.kh ? (c.kh - ry + s - 1) / s : 0;
                const int ntx = rx < c.kw ? (c.kw - rx + s - 1) / s : 0;
                std::vector<Tap> t;
                for (int dy = 0; dy < nty; ++dy)
                    for (int dx = 0; dx < ntx; ++dx) t.push_back({dy, dx, ry + s * (nty - 1 - dy), rx + s * (ntx - 1 - dx)});
                p.taps.push_back(t);
                p.in_oy.push_back(-(nty > 0 ? nty - 1 : 0)); p.in_ox.push_back(-(ntx > 0 ? ntx - 1 : 0));
                p.out_oy.push_back(ry - c.pad_h); p.out_ox.push_back(rx - c.pad_w);
            }
    }
    p.nclasses = (int)p.taps.size();
    LAV_REQUIRE(p.nclasses <= MAX_CLASSES, "lav_conv: stride %d gives %d classes > %d", c.stride, p.nclasses, MAX_CLASSES);
    p.taps_per_class = 0; p.max_dy = 0; p.max_dx = 0;
    p.cin_pad = (c.cin + CK - 1) / CK * CK;
    p.cout_pad = (c.cout + 63) / 64 * 64;
    size_t off = 0;
    for (auto &t : p.taps) {
        p.taps_per_class = std::max<int>(p.taps_per_class, (int)t.size());
        for (auto &tp : t) { p.max_dy = std::max(p.max_dy, tp.dy); p.max_dx = std::max(p.max_dx, tp.dx); }
        p.woff.push_back(off);
        off += t.size() * (size_t)p.cin_pad * p.cout_pad;
    }
    p.wfloats = off;
    LAV_REQUIRE(p.taps_per_class * p.nclasses <= MAX_TAPS, "lav_conv: %d taps x %d classes exceed %d", p.taps_per_class, p.nclasses, MAX_TAPS);
    return LAV_OK;
}

#include "conv_split.hpp"
#include "conv_smallcin.hpp"

// Tile shape + staging geometry.  Cost model (units: MFMA time of one k-step): a CU runs ceil(nwg/256) workgroups
// back to back on its matrix pipes, each costing MP*MC MFMAs per k-step plus ~0.5 of LDS staging / operand fetch.
// Wide images switch to row-blocked tiles (a tile = PIXW pixels of ONE output-grid row) when the full-width rows of
// a linearised tile do not fit the staging map / LDS.
int choose_tile(const lav_conv &c, const Plan &p, ConvArgs &a, int &MP, int &MC, size_t &lds, double *cost = nullptr, double *raw_cost = nullptr) {
    // *cost: the plan's estimate in us; made infinite-cheap (0) for deep 2x2-tile plans, which the direct kernel never beats
    a.cin_pad = p.cin_pad; a.cout_pad = p.cout_pad;
    const long Q = (long)p.QH * p.QW;
    struct Geo { int rowblock, xblocks, Wst, ROWS, plane_pad, tap_group, in_bufs; size_t lds; long nwg; bool ok; };
    auto geo = [&](int mp, int mc, int rowblock, int in_bufs, int tg_opt) {
        Geo g;
        const int PIXW = 128 * mp, CO_T = 32 * mc;
        g.rowblock = rowblock; g.in_bufs = in_bufs; g.xblocks = 1;
        const int span_rows = (int)std::min<long>((PIXW - 1 + p.QW - 1) / p.QW + 1, p.QH);
        g.Wst = (p.QW - 1) * p.in_s + p.max_dx + 1;
        g.ROWS = (span_rows - 1) * p.in_s + p.max_dy + 1;
        if (rowblock) {
            g.xblocks = (p.QW + PIXW - 1) / PIXW;
            g.Wst = (std::min(PIXW, p.QW) - 1) * p.in_s + p.max_dx + 1;
            g.ROWS = p.max_dy + 1;
        }
        const size_t LDS_MAX = 160 * 1024;
        const long pad = ((long)g.ROWS * g.Wst + 63) / 64 * 64;
        auto bytes_with = [&](int tg) { return (size_t)(g.in_bufs * (size_t)CK * pad + 2 * (size_t)tg * CK * CO_T) * 4; };
        // weight slab: all taps of a class, one kernel row, or a single tap (tg_opt 0 / 1 / 2)
        static const int force_tg = [] { const char *e = getenv("LAV_CONV_FORCE_TG"); return e ? atoi(e) : 0; }();   // experiments
        const int opts[3] = {p.taps_per_class, p.nclasses == 1 ? c.kw : 1, 1};
        g.tap_group = force_tg ? force_tg : opts[tg_opt];
        if (g.tap_group < 1 || g.tap_group > TAP_GROUP || bytes_with(g.tap_group) > LDS_MAX) g.tap_group = 0;
        g.plane_pad = (int)pad;
        g.lds = bytes_with(std::max(g.tap_group, 1));
        g.ok = g.tap_group >= 1 && pad <= 256 * NPOS_MAX && g.lds <= LDS_MAX;
        const long xt = g.rowblock ? (long)p.QH * g.xblocks : (Q + PIXW - 1) / PIXW;
        g.nwg = xt * ((c.cout + CO_T - 1) / CO_T) * c.batch * p.nclasses;
        return g;
    };
    // Cost of a candidate in microseconds, calibrated on MI355X traces (LAV_CONV_TRACE): one (16-channel chunk x tap) of
    // a tile costs 0.45 + 0.3*MP*MC us on the matrix pipes (a single-buffered input cannot overlap its DMA: x1.2; a
    // one-tap weight slab pays a barrier per tap: x1.3); a workgroup costs ~5 us of set-up + epilogue; a split-K reduce
    // launch ~6 us.  Matrix work is conserved per CU (rounds = workgroups / 256), fixed costs overlap between the
    // workgroups that share a CU's LDS; staging a chunk costs ~1 ns per staged position (instruction issue).  Wasted pixels (a 160-pixel row cut into 128 + 32, a 48-pixel row in a 128-pixel
    // row block) show up as extra workgroups.  Split-K (ks workgroups per tile, each a slice of the channel loop) is
    // part of the search whenever a layer has fewer tiles than CUs, so a layer never lands a few workgroups above a
    // multiple of 256.
    const int nchunks = (c.cin + CK - 1) / CK;
    // CUs the plan aims to fill: a layer that runs beside another stream's kernels is better off NOT claiming every CU
    static const long ncu_env = [] { const char *e = getenv("LAV_CONV_CUS"); const int v = e ? atoi(e) : 0; return (long)(v >= 16 && v <= 256 ? v : 256); }();
    const long ncu = c.target_cus >= 16 && c.target_cus <= 256 ? c.target_cus : ncu_env;
    double best = 1e30;
    Geo bg{};
    bool found = false;
    a.ksplit = 1;
    const int cand[3][2] = {{1, 1}, {1, 2}, {2, 2}};
    const int full_group = std::min(p.taps_per_class, p.nclasses == 1 ? c.kw : 1);
    for (auto &cd : cand) {
        if (cd[1] == 2 && c.cout <= 32) continue;
        for (int mode = 0; mode < 4; ++mode) {   // preference on ties: linearised before row-blocked, double before single buffer
          for (int tg_opt = 0; tg_opt < 3; ++tg_opt) {
            if (tg_opt > 0 && (tg_opt == 1 ? (p.nclasses == 1 ? c.kw : 1) == p.taps_per_class : (p.nclasses == 1 ? c.kw : 1) == 1)) continue;  // duplicate slab size
            const Geo g = geo(cd[0], cd[1], mode >> 1, (mode & 1) ? 1 : 2, tg_opt);
            if (!g.ok) continue;
            double unit = (0.45 + 0.3 * cd[0] * cd[1]) * p.taps_per_class;
            const long per_cu = std::max<long>(1, std::min<long>(2, (long)(160 * 1024 / g.lds)));
            if (g.in_bufs == 1) unit *= 1.2;
            if (g.tap_group < full_group) unit *= 1.3;
            unit += 0.001 * g.plane_pad;   // issuing the chunk's DMA: ~9 instructions per 64 staged positions and channel, same waves
            // two workgroups per CU hide each other's barrier / DMA stalls: measured on the 384->256 head convolution
            // (1200 workgroups): 76 KB single-buffered tiles 556 us vs 128 KB double-buffered ones 643 us.  It only pays
            // when a layer is several rounds deep (BEV-size layers measured no gain).
            const bool deep = g.nwg * std::max(1, std::min(16, nchunks / 2)) >= 3 * ncu && g.nwg >= ncu;
            if (per_cu >= 2 && deep) unit *= g.in_bufs == 1 ? 0.85 / 1.2 : 0.85;
            static const int force_ks = [] { const char *e = getenv("LAV_CONV_FORCE_KS"); return e ? atoi(e) : 0; }();   // experiments
            static const int force_mode = [] { const char *e = getenv("LAV_CONV_FORCE_MODE"); return e ? atoi(e) : -1; }();
            static const int force_tile = [] { const char *e = getenv("LAV_CONV_FORCE_TILE"); return e ? atoi(e) : 0; }();
            if (force_mode >= 0 && mode != force_mode) continue;
            if (force_tile && cd[0] * 10 + cd[1] != force_tile) continue;
            const int ks_max = force_ks ? force_ks : (nchunks >= 4 ? std::min(16, nchunks / 2) : 1);
            // partial sums: ks slabs of the output written by the tiles and read back by the reduce launch (~4 TB/s)
            const double slab_us = (double)c.batch * c.cout * p.OH * p.OW * 4.0 * 2.0 / 4e6;
            // what a split costs besides its partial-sum traffic: the reduce launch.  6 us was calibrated on stand-alone layers; LAV_CONV_KS_COST
            // re-prices it (round 6: inside the frame the extra launch and its workgroups also cost the other streams - tools/frame_ab.py)
            static const double ks_cost = [] { const char *e = getenv("LAV_CONV_KS_COST"); return e ? atof(e) : 6.0; }();
            for (int ks = force_ks ? force_ks : 1; ks <= ks_max; ++ks) {
                const long wgs = g.nwg * ks;
                const double t = (double)((wgs + ncu - 1) / ncu) * ((nchunks + ks - 1) / ks) * unit +
                                 (double)((wgs + ncu * per_cu - 1) / (ncu * per_cu)) * 5.0 + (ks > 1 ? ks_cost + ks * slab_us : 0.0);
                if (t < best * (ks > 1 ? 0.97 : 1.0) - 1e-9) {   // a larger split must pay for its partial-sum traffic
                    best = t; MP = cd[0]; MC = cd[1]; bg = g; found = true; a.ksplit = ks;
                }
            }
          }
        }
    }
    if (cost) *cost = (found && MP == 2 && MC == 2 && bg.nwg * a.ksplit >= 512) ? 0.0 : best;   // 384->256 head conv: tiled 486 us (93 TF/s), direct 565
    if (raw_cost) *raw_cost = best;   // the estimate itself (what the split kernel's plan is compared with)
    LAV_REQUIRE(found, "lav_conv2d: no tile shape fits (grid %dx%d, stride %d, %d taps)", p.QH, p.QW, p.in_s, p.taps_per_class);
    a.rowblock = bg.rowblock; a.xblocks = bg.xblocks; a.Wst = bg.Wst; a.ROWS = bg.ROWS;
    a.plane_pad = bg.plane_pad; a.tap_group = bg.tap_group; a.in_bufs = bg.in_bufs;
    lds = bg.lds;
    // chunks per stage: a stage of one 16-channel chunk is only taps*8*MP*MC MFMAs (a few hundred ns) while the DMA
    // of the next stage needs a memory round trip, so short layers stage several chunks (up to all of K) at once.
    // Layers that fill the chip more than once keep their LDS footprint (two workgroups per CU) instead.
    {
        const int per_split = (nchunks + a.ksplit - 1) / a.ksplit;
        const int CO_T = 32 * MC;
        // keep the footprint at <= 64 KB so that workgroups of OTHER streams (the frame graph runs the brake net, the ego
        // branch and the LiDAR chain concurrently) can share the CU: a 160 KB workgroup monopolises its CU's LDS
        // ... unless the whole layer is at most one workgroup per CU: then nothing of this layer queues behind the
        // footprint, and staging more of K up front saves a barrier + DMA round trip per stage (small gain, measured)
        static const size_t small_budget = [] { const char *e = getenv("LAV_CONV_SMALL_LDS_KB"); return (size_t)(e ? atoi(e) : 64) * 1024; }();
        const bool one_wave = bg.nwg * a.ksplit <= 256;
        const size_t budget = std::max<size_t>(lds, one_wave ? small_budget : 64 * 1024);
        int cps = 1;
        for (int cand : {2, 4, 8}) {
            if (cand > per_split * 2 - 1 && cand > per_split) break;
            const size_t b = ((size_t)a.in_bufs * cand * CK * a.plane_pad + 2 * (size_t)cand * a.tap_group * CK * CO_T) * 4;
            if (b <= budget) cps = cand;
        }
        cps = std::min(cps, per_split);
        a.cps = std::max(cps, 1);
        lds = ((size_t)a.in_bufs * a.cps * CK * a.plane_pad + 2 * (size_t)a.cps * a.tap_group * CK * CO_T) * 4;
    }
    return LAV_OK;
}

// ------------------------------------------------------------------------------------------------ direct path
// Small layers (the 24x24 ... 3x3 maps of the ResNet-18 embedders, a few hundred output pixels per image): the tiled
// kernel above costs ~17-30 us there whatever the work, because every workgroup stages a 128-pixel tile through LDS
// behind barriers, each IMAGE gets its own (mostly empty) tiles, and split-K needs a second launch.  Here the GEMM's
// M dimension is the pixels of the whole batch (n, oy, ox linearised), a workgroup owns 32 pixels x 32 couts, its
// waves split the input channels between them and feed the MFMAs straight from global memory / L2 (one dword per lane
// and operand per k-step: the packed weights [tap][cin][cout] are already cout-contiguous, the activation gather is
// pixel-contiguous), and the waves' partial tiles are summed through LDS in a fixed order.  No staging, one barrier.
struct DirectArgs {
    const float *x, *w, *bias, *scale, *shift, *res;
    const int *n_valid;   // lav_batch_limit: workgroups whose first pixel belongs to an image >= *n_valid exit (null: all)
    float *y, *partial;
    int in_c_total, in_c_offset, H, W;
    int cout, out_c_total, out_c_offset, OH, OW;
    int cin_pad, cout_pad, dil_h, dil_w;
    // output-grid geometry (Plan): a Conv2d is ONE class over its output pixels; a ConvTranspose2d of stride s is s*s parity
    // classes over the grid (QH, QW), class c reading input (q*in_s + in_o + tap*dil) and writing output (q*out_s + out_o)
    int QH, QW, in_s, out_s, nclasses;
    int cls_nty[MAX_CLASSES], cls_ntx[MAX_CLASSES], cls_in_oy[MAX_CLASSES], cls_in_ox[MAX_CLASSES];
    int cls_out_oy[MAX_CLASSES], cls_out_ox[MAX_CLASSES], cls_woff[MAX_CLASSES];
    int M;        // batch * QH * QW
    int cw;       // input channels per wave (even)
    int cks;      // input channels per split-K slice (= waves * cw)
    int ksplit;
    int relu_pre, relu_post, sigmoid;
    float pad_value;
    float *amax_out;   // whole-K launches: the workgroup's largest finite |y| -> amax_out[linear workgroup] (null: not wanted)
};

// DEPTH = load batches in flight per wave: every weight byte is used once, so the stream runs at (bytes in flight) /
// (HBM latency) - a wave keeps up to DEPTH * (PAIRS/4 KB of weights + PAIRS gathers) outstanding.
// MC = 32-cout blocks per workgroup: the gathered activations of a batch feed MC MFMAs each (the texture path issues
// ~16 cycles per wave load, a CU's four matrix pipes want an operand pair every 16: one cout block is gather bound).
template <int DEPTH, int MC>
__global__ __launch_bounds__(MC == 1 ? 1024 : 512) void k_conv_direct(DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_red[];   // [waves][MC * 32][33]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int ks = blockIdx.z % a.ksplit, cls = blockIdx.z / a.ksplit;
    const int plane_o = a.OH * a.OW, plane_q = a.QH * a.QW;
    const int nty = a.cls_nty[cls], ntx = a.cls_ntx[cls];
    const long wg_lin = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (a.n_valid && (int)(blockIdx.x * 32) / plane_q >= *a.n_valid) {   // workgroup-uniform
        if (a.amax_out && a.ksplit == 1 && tid == 0) a.amax_out[wg_lin] = 0.f;
        return;
    }
    const int m = blockIdx.x * 32 + l31;
    const bool mvalid = m < a.M;
    const int mcl = min(m, a.M - 1);
    const int n = mcl / plane_q, pix = mcl - n * plane_q, qy = pix / a.QW, qx = pix - qy * a.QW;
    const int cb = blockIdx.y * 32 * MC;
    const int cplane = a.H * a.W;
    const int c0 = ks * a.cks + wid * a.cw + half;   // this lane's channel of pair 0 (pair j: c0 + 2j)
    // Addressing is kept off the vector ALU: both operands are (scalar base) + (32-bit lane offset) loads.  Activations:
    // byte offset of (image, channel c0, pixel 0) per lane, + the tap's pixel, advanced by 8 channels per batch.
    // Packed weights [cout block][tap][8-channel group][lane][pair]: a wave's slice of a tap is contiguous, one dwordx4
    // per lane = the A operands of 4 k-steps; the base pointer is a scalar that walks through the slice.
    const unsigned xlane = (unsigned)((n * a.in_c_total + a.in_c_offset + c0) * cplane) * 4u;
    const unsigned pstride = (unsigned)cplane * 8u;                 // bytes between the channels of consecutive pairs
    const char *xbase = reinterpret_cast<const char *>(a.x);
    const int wslice = ks * a.cks + wid * a.cw;                     // first channel of this wave
    const long wtap = (long)a.cin_pad * 32;
    const long wblk = (long)(nty * ntx) * wtap;                     // floats per cout block (of this class)
    const float *wtap0 = a.w + a.cls_woff[cls] + (long)blockIdx.y * MC * wblk + (long)wslice * 32;   // scalar
    const unsigned wlane = lane * 16u;
    const int iy0 = qy * a.in_s + a.cls_in_oy[cls], ix0 = qx * a.in_s + a.cls_in_ox[cls];
    constexpr int PAIRS = 4;                                        // channel pairs (= MFMAs) per load batch
    const int spt = a.cw / (2 * PAIRS);                             // load batches per tap
    const int nsteps = nty * ntx * spt;

    // running state of the load stream: tap (ky, kx), batch j within the tap
    int ky = 0, kx = 0, j = 0;
    const char *wp = reinterpret_cast<const char *>(wtap0);         // scalar
    bool ok;
    unsigned xoff;
    auto enter_tap = [&]() {
        const int iy = iy0 + ky * a.dil_h, ix = ix0 + kx * a.dil_w;
        ok = mvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        xoff = xlane + (ok ? (unsigned)(iy * a.W + ix) * 4u : 0u);
    };
    enter_tap();
    struct Ops { float av[MC][PAIRS], bv[PAIRS]; bool ok; };
    auto load = [&](Ops &o) {
#pragma unroll
        for (int b = 0; b < MC; ++b) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + b * wblk * 4 + wlane);
            o.av[b][0] = v.x; o.av[b][1] = v.y; o.av[b][2] = v.z; o.av[b][3] = v.w;
        }
        // every lane loads (padding lanes read pixel 0 of their channel) and the pad value is selected when the batch is
        // consumed: loads under an exec-mask branch would hide from the compiler's vmcnt bookkeeping and force it to
        // drain the whole queue at each wait
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) o.bv[p] = *reinterpret_cast<const float *>(xbase + (xoff + p * pstride));
        o.ok = ok;
        // advance (scalar control): next batch of this tap, or the first of the next tap
        wp += 256 * 4;
        xoff += PAIRS * pstride;
        if (++j == spt) {
            j = 0;
            if (++kx == ntx) { kx = 0; ++ky; }
            wp += (wtap - (long)a.cw * 32) * 4;
            enter_tap();
        }
    };
    f32x16 acc[MC];
#pragma unroll
    for (int b = 0; b < MC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    auto mma = [&](const Ops &o) {
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            const float bvp = o.ok ? o.bv[p] : a.pad_value;
#pragma unroll
            for (int b = 0; b < MC; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[b][p], bvp, acc[b], 0, 0, 0);
        }
    };
    // nsteps is a multiple of DEPTH (host picks DEPTH that way), so every load / MFMA batch below is unconditional and
    // the compiler can wait for exactly the oldest batch (s_waitcnt vmcnt(n)) instead of draining the queue
    if (nsteps > 0) {   // (a parity class of a transposed convolution can be without taps: its pixels are bias only)
        Ops ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) load(ring[i]);
        for (int s = DEPTH; s < nsteps; s += DEPTH) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                mma(ring[i]);
                load(ring[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) mma(ring[i]);
    }

    float *mine = s_red + wid * (MC * 32 * 33);
#pragma unroll
    for (int b = 0; b < MC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(b * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = acc[b][r];
    __syncthreads();
    // (A ticket / last-arriver reduction inside this kernel was measured slower than the second launch: 15.1 vs 13.6 us on
    // the 512-channel 3x3 layer - the write-through stores, the atomic round trip and the acquire cost as much as a launch.)
    float wmax = 0.f;
    for (int e = tid; e < MC * 1024; e += blockDim.x) {
        const int i = e >> 5, col = e & 31;
        const int co = cb + i, mm = blockIdx.x * 32 + col;
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += s_red[(w * (MC * 32) + i) * 33 + col];
        if (co >= a.cout || mm >= a.M) continue;
        const int on = mm / plane_q, oq = mm - on * plane_q, oqy = oq / a.QW, oqx = oq - oqy * a.QW;
        const int oy = oqy * a.out_s + a.cls_out_oy[cls], ox = oqx * a.out_s + a.cls_out_ox[cls];
        if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
        const int opix = oy * a.OW + ox;
        if (a.ksplit > 1) {   // raw partial sums in k_conv_reduce's layout [ks][n][cout][OH][OW]
            a.partial[((long)ks * (a.M / plane_q) + on) * a.cout * plane_o + (long)co * plane_o + opix] = v;
            continue;
        }
        if (a.bias) v += a.bias[co];
        if (a.relu_pre) v = v > 0.f ? v : 0.f;
        if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
        const long idx = ((long)on * a.out_c_total + a.out_c_offset + co) * plane_o + opix;
        if (a.res) v += a.res[idx];
        if (a.relu_post) v = v > 0.f ? v : 0.f;
        if (a.sigmoid && co >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
        a.y[idx] = v;
        wmax = fmaxf(wmax, reduce_finite_abs(v));
    }
    if (a.amax_out && a.ksplit == 1) {   // (kernel-uniform) the partial tiles in s_red are consumed: its first floats carry the waves' maxima
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
        __syncthreads();
        if (lane == 0) s_red[wid] = wmax;
        __syncthreads();
        if (tid == 0) {
            float mm = 0.f;
            for (int w = 0; w < nw; ++w) mm = fmaxf(mm, s_red[w]);
            a.amax_out[wg_lin] = mm;
        }
    }
}

// Direct-path plan: waves per workgroup, split-K factor and a cost estimate (us) comparable with choose_tile's.
struct DirectPlan { bool ok; int waves, ksplit, cw; double cost; long tiles; int mc; };

DirectPlan choose_direct(const lav_conv &c, const Plan &p) {
    DirectPlan d{false, 0, 1, 0, 1e30, 0, 1};
    // experiments (read per call so that a probe can sweep them): LAV_CONV_DIRECT 0 never / 1 by cost / 2 whenever possible,
    // LAV_CONV_DIRECT_WAVES and LAV_CONV_DIRECT_KS pin the workgroup size and the split
    auto env_int = [](const char *k, int dflt) { const char *e = getenv(k); return e ? atoi(e) : dflt; };
    const int mode = env_int("LAV_CONV_DIRECT", 1), force_w = env_int("LAV_CONV_DIRECT_WAVES", 0), force_k = env_int("LAV_CONV_DIRECT_KS", 0);
    if (!mode || c.cin % 16 != 0 || (c.transposed && !env_int("LAV_CONV_DIRECT_TR", 1))) return d;
    const long M = (long)c.batch * p.QH * p.QW;   // per class
    // the kernel addresses activations with 32-bit BYTE offsets from the tensor base
    if (M * c.cout >= (1l << 31) || (long)c.batch * c.in_c_total * c.h * c.w >= (1l << 30)) return d;
    const int taps = p.taps_per_class;   // the largest class
    if (taps > 9 && mode != 2) return d;   // 7x7 stems re-read too much without an LDS tile (measured 93 vs 86 us, 475 vs 416)
    const double slab_us = (double)c.batch * c.cout * p.OH * p.OW * 4.0 * 2.0 / 4e6;
    const int force_mc = env_int("LAV_CONV_DIRECT_MC", 0);
    for (int mc = 1; mc <= 2; ++mc) {
        if ((mc == 2 && c.cout < 64) || (force_mc && mc != force_mc)) continue;
        // transposed: measured a gain only for 4-tap classes with >= 128 couts (4x4 s2 up-convolution 53.7 -> 49.8 us)
        if (mc == 2 && !force_mc && c.transposed && (p.taps_per_class < 4 || c.cout < 128)) continue;
        const long tiles = (M + 31) / 32 * ((c.cout + 32 * mc - 1) / (32 * mc)) * p.nclasses;
        if (tiles > 8192) continue;
        // sharing a gather between two cout blocks pays once the layer is a few workgroups per CU deep (measured: 8-12 % at
        // >= 400 tiles, a loss on the small ResNet maps)
        if (mc == 2 && !force_mc && (M + 31) / 32 * ((c.cout + 31) / 32) * p.nclasses < 512) continue;
        for (int ks = 1; ks <= 16; ks *= 2) {
            if (c.cin % (ks * 8) != 0) break;
            if (force_k && ks != force_k) continue;
            const int cks = c.cin / ks;
            if (ks > 1 && cks < 32 && !force_k) break;   // keep at least 4 waves x 8 channels per slice
            for (int waves : {1, 2, 4, 8, 16}) {
                if (cks % (8 * waves) != 0 || (force_w && waves != force_w) || (mc == 2 && waves > 8)) continue;
                if (waves < 4 && cks % 32 == 0 && !force_w) continue;   // 1-2 waves only for 16-channel inputs
                const int cw = cks / waves;
                // calibrated on MI355X (tools/direct_probe.py): ~6.5 us of launch + prologue + LDS reduction + epilogue, the
                // MFMAs of the waves that share a SIMD at ~2/3 of the pipe's rate (gather bound; two cout blocks per
                // gather: ~0.87), ~3.5 us for the split-K reduce launch
                const long wgs = tiles * ks;
                const double mpw = (double)taps * (cw / 2) * mc;   // MFMAs per wave
                const double waves_per_simd = std::max((double)waves / 4.0, (double)wgs * waves / 1024.0);
                // every tap re-reads its operands from L2 (no LDS tile): ~10.8 TB/s over the chip bounds large layers; a wave's
                // load batches are serialised a ring at a time (+0.05 us per batch favours more, shorter waves)
                const double l2_us = (double)wgs * taps * cks * 32 * 4.0 * (mc + 1) / 10.8e6;
                // (the reduce launch: 3.5 us is what it adds to a layer measured alone (tools/direct_probe.py); inside the frame it is one more
                //  launch in a chain of dependent ones and its workgroups compete with the other streams'.  Round 6 priced it by the frame
                //  (tools/frame_ab.py, profiles/r06_frame_experiments.txt section 8): 3.5 -> 2.031-2.040 ms, 7 / 8 -> 2.021-2.031, 9.5 -> 2.012-2.025,
                //  10.5 -> 2.017-2.031, 11.2 / 12 -> 2.026-2.038 (the ego branch's 256- and 512-channel layers lose their split: 531 -> 564 us),
                //  never -> 2.052-2.061.  LAV_CONV_DIRECT_KS_COST re-prices it.)
                static const double dks_cost = [] { const char *e = getenv("LAV_CONV_DIRECT_KS_COST"); return e ? atof(e) : 9.5; }();
                const double t = 6.5 + std::max((mc == 1 ? 1.5 : 1.15) * waves_per_simd * mpw * 0.0267, l2_us) + 0.05 * taps * (cw / 8) +
                                 0.04 * mc * waves * std::max(1.0, (double)wgs / 256.0) + (ks > 1 ? dks_cost + ks * slab_us : 0.0);   // + LDS reduction per workgroup round
                if (t < d.cost) d = DirectPlan{true, waves, ks, cw, t, tiles, mc};
            }
        }
    }
    if (mode == 2 && d.ok) d.cost = 0.0;
    return d;
}
// precision of a layer: LAV_CONV_F32 (exact fp32 MFMA kernels only) or LAV_CONV_BF16X6 (the split kernel where its plan
// wins); 0 in the descriptor = LAV_CONV_PRECISION (f32 | bf16x6), default bf16x6
int resolve_precision(const lav_conv &c) {
    if (c.precision == LAV_CONV_F32 || c.precision == LAV_CONV_BF16X6 || c.precision == LAV_CONV_F16X3) return c.precision;
    static const int dflt = [] {
        const char *e = getenv("LAV_CONV_PRECISION");
        return e && (!strcmp(e, "f32") || !strcmp(e, "fp32")) ? LAV_CONV_F32 : LAV_CONV_BF16X6;
    }();
    return dflt;
}

inline bool has_split_packing(int prec) { return prec == LAV_CONV_BF16X6 || prec == LAV_CONV_F16X3; }   // (F16X3 layers carry the bf16 pieces too: their fall-back)

// which kernel runs a layer: 0 tiled fp32, 1 direct fp32, 2 split bf16x6
struct Choice {
    int kind;
    DirectPlan dp;
    SplitPlan sp;
};

Choice decide(const lav_conv &c, const Plan &p, double tile_cost, double tile_raw) {
    Choice ch;
    ch.dp = choose_direct(c, p);
    ch.sp.ok = false;
    ch.kind = ch.dp.ok && ch.dp.cost < tile_cost ? 1 : 0;
    if (has_split_packing(resolve_precision(c))) {
        const char *e = getenv("LAV_CONV_SPLIT");   // 0 never / 1 by cost / 2 whenever the split kernel can take the layer
        const int mode = e ? atoi(e) : 1;
        if (mode) {
            ch.sp = choose_split(c, p);
            // the direct kernel's estimate is calibrated on small maps and runs ~30 % optimistic once a layer has several
            // hundred tiles (measured 64ch @ 160x160: 34.6 us against a model of 24.8; split kernel 26.1 against 25.0)
            double other = ch.kind == 1 ? ch.dp.cost * (1.0 + 0.3 * std::min(1.0, (double)ch.dp.tiles / 800.0)) : tile_raw;
            // deep 7x7 stems: the split kernel's tiles are LDS-bound there (64 pixels x 64 couts) and only match the tiled
            // fp32 kernel (measured 451 vs 431 us at 7 crops)
            const bool deep_stem = p.taps_per_class > 16 && c.cin >= 64 && !ch.sp.tp;   // (round 4: tap-pair plans hold 128-pixel tiles)
            if (ch.sp.ok && (mode == 2 || (ch.sp.cost < other && !deep_stem))) ch.kind = 2;
        }
    }
    // LAV_CONV_F16X3: wherever the split kernel runs the layer (round 5: the head convolution's plan only); on the fp32 kernels the
    // precision means nothing
    ch.sp.f16 = ch.kind == 2 && resolve_precision(c) == LAV_CONV_F16X3 && f16x3_layer(c, p) && !ch.sp.sk_w ? 1 : 0;
    if (smallcin_applies(c)) ch.kind = 3;   // camera stems: K = 3 x taps on packed fp32 FMAs (conv_smallcin.hpp)
    static const bool dbg = getenv("LAV_CONV_PLAN_DEBUG") != nullptr;
    if (dbg)
        fprintf(stderr, "[conv plan] B%d %d->%d k%dx%d s%d %dx%d%s: tiled %.1f us, direct %.1f us, split %.1f us -> %s\n", c.batch, c.cin, c.cout, c.kh, c.kw,
                c.stride, c.h, c.w, c.transposed ? " T" : "", tile_raw, ch.dp.ok ? ch.dp.cost : -1.0, ch.sp.ok ? ch.sp.cost : -1.0,
                ch.kind == 3 ? "small-cin" : ch.kind == 2 ? "split" : ch.kind == 1 ? "direct" : "tiled");
    return ch;
}
}  // namespace

extern "C" int lav_conv_tile_info(const lav_conv *c, int *info) {
    LAV_REQUIRE(c && info, "lav_conv_tile_info: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    ConvArgs a;
    int MP, MC;
    size_t lds;
    double cost = 0, raw = 0;
    rc = choose_tile(*c, p, a, MP, MC, lds, &cost, &raw);
    if (rc) return rc;
    const Choice ch = decide(*c, p, cost, raw);
    const DirectPlan &d = ch.dp;
    if (ch.kind == 3) {   // small-cin vector kernel: info[0] = -2, tile width, tile rows, workgroups; LDS; no split-K
        const int th = smallcin_tile_rows(*c);
        info[0] = -2; info[1] = SC_TW; info[2] = th; info[3] = ((p.OW + SC_TW - 1) / SC_TW) * ((p.OH + th - 1) / th) * c->batch; info[4] = 0;
        info[5] = 0; info[6] = 1; info[7] = 1; info[8] = 1;
        return LAV_OK;
    }
    if (ch.kind == 2) {   // split kernel: info[0] = -1, then MP, MC, pixel waves, tile width (0 = linearised), LDS, split-K, tap group, tile rows
        info[0] = -1; info[1] = ch.sp.MP; info[2] = ch.sp.MC; info[3] = ch.sp.WPX; info[4] = ch.sp.tw; info[5] = (int)ch.sp.lds;
        info[6] = ch.sp.sk_w ? -ch.sp.sk_w : ch.sp.ksplit; info[7] = ch.sp.tap_group + 100 * ch.sp.tp + 200 * ch.sp.f16; info[8] = ch.sp.th;   // (tap group + 100 in tap-pair mode; split-K < 0: stream-K over that many workgroups)
        return LAV_OK;
    }
    if (ch.kind == 1) {   // direct path: info[0] = 0, info[1] = waves per workgroup
        info[0] = 0; info[1] = d.waves; info[2] = d.mc; info[3] = 0; info[4] = 0; info[5] = d.waves * d.mc * 32 * 33 * 4;
        info[6] = d.ksplit; info[7] = 1; info[8] = 1;
        return LAV_OK;
    }
    info[0] = MP; info[1] = MC; info[2] = a.rowblock; info[3] = a.Wst; info[4] = a.ROWS; info[5] = (int)lds;
    info[6] = a.ksplit; info[7] = a.tap_group; info[8] = a.cps;
    return LAV_OK;
}

namespace {
template <int MP, int MC>
int launch(const ConvArgs &a, const Plan &p, int batch, size_t lds, hipStream_t st, float *amax_out = nullptr) {
    static bool attr_set = false;
    if (!attr_set) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv<MP, MC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int Q = p.QH * p.QW;
    dim3 grid(a.rowblock ? p.QH * a.xblocks : (Q + 128 * MP - 1) / (128 * MP), (a.cout + 32 * MC - 1) / (32 * MC), batch * p.nclasses * a.ksplit);
    const int tok = timer_begin("conv2d", st);
    static const bool want_trace = getenv("LAV_CONV_TRACE") != nullptr;
    if (want_trace) {  // debug: per-workgroup phase stamps of every 10th launch
        static unsigned long long *d_trace = nullptr;
        static int runs = 0;
        static bool attr2 = false;
        const size_t nwg = (size_t)grid.x * grid.y * grid.z;
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)65536 * 8 * sizeof(unsigned long long)));
        if (!attr2) {
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv<MP, MC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr2 = true;
        }
        if (nwg <= 65536) {
            ConvArgs at = a;
            at.trace = d_trace;
            hipLaunchKernelGGL((k_conv<MP, MC, true>), grid, dim3(256), lds, st, at);
            if (a.ksplit > 1) {
                const long total = (long)batch * a.cout * p.OH * p.OW;
                hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, batch, amax_out);
            }
            if (++runs % 10 == 0) {
                std::vector<unsigned long long> h(nwg * 8);
                if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (size_t i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 4]); }
                    double ph[4] = {0, 0, 0, 0}, last_start = 0;
                    for (size_t i = 0; i < nwg; ++i) {
                        for (int k = 0; k < 4; ++k) ph[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]) / 100.0;
                        last_start = std::max(last_start, (double)(h[i * 8] - t0) / 100.0);
                    }
                    fprintf(stderr, "[conv trace] %zu wgs (%ux%ux%u) lds %zu KB ksplit %d cps %d: span %.2f us, last start %.2f | mean us: setup+issue %.2f | first stage wait %.2f | main loop %.2f | epilogue %.2f\n",
                            nwg, grid.x, grid.y, grid.z, lds / 1024, a.ksplit, a.cps, (double)(t1 - t0) / 100.0, last_start, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg);
                }
            }
            timer_end(tok, st);
            LAV_LAUNCH_CHECK();
            return LAV_OK;
        }
    }
    hipLaunchKernelGGL((k_conv<MP, MC>), grid, dim3(256), lds, st, a);
    if (a.ksplit > 1) {
        const long total = (long)batch * a.cout * p.OH * p.OW;
        hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, batch, amax_out);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}
}  // namespace

extern "C" int lav_conv_out_hw(const lav_conv *c, int *oh, int *ow) {
    LAV_REQUIRE(c, "lav_conv_out_hw: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    if (oh) *oh = p.OH;
    if (ow) *ow = p.OW;
    return LAV_OK;
}

extern "C" size_t lav_conv_packed_weight_floats(const lav_conv *c) {
    if (!c) return 0;
    Plan p;
    if (build_plan(*c, p)) return 0;
    // the fp32 packing, followed (16-byte aligned) by the three-piece bf16 packing of the split kernel
    const int prec = resolve_precision(*c);
    size_t n = has_split_packing(prec) ? (p.wfloats + 3) / 4 * 4 + split_weight_bytes(p) / 4 : p.wfloats;
    if (prec == LAV_CONV_F16X3 && f16x3_layer(*c, p)) n += split_weight_bytes_f16(p) / 4 + 4;   // the fp16 pieces + their scale (16 bytes)
    return n;
}

extern "C" int lav_conv_pack_weights(const lav_conv *c, const float *h_weight, float *h_packed) {
    LAV_REQUIRE(c && h_weight && h_packed, "lav_conv_pack_weights: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    memset(h_packed, 0, p.wfloats * sizeof(float));
    for (int cls = 0; cls < p.nclasses; ++cls) {
        float *dst = h_packed + p.woff[cls];
        const auto &t = p.taps[cls];
        parallel_for((int)t.size(), [&, dst](int ti_) {
            const size_t ti = (size_t)ti_;
            for (int ci = 0; ci < c->cin; ++ci)
                for (int co = 0; co < c->cout; ++co) {
                    const size_t src = c->transposed
                                           ? (((size_t)ci * c->cout + co) * c->kh + t[ti].ky) * c->kw + t[ti].kx
                                           : (((size_t)co * c->cin + ci) * c->kh + t[ti].ky) * c->kw + t[ti].kx;
                    // [cout block][tap][8-channel group][lane = parity*32 + cout][pair]
                    dst[(size_t)(co / 32) * t.size() * p.cin_pad * 32 + ((ti * p.cin_pad + ci) / 8) * 256 + ((ci & 1) * 32 + co % 32) * 4 + (ci % 8) / 2] = h_weight[src];
                }
        });
    }
    if (has_split_packing(resolve_precision(*c)))
        split_pack_weights(*c, p, h_weight, reinterpret_cast<unsigned char *>(h_packed + (p.wfloats + 3) / 4 * 4));
    if (resolve_precision(*c) == LAV_CONV_F16X3 && f16x3_layer(*c, p))
        split_pack_weights_f16(*c, p, h_weight, reinterpret_cast<unsigned char *>(h_packed + (p.wfloats + 3) / 4 * 4 + split_weight_bytes(p) / 4));
    return LAV_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Re-packing on the device: where every slot of the packed buffer comes from, computed ONCE on the host by running the packers
// above over a weight tensor that holds its own indices (exact in fp32 below 2^24, and - the split packing stores x as three bf16
// pieces that sum to x exactly - also through the bf16x6 layout); a training run that evaluates its student through these
// kernels after every optimiser step then re-packs each layer with one gather launch from the live parameter in HBM.
extern "C" size_t lav_conv_pack_map_ints(const lav_conv *c) {
    if (!c) return 0;
    Plan p;
    if (build_plan(*c, p)) return 0;
    if (!has_split_packing(resolve_precision(*c))) return p.wfloats;
    // one entry per fp32 slot, then one per bf16 triple, then (LAV_CONV_F16X3) one per fp16 pair
    return (p.wfloats + 3) / 4 * 4 + split_weight_bytes(p) / 6 + (resolve_precision(*c) == LAV_CONV_F16X3 && f16x3_layer(*c, p) ? split_weight_bytes_f16(p) / 4 : 0);
}

extern "C" int lav_conv_pack_map(const lav_conv *c, int *h_map) {
    LAV_REQUIRE(c && h_map, "lav_conv_pack_map: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    const size_t nw = (size_t)c->cout * c->cin * c->kh * c->kw;
    LAV_REQUIRE(nw < (1u << 24), "lav_conv_pack_map: more than 2^24 weights");
    std::vector<float> iota(nw), packed(lav_conv_packed_weight_floats(c));
    for (size_t i = 0; i < nw; ++i) iota[i] = (float)(i + 1);
    rc = lav_conv_pack_weights(c, iota.data(), packed.data());
    if (rc) return rc;
    const bool split = has_split_packing(resolve_precision(*c));
    const size_t nf = split ? (p.wfloats + 3) / 4 * 4 : p.wfloats;
    for (size_t i = 0; i < nf; ++i) h_map[i] = i < p.wfloats ? (int)packed[i] - 1 : -1;
    if (split) {
        const unsigned short *o = reinterpret_cast<const unsigned short *>(packed.data() + nf);
        const size_t ntrip = split_weight_bytes(p) / 6;
        auto bf = [](unsigned short h) { const unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
        for (size_t j = 0; j < ntrip; ++j) {
            const size_t frag = j / 512, within = j % 512;
            const float v = (bf(o[frag * 1536 + within]) + bf(o[frag * 1536 + 512 + within])) + bf(o[frag * 1536 + 1024 + within]);
            h_map[nf + j] = (int)v - 1;
        }
        if (resolve_precision(*c) == LAV_CONV_F16X3 && f16x3_layer(*c, p)) {
            // the fp16 section: pair j = (fragment j / 512, slot j % 512) in split_pack_weights_f16's order, enumerated directly (two
            // fp16 pieces carry 22 bits: an index above 2^22 would not survive the round trip through the packer)
            int *m16 = h_map + nf + ntrip;
            const int nblk = p.cout_pad / 32, nchunks = p.cin_pad / 16;
            size_t fr = 0;
            for (int cls = 0; cls < p.nclasses; ++cls) {
                const auto &t = p.taps[cls];
                for (int blk = 0; blk < nblk; ++blk)
                    for (size_t ti = 0; ti < t.size(); ++ti)
                        for (int ch = 0; ch < nchunks; ++ch, ++fr)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int el = 0; el < 8; ++el) {
                                    const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + el;
                                    long src = -1;
                                    if (co < c->cout && ci < c->cin)
                                        src = c->transposed ? (((long)ci * c->cout + co) * c->kh + t[ti].ky) * c->kw + t[ti].kx
                                                            : (((long)co * c->cin + ci) * c->kh + t[ti].ky) * c->kw + t[ti].kx;
                                    m16[fr * 512 + lane * 8 + el] = (int)src;
                                }
            }
        }
    }
    return LAV_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_conv_repack(const float *__restrict__ w, const int *__restrict__ map, long nf, long ntrip,
                                                     float *__restrict__ packed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nf) {
        const int m = map[i];
        packed[i] = m >= 0 ? w[m] : 0.f;
    } else if (i < nf + ntrip) {
        const long j = i - nf;
        const int m = map[i];
        unsigned p0, p1, p2;
        split3(m >= 0 ? w[m] : 0.f, p0, p1, p2);
        unsigned short *o = reinterpret_cast<unsigned short *>(packed + nf);
        const long frag = j / 512, within = j % 512;
        o[frag * 1536 + within] = (unsigned short)(p0 >> 16);
        o[frag * 1536 + 512 + within] = (unsigned short)(p1 >> 16);
        o[frag * 1536 + 1024 + within] = (unsigned short)(p2 >> 16);
    }
}

// LAV_CONV_F16X3 section on the device (round 6: a trainer's forward / data-gradient convolutions on fp16 pieces): the scale is the
// power of two that puts the largest finite |w| (512 parts of launch_absmax_parts) into [16384, 32768) - split_pack_weights_f16's
// rule - then pair j = fp16(w / s), fp16(w / s - first piece); block 0 leaves the scale behind the section.
__global__ __launch_bounds__(256) void k_conv_repack_f16(const float *__restrict__ w, const int *__restrict__ map16, long npairs, const float *__restrict__ parts,
                                                         int nparts, unsigned char *__restrict__ out16) {
    __shared__ float s_m[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, parts[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    int e = 0;
    (void)frexpf(m, &e);
    const float sw = ldexpf(1.f, m > 0.f ? max(e, -100) - 15 : 0), inv = 1.f / sw;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j < npairs) {
        const int src = map16[j];
        const float v = src >= 0 ? w[src] * inv : 0.f;
        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
        _Float16 *o = reinterpret_cast<_Float16 *>(out16);
        const long frag = j >> 9, within = j & 511;
        o[frag * 1024 + within] = h0;
        o[frag * 1024 + 512 + within] = h1;
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) reinterpret_cast<float *>(out16 + npairs * 4)[threadIdx.x] = threadIdx.x == 0 ? sw : 0.f;
}

__global__ __launch_bounds__(256) void k_bn_fold(const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, double eps, int n, float *__restrict__ scale,
                                                 float *__restrict__ shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = (double)gamma[i] / sqrt((double)var[i] + eps);
    scale[i] = (float)s;
    shift[i] = (float)((double)beta[i] - (double)mean[i] * s);
}
}  // namespace

extern "C" int lav_conv_repack(const lav_conv *c, const float *d_weight, const int *d_map, float *d_packed, void *stream) {
    return lav_conv_repack_scratch(c, d_weight, d_map, d_packed, nullptr, 0, stream);
}

extern "C" int lav_conv_repack_scratch(const lav_conv *c, const float *d_weight, const int *d_map, float *d_packed, float *d_parts, size_t parts_floats,
                                       void *stream) {
    LAV_REQUIRE(c && d_weight && d_map && d_packed, "lav_conv_repack: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    const bool split = has_split_packing(resolve_precision(*c));
    const long nf = split ? (long)((p.wfloats + 3) / 4 * 4) : (long)p.wfloats, ntrip = split ? (long)(split_weight_bytes(p) / 6) : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_conv_repack, dim3((unsigned)((nf + ntrip + 255) / 256)), dim3(256), 0, st, d_weight, d_map, nf, ntrip, d_packed);
    if (resolve_precision(*c) == LAV_CONV_F16X3 && f16x3_layer(*c, p)) {
        // round 6: the fp16 section too - the weights' largest magnitude is measured on the device (512 parts, kept in the 2 KB behind
        // the section's scale word: lav_conv_packed_weight_floats reserves them), then one gather launch splits w / s into pieces
        LAV_REQUIRE(d_parts && parts_floats >= F16_PARTS, "lav_conv_repack: a LAV_CONV_F16X3 layer needs %d floats of scratch for the weights' maxima", F16_PARTS);
        const long npairs = (long)(split_weight_bytes_f16(p) / 4);
        const long nw = (long)c->cout * c->cin * c->kh * c->kw;
        int rc2 = launch_absmax_parts(d_weight, 1, 1, 0, 1, nw, d_parts, nullptr, st);
        if (rc2) return rc2;
        unsigned char *out16 = reinterpret_cast<unsigned char *>(d_packed + nf) + split_weight_bytes(p);
        hipLaunchKernelGGL(k_conv_repack_f16, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, d_weight, d_map + nf + ntrip, npairs, d_parts, F16_PARTS, out16);
    }
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_bn_fold(const float *mean, const float *var, const float *gamma, const float *beta, double eps, int n, float *scale,
                           float *shift, void *stream) {
    LAV_REQUIRE(mean && var && gamma && beta && scale && shift && n >= 1, "lav_bn_fold: bad argument");
    hipLaunchKernelGGL(k_bn_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mean, var, gamma, beta, eps, n,
                       scale, shift);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" size_t lav_conv_workspace_bytes(const lav_conv *c) {
    if (!c) return 0;
    Plan p;
    if (build_plan(*c, p)) return 0;
    ConvArgs a;
    int MP, MC;
    size_t lds;
    double cost = 0, raw = 0;
    if (choose_tile(*c, p, a, MP, MC, lds, &cost, &raw)) return 0;
    const Choice ch = decide(*c, p, cost, raw);
    if (ch.kind == 1) a.ksplit = ch.dp.ksplit;
    if (ch.kind == 2) a.ksplit = ch.sp.ksplit;
    if (ch.kind == 3) return 0;
    const int slabs = ch.kind == 2 && ch.sp.sk_w ? 2 : (a.ksplit > 1 ? a.ksplit : 0);   // stream-K: head and tail parts of the cut tiles
    // LAV_CONV_F16X3: F16_PARTS floats behind the slabs for the launch that measures x when no producer maxima are handed in
    return (size_t)slabs * c->batch * c->cout * p.OH * p.OW * sizeof(float) + (ch.kind == 2 && ch.sp.f16 ? (size_t)F16_PARTS * sizeof(float) : 0);
}

namespace {
// How many floats a launch of this plan leaves in amax_out, and whether its own kernels write them (else lav_conv2d measures y with
// one more launch: F16_PARTS floats).  Whole-K split / direct launches: one per workgroup; split-K: one per block of k_conv_reduce.
struct AmaxPlan { int count; bool in_kernel; };
AmaxPlan amax_plan(const lav_conv &c, const Plan &p, const Choice &ch, int tiled_ksplit) {
    const long reduce_blocks = ((long)c.batch * c.cout * p.OH * p.OW + 255) / 256;
    long n = 0;
    if (ch.kind == 2 && !ch.sp.sk_w) {
        const int NBLK = (4 / ch.sp.WPX) * ch.sp.MC;
        n = ch.sp.ksplit > 1 ? reduce_blocks : (long)((c.cout + NBLK * 32 - 1) / (NBLK * 32)) * ch.sp.tiles * c.batch * p.nclasses;
    } else if (ch.kind == 1) {
        n = ch.dp.ksplit > 1 ? reduce_blocks : (long)((c.batch * p.QH * p.QW + 31) / 32) * ((c.cout + 32 * ch.dp.mc - 1) / (32 * ch.dp.mc)) * p.nclasses;
    } else if (ch.kind == 0 && tiled_ksplit > 1) {
        n = reduce_blocks;
    }
    if (n < 1 || n > AMAX_MAX) return AmaxPlan{F16_PARTS, false};
    return AmaxPlan{(int)n, true};
}
}  // namespace

extern "C" int lav_conv_amax_count(const lav_conv *c) {
    if (!c) return 0;
    Plan p;
    if (build_plan(*c, p)) return 0;
    ConvArgs a;
    int MP, MC;
    size_t lds;
    double cost = 0, raw = 0;
    if (choose_tile(*c, p, a, MP, MC, lds, &cost, &raw)) return 0;
    const Choice ch = decide(*c, p, cost, raw);
    return amax_plan(*c, p, ch, a.ksplit).count;
}

extern "C" int lav_conv2d(const lav_conv *c, const float *x, const float *w_packed, const float *bias, const float *scale,
                          const float *shift, const float *residual, float *y, void *workspace, size_t workspace_bytes,
                          void *stream) {
    return lav_conv2d_amax(c, x, w_packed, bias, scale, shift, residual, y, workspace, workspace_bytes, nullptr, 0, nullptr, stream);
}

extern "C" int lav_conv2d_amax(const lav_conv *c, const float *x, const float *w_packed, const float *bias, const float *scale,
                               const float *shift, const float *residual, float *y, void *workspace, size_t workspace_bytes,
                               const float *amax_in, int amax_in_count, float *amax_out, void *stream) {
    LAV_REQUIRE(c && x && w_packed && y, "lav_conv2d: null argument");
    LAV_REQUIRE(!amax_in || amax_in_count >= 1, "lav_conv2d_amax: amax_in without a count");
    LAV_REQUIRE((scale == nullptr) == (shift == nullptr), "lav_conv2d: scale and shift go together");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.n_valid = lav::batch_limit();
    a.in_c_total = c->in_c_total; a.in_c_offset = c->in_c_offset; a.cin = c->cin; a.H = c->h; a.W = c->w;
    a.cout = c->cout; a.out_c_total = c->out_c_total; a.out_c_offset = c->out_c_offset; a.OH = p.OH; a.OW = p.OW;
    a.QH = p.QH; a.QW = p.QW; a.in_s = p.in_s; a.out_s = p.out_s;
    a.nclasses = p.nclasses; a.taps_per_class = p.taps_per_class;
    a.relu_pre = c->relu_pre; a.relu_post = c->relu_post; a.sigmoid = c->sigmoid;

    int MP, MC;
    size_t lds;
    double cost = 0, raw = 0;
    rc = choose_tile(*c, p, a, MP, MC, lds, &cost, &raw);
    if (rc) return rc;
    const Choice ch = decide(*c, p, cost, raw);
    const DirectPlan &dp = ch.dp;
    const bool direct = ch.kind == 1;
    if (direct) a.ksplit = dp.ksplit;
    if (ch.kind == 2) a.ksplit = ch.sp.ksplit;
    const int slabs = ch.kind == 2 && ch.sp.sk_w ? 2 : (a.ksplit > 1 ? a.ksplit : 0);
    const size_t slab_bytes = (size_t)slabs * c->batch * c->cout * p.OH * p.OW * sizeof(float);
    AmaxIO io{amax_in, amax_in_count, nullptr, nullptr};
    if (ch.kind == 2 && ch.sp.f16 && !amax_in) {   // the launch that measures x writes behind the slabs
        if (!workspace || workspace_bytes < slab_bytes + F16_PARTS * sizeof(float))
            return fail(LAV_EWORKSPACE, "lav_conv2d: workspace %zu < %zu bytes (fp16 scale)", workspace_bytes, slab_bytes + F16_PARTS * sizeof(float));
        io.scratch = reinterpret_cast<float *>(static_cast<char *>(workspace) + slab_bytes);
    }
    if (slabs) {
        if (!workspace || workspace_bytes < slab_bytes) return fail(LAV_EWORKSPACE, "lav_conv2d: workspace %zu < %zu bytes (split-K)", workspace_bytes, slab_bytes);
        a.partial = static_cast<float *>(workspace);
    } else {
        a.partial = nullptr;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the maxima of |y| for the next layer's fp16 scale: from this launch's own epilogue / reduce pass where the plan has one, else
    // measured behind it (ap.count floats either way: lav_conv_amax_count)
    const AmaxPlan ap = amax_plan(*c, p, ch, a.ksplit);
    io.out = amax_out && ap.in_kernel ? amax_out : nullptr;
    auto measure_y = [&]() -> int {
        if (!amax_out || ap.in_kernel) return LAV_OK;
        const int rcm = launch_absmax_parts(y, c->batch, c->out_c_total, c->out_c_offset, c->cout, (long)p.OH * p.OW, amax_out, lav::batch_limit(), st);
        if (rcm) return rcm;
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    };
    {
        a.pad_value = c->pad_value;
        a.trace = nullptr;
    }
    for (int i = 0; i < MAX_CLASSES; ++i) {
        const bool live = i < p.nclasses;
        a.cls_ntaps[i] = live ? (int)p.taps[i].size() : 0;
        a.cls_in_oy[i] = live ? p.in_oy[i] : 0; a.cls_in_ox[i] = live ? p.in_ox[i] : 0;
        a.cls_out_oy[i] = live ? p.out_oy[i] : 0; a.cls_out_ox[i] = live ? p.out_ox[i] : 0;
        a.cls_woff[i] = live ? (int)p.woff[i] : 0;
    }
    for (int i = 0; i < MAX_TAPS; ++i) a.toff[i] = 0;
    for (int cl = 0; cl < p.nclasses; ++cl)
        for (size_t t = 0; t < p.taps[cl].size(); ++t) a.toff[cl * p.taps_per_class + t] = p.taps[cl][t].dy * a.Wst + p.taps[cl][t].dx;

    if (ch.kind == 3) {
        rc = launch_smallcin(*c, p, a, st);
        return rc ? rc : measure_y();
    }
    if (ch.kind == 2) {
        const unsigned char *w_split = reinterpret_cast<const unsigned char *>(w_packed + (p.wfloats + 3) / 4 * 4);
        rc = launch_split(*c, p, ch.sp, a, w_split, st, ch.sp.f16 ? w_split + split_weight_bytes(p) : nullptr, io);
        return rc ? rc : measure_y();
    }
    if (direct) {
        DirectArgs d;
        d.x = x; d.w = w_packed; d.bias = bias; d.scale = scale; d.shift = shift; d.res = residual; d.y = y; d.partial = a.partial;
        d.n_valid = lav::batch_limit();
        d.in_c_total = c->in_c_total; d.in_c_offset = c->in_c_offset; d.H = c->h; d.W = c->w;
        d.cout = c->cout; d.out_c_total = c->out_c_total; d.out_c_offset = c->out_c_offset; d.OH = p.OH; d.OW = p.OW;
        d.cin_pad = p.cin_pad; d.cout_pad = p.cout_pad; d.dil_h = c->dil_h; d.dil_w = c->dil_w;
        d.QH = p.QH; d.QW = p.QW; d.in_s = p.in_s; d.out_s = p.out_s; d.nclasses = p.nclasses;
        int steps_gcd = 0;   // the load ring's depth must divide every class's step count
        for (int i = 0; i < MAX_CLASSES; ++i) {
            const bool live = i < p.nclasses;
            int nty = 0, ntx = 0;
            if (live && !p.taps[i].empty()) {
                ntx = 1;
                while (ntx < (int)p.taps[i].size() && p.taps[i][ntx].dy == p.taps[i][0].dy) ++ntx;   // taps are dy-major
                nty = (int)p.taps[i].size() / ntx;
            }
            d.cls_nty[i] = nty; d.cls_ntx[i] = ntx;
            d.cls_in_oy[i] = live ? p.in_oy[i] : 0; d.cls_in_ox[i] = live ? p.in_ox[i] : 0;
            d.cls_out_oy[i] = live ? p.out_oy[i] : 0; d.cls_out_ox[i] = live ? p.out_ox[i] : 0;
            d.cls_woff[i] = live ? (int)p.woff[i] : 0;
            const int st = nty * ntx * (dp.cw / 8);
            for (int x = st, y = steps_gcd; ; ) { if (!y) { steps_gcd = x; break; } const int t = x % y; x = y; y = t; }
        }
        d.M = c->batch * p.QH * p.QW; d.cw = dp.cw; d.cks = dp.cw * dp.waves; d.ksplit = dp.ksplit;
        d.relu_pre = c->relu_pre; d.relu_post = c->relu_post; d.sigmoid = c->sigmoid; d.pad_value = c->pad_value;
        d.amax_out = io.out;
        const int tok = timer_begin("conv2d", st);
        dim3 grid((d.M + 31) / 32, (c->cout + 32 * dp.mc - 1) / (32 * dp.mc), dp.ksplit * p.nclasses);
        const dim3 block(64 * dp.waves);
        const size_t lds_red = (size_t)dp.waves * dp.mc * 32 * 33 * 4;   // <= 66 KB
        int depth = dp.mc == 1 ? 9 : 6;   // ring registers: DEPTH * (4 * MC + 4)
        while (steps_gcd % depth) --depth;
        switch (depth * 10 + dp.mc) {
#define LAV_DIRECT_CASE(D, MC_) case D * 10 + MC_: { \
            static bool attr = false; \
            if (!attr) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_direct<D, MC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 32 * 33 * 4)); attr = true; } \
            hipLaunchKernelGGL((k_conv_direct<D, MC_>), grid, block, lds_red, st, d); } break;
            LAV_DIRECT_CASE(1, 1) LAV_DIRECT_CASE(2, 1) LAV_DIRECT_CASE(3, 1) LAV_DIRECT_CASE(4, 1) LAV_DIRECT_CASE(5, 1)
            LAV_DIRECT_CASE(6, 1) LAV_DIRECT_CASE(7, 1) LAV_DIRECT_CASE(8, 1) LAV_DIRECT_CASE(9, 1)
            LAV_DIRECT_CASE(1, 2) LAV_DIRECT_CASE(2, 2) LAV_DIRECT_CASE(3, 2) LAV_DIRECT_CASE(4, 2) LAV_DIRECT_CASE(5, 2) LAV_DIRECT_CASE(6, 2)
#undef LAV_DIRECT_CASE
        }
        if (dp.ksplit > 1) {
            const long total = (long)c->batch * c->cout * p.OH * p.OW;
            hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, c->batch, io.out);
        }
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        return measure_y();
    }
    if (MP == 2 && MC == 2) rc = launch<2, 2>(a, p, c->batch, lds, st, io.out);
    else if (MP == 1 && MC == 2) rc = launch<1, 2>(a, p, c->batch, lds, st, io.out);
    else rc = launch<1, 1>(a, p, c->batch, lds, st, io.out);
    return rc ? rc : measure_y();
}

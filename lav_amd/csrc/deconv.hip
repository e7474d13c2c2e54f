// Grouped ConvTranspose2d with a handful of output channels - the tails of the detection / segmentation heads
// (team_code_v2/models/lidar.py:30-33: ConvTranspose2d(64, out, 3, stride 2, padding 1, output_padding 1) of the
// center / box / ori / seg heads, fused here into ONE launch over the 4 x 64-channel feature map) and ERFNet's
// output layer (lav/models/erfnet.py:137, ConvTranspose2d(16, classes, 2, stride 2)).
//
// The implicit-GEMM kernel is the wrong tool for them: 9 output channels fill 9 of an MFMA tile's 32 rows, every output
// parity class re-stages the 26 MB input through LDS, and a block-diagonal weight multiplies by zeros for 3/4 of K
// (measured 90 us for the heads).  The layer is memory bound (read the input once: 26 MB, ~5 us), so this is a plain
// vector-ALU kernel:
//   thread      = one input-grid position q = (n, qy, qx) of one channel GROUP; it owns the S x S output pixels
//                 (S*qy + ry - pad, S*qx + rx - pad) of all the group's output channels (<= NC) in registers
//   per channel : ceil(K/S)^2 coalesced input loads (q and its upper / left neighbours) and K*K*nc FMAs; the weights of
//                 a group are wave-uniform and come through the scalar cache
//   epilogue    : + bias -> sigmoid on channels >= sigmoid_from -> NCHW store
// Groups never share output channels, so there is no reduction and the result is deterministic.
#include <cstdlib>

#include "common.hpp"

namespace {
using namespace lav;

struct DeconvArgs {
    const float *x, *w, *bias;
    float *y;
    int B, cin, H, W, cout, OH, OW, QH, QW;
    int cin_g;          // input channels per group
    int pad, sigmoid_from;
    int cout_off[9];    // group g writes output channels [cout_off[g], cout_off[g+1])
    int w_off[8];       // float offset of group g's weights [cin_g][nc_g][K][K] (PyTorch ConvTranspose2d layout)
};

// U = channels per batch of loads: all NJ*NJ*U gathers of a batch are requested before its first FMA (the loop is a chain of
// L2 round trips: 64 channels at U = 2 were 32 of them)
template <int K, int S, int NC, int U>
__global__ __launch_bounds__(256) void k_deconv_grouped(DeconvArgs a) {
    constexpr int NJ = (K + S - 1) / S;   // input rows / columns one output parity class can reach
    const int g = blockIdx.y;
    const int co0 = a.cout_off[g], nc = a.cout_off[g + 1] - co0;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_img = (long)a.QH * a.QW;
    if (q >= a.B * per_img) return;
    const int n = (int)(q / per_img);
    const int r = (int)(q - n * per_img);
    const int qy = r / a.QW, qx = r - qy * a.QW;

    // the NJ x NJ input pixels this thread reads in every channel (rows qy - jy, columns qx - jx); out of range -> 0
    int off[NJ][NJ];
    bool ok[NJ][NJ];
#pragma unroll
    for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx) {
            const int iy = qy - jy, ix = qx - jx;
            ok[jy][jx] = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            off[jy][jx] = ok[jy][jx] ? iy * a.W + ix : 0;
        }
    float acc[S][S][NC];
#pragma unroll
    for (int ry = 0; ry < S; ++ry)
#pragma unroll
        for (int rx = 0; rx < S; ++rx)
#pragma unroll
            for (int co = 0; co < NC; ++co) acc[ry][rx][co] = 0.f;

    const long cplane = (long)a.H * a.W;
    const float *xg = a.x + ((long)n * a.cin + (long)g * a.cin_g) * cplane;
    const float *wg = a.w + a.w_off[g];   // wave-uniform
    for (int c0 = 0; c0 < a.cin_g; c0 += U) {
        float v[U][NJ][NJ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(c0 + u, a.cin_g - 1);   // (a ragged last batch re-reads the last channel and skips its FMAs)
#pragma unroll
            for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
                for (int jx = 0; jx < NJ; ++jx) {
                    const float t = xg[c * cplane + off[jy][jx]];
                    v[u][jy][jx] = ok[jy][jx] ? t : 0.f;
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u < a.cin_g) {   // wave-uniform
                const float *wc = wg + (long)(c0 + u) * nc * (K * K);
#pragma unroll
                for (int co = 0; co < NC; ++co) {
                    if (co < nc) {   // wave-uniform
#pragma unroll
                        for (int ky = 0; ky < K; ++ky)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx)   // output parity (ky % S, kx % S), input row qy - ky / S
                                acc[ky % S][kx % S][co] = fmaf(v[u][ky / S][kx / S], wc[co * (K * K) + ky * K + kx], acc[ky % S][kx % S][co]);
                    }
                }
            }
        }
    }
    const long oplane = (long)a.OH * a.OW;
    if (a.sigmoid_from == -2) {   // softmax over the group's channels (ERFNet's class scores, lav_agent_fast.py:264): all in registers
#pragma unroll
        for (int ry = 0; ry < S; ++ry)
#pragma unroll
            for (int rx = 0; rx < S; ++rx) {
                const int oy = S * qy + ry - a.pad, ox = S * qx + rx - a.pad;
                if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
                float v[NC], m = -INFINITY, sum = 0.f;
#pragma unroll
                for (int co = 0; co < NC; ++co) {
                    v[co] = co < nc ? acc[ry][rx][co] + (a.bias ? a.bias[co0 + co] : 0.f) : -INFINITY;
                    m = fmaxf(m, v[co]);
                }
#pragma unroll
                for (int co = 0; co < NC; ++co) { v[co] = co < nc ? expf(v[co] - m) : 0.f; sum += v[co]; }
#pragma unroll
                for (int co = 0; co < NC; ++co)
                    if (co < nc) a.y[((long)n * a.cout + co0 + co) * oplane + (long)oy * a.OW + ox] = v[co] / sum;
            }
        return;
    }
#pragma unroll
    for (int co = 0; co < NC; ++co) {
        if (co >= nc) break;
        const int cg = co0 + co;
        const float b = a.bias ? a.bias[cg] : 0.f;
        float *yo = a.y + ((long)n * a.cout + cg) * oplane;
#pragma unroll
        for (int ry = 0; ry < S; ++ry)
#pragma unroll
            for (int rx = 0; rx < S; ++rx) {
                const int oy = S * qy + ry - a.pad, ox = S * qx + rx - a.pad;
                if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
                float v = acc[ry][rx][co] + b;
                if (a.sigmoid_from >= 0 && cg >= a.sigmoid_from) v = 1.f / (1.f + expf(-v));
                yo[(long)oy * a.OW + ox] = v;
            }
    }
}

// Round 5: the 3x3 stride-2 case (the four head tails: 26 MB in, 3.7 MB out) as a STAGED kernel.  The gathering kernel above is a
// chain of L2 round trips at 1.6 waves per SIMD (47 us for the heads = 0.08 of the HBM roof), its weights arrive through the scalar
// cache behind a full lgkmcnt(0) wait per batch.  Here a workgroup owns a tile of 8 x 32 input-grid positions of one group: the
// (8 + 1) x (32 + 1) input pixels its positions read are streamed through LDS 16 channels at a time with coalesced 16-byte row loads
// (unconditional, from clamped addresses; the next chunk's loads are in flight while this chunk is multiplied), the group's weights
// sit in LDS as one 16-byte aligned row per channel (read as broadcasts), a thread reads its four pixels per channel from LDS and
// multiplies two output channels per instruction (v_pk_fma_f32).  Same products in the same order as the gathering kernel (channels
// ascending, taps ascending inside a channel): bit-identical results (tests/test_gpu_conv.py).
constexpr int DT_Y = 8, DT_X = 32, DT_CH = 16, DT_RS = DT_X + 4, DT_ROWS = DT_Y + 1;   // row: [3] = left halo, [4, 36) = the tile's columns
constexpr int DT_MAXC = 64;                                                            // channels per group the weight rows are sized for
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v2f_a4 __attribute__((ext_vector_type(2), aligned(4)));   // a pair at a 4-byte aligned LDS address: two 32-bit reads (ds_read2_b32)

template <int NC>
__global__ __launch_bounds__(256) void k_deconv_tile(DeconvArgs a, int tiles_y) {
    constexpr int K = 3, S = 2, NW = (K * K * NC + 3) / 4 * 4;   // floats per weight row [tap][NC], padded to 16 bytes
    // (+ 4 floats: the last column's (own, right) pair reads one element past its row - the next row's first pad word, or this tail)
    __shared__ __attribute__((aligned(16))) float s_in_flat[2 * DT_CH * DT_ROWS * DT_RS + 4];
    float (*s_in)[DT_CH][DT_ROWS][DT_RS] = reinterpret_cast<float (*)[DT_CH][DT_ROWS][DT_RS]>(s_in_flat);
    __shared__ __attribute__((aligned(16))) float s_w[DT_MAXC][NW];
    const int g = blockIdx.z;
    const int n = blockIdx.y / tiles_y, qy0 = (blockIdx.y - n * tiles_y) * DT_Y, qx0 = blockIdx.x * DT_X;
    const int tid = threadIdx.x, tx = tid & (DT_X - 1), ty = tid / DT_X;
    const int co0 = a.cout_off[g], nc = a.cout_off[g + 1] - co0;
    const long cplane = (long)a.H * a.W;
    const float *xg = a.x + ((long)n * a.cin + (long)g * a.cin_g) * cplane;
    const float *wg = a.w + a.w_off[g];
    const int nchunk = (a.cin_g + DT_CH - 1) / DT_CH;

    // what this thread stages per chunk: 16-byte pieces idx = tid + 256 i of the DT_CH x DT_ROWS row segments (8 per segment) and, for
    // tid < DT_CH * DT_ROWS, the segment's left halo pixel.  Every load is unconditional from a clamped address; what lies outside
    // the image (or past the last channel) is masked to zero on its way into LDS - not at the load, which would wait for it.
    constexpr int NV4 = DT_CH * DT_ROWS * (DT_X / 4), NI = (NV4 + 255) / 256;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 rv[NI];
    unsigned rh = 0, keep[NI], keeph = 0;
    auto fetch = [&](int k) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = min(tid + 256 * i, NV4 - 1), seg = idx / (DT_X / 4), v4 = idx - seg * (DT_X / 4);
            const int c = seg / DT_ROWS, r = seg - c * DT_ROWS;
            const int iy = qy0 - 1 + r, ix = qx0 + 4 * v4, ch = k * DT_CH + c;
            keep[i] = (iy >= 0 && iy < a.H && ix < a.W && ch < a.cin_g) ? 0xffffffffu : 0u;
            rv[i] = *reinterpret_cast<const u4 *>(xg + (long)min(ch, a.cin_g - 1) * cplane + (long)min(max(iy, 0), a.H - 1) * a.W + min(ix, a.W - 4));
        }
        {
            const int seg = min(tid, DT_CH * DT_ROWS - 1), c = seg / DT_ROWS, r = seg - c * DT_ROWS;
            const int iy = qy0 - 1 + r, ix = qx0 - 1, ch = k * DT_CH + c;
            keeph = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ch < a.cin_g) ? 0xffffffffu : 0u;
            rh = __float_as_uint(xg[(long)min(ch, a.cin_g - 1) * cplane + (long)min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)]);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NV4) {
                const int seg = idx / (DT_X / 4), v4 = idx - seg * (DT_X / 4), c = seg / DT_ROWS, r = seg - c * DT_ROWS;
                *reinterpret_cast<u4 *>(&s_in[buf][c][r][4 + 4 * v4]) = rv[i] & keep[i];
            }
        }
        if (tid < DT_CH * DT_ROWS) s_in[buf][tid / DT_ROWS][tid % DT_ROWS][3] = __uint_as_float(rh & keeph);
    };

    fetch(0);
    // the group's weights [cin_g][nc][3][3] -> one row [tap][NC] per channel (output channels past nc: zero, never multiplied)
    for (int i = tid; i < a.cin_g * NW; i += 256) {
        const int c = i / NW, j = i - c * NW, tap = j / NC, co = j - tap * NC;
        s_w[c][j] = (tap < K * K && co < nc) ? wg[((long)c * nc + co) * (K * K) + tap] : 0.f;
    }
    stage(0);
    __syncthreads();

    // acc[parity][co pair]: two output channels per packed multiply-add
    constexpr int NP = (NC + 1) / 2;
    v2f acc[S][S][NP];
#pragma unroll
    for (int ry = 0; ry < S; ++ry)
#pragma unroll
        for (int rx = 0; rx < S; ++rx)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[ry][rx][p] = v2f{0.f, 0.f};

    for (int k = 0; k < nchunk; ++k) {
        const bool more = k + 1 < nchunk;
        if (more) fetch(k + 1);
        const int buf = k & 1, cn = min(DT_CH, a.cin_g - k * DT_CH);
        auto channel = [&](int c) __attribute__((always_inline)) {
            // input pixels (qy - jy, qx - jx), each as the LOW half of a register pair: the packed multiply-add then broadcasts it with
            // op_sel_hi = 0 only.  (A value sitting in the HIGH register of a pair would be selected with op_sel = 1 - the form that
            // returns wrong lanes beside matrix + LDS neighbours on gfx950, DESIGN 4.4c / profiles/r05_coresidency.md; a CPU test
            // disassembles the library and rejects it.)
            v2f v[2][2];
            v[0][1] = *reinterpret_cast<const v2f_a4 *>(&s_in[buf][c][ty + 1][3 + tx]);   // (left, own): left in the low half
            v[0][0] = *reinterpret_cast<const v2f_a4 *>(&s_in[buf][c][ty + 1][4 + tx]);   // (own, right)
            v[1][1] = *reinterpret_cast<const v2f_a4 *>(&s_in[buf][c][ty][3 + tx]);
            v[1][0] = *reinterpret_cast<const v2f_a4 *>(&s_in[buf][c][ty][4 + tx]);
            float wrow[NW];
            const float4 *wp = reinterpret_cast<const float4 *>(&s_w[k * DT_CH + c][0]);   // the same address in every lane: a broadcast read
#pragma unroll
            for (int q = 0; q < NW / 4; ++q) { const float4 t = wp[q]; wrow[4 * q] = t.x; wrow[4 * q + 1] = t.y; wrow[4 * q + 2] = t.z; wrow[4 * q + 3] = t.w; }
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {   // output parity (ky % S, kx % S), input row qy - ky / S
                    const v2f xp = v[ky / S][kx / S];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int t0 = (ky * K + kx) * NC + 2 * p;
                        const v2f w2 = v2f{wrow[t0], 2 * p + 1 < NC ? wrow[t0 + 1] : 0.f};
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[ky % S][kx % S][p]) : "v"(xp), "v"(w2));
                    }
                }
        };
        if (cn == DT_CH) {
#pragma unroll 4
            for (int c = 0; c < DT_CH; ++c) channel(c);
        } else {
            for (int c = 0; c < cn; ++c) channel(c);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    const int qy = qy0 + ty, qx = qx0 + tx;
    if (qy >= a.QH || qx >= a.QW) return;
    const long oplane = (long)a.OH * a.OW;
#pragma unroll
    for (int co = 0; co < NC; ++co) {
        if (co >= nc) break;
        const int cg = co0 + co;
        const float b = a.bias ? a.bias[cg] : 0.f;
        float *yo = a.y + ((long)n * a.cout + cg) * oplane;
#pragma unroll
        for (int ry = 0; ry < S; ++ry)
#pragma unroll
            for (int rx = 0; rx < S; ++rx) {
                const int oy = S * qy + ry - a.pad, ox = S * qx + rx - a.pad;
                if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
                float v = acc[ry][rx][co / 2][co % 2] + b;
                if (a.sigmoid_from >= 0 && cg >= a.sigmoid_from) v = 1.f / (1.f + expf(-v));
                yo[(long)oy * a.OW + ox] = v;
            }
    }
}

template <int K, int S, int NC>
void launch(const DeconvArgs &a, int groups, hipStream_t st) {
    if constexpr (K == 3 && S == 2) {
        // staged kernel: rows of whole 16-byte pieces at 16-byte aligned addresses, no softmax epilogue (LAV_DECONV_IMPL=gather: the old one)
        const char *impl = getenv("LAV_DECONV_IMPL");
        const bool gather = impl && impl[0] == 'g';
        if (!gather && a.W % 4 == 0 && a.W >= 4 && a.cin_g <= DT_MAXC && reinterpret_cast<uintptr_t>(a.x) % 16 == 0 && a.sigmoid_from != -2) {
            const int tiles_x = (a.QW + DT_X - 1) / DT_X, tiles_y = (a.QH + DT_Y - 1) / DT_Y;
            hipLaunchKernelGGL((k_deconv_tile<NC>), dim3(tiles_x, tiles_y * a.B, groups), dim3(256), 0, st, a, tiles_y);
            return;
        }
    }
    const long nq = (long)a.B * a.QH * a.QW;
    static const int unroll = [] { const char *e = getenv("LAV_DECONV_UNROLL"); return e ? atoi(e) : 8; }();
    const dim3 grid((unsigned)((nq + 255) / 256), groups);
    if (unroll >= 8) hipLaunchKernelGGL((k_deconv_grouped<K, S, NC, 8>), grid, dim3(256), 0, st, a);
    else if (unroll >= 4) hipLaunchKernelGGL((k_deconv_grouped<K, S, NC, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_deconv_grouped<K, S, NC, 2>), grid, dim3(256), 0, st, a);
}
}  // namespace

extern "C" int lav_deconv_grouped(int batch, int cin, int h, int w, int groups, const int *cout_per_group, int kernel, int stride,
                                  int pad, int out_pad, const float *x, const float *weight, const float *bias, int sigmoid_from,
                                  float *y, void *stream) {
    LAV_REQUIRE(batch >= 1 && cin >= 1 && h >= 1 && w >= 1 && cout_per_group && x && weight && y, "lav_deconv_grouped: bad argument");
    LAV_REQUIRE(groups >= 1 && groups <= 8 && cin % groups == 0, "lav_deconv_grouped: %d groups over %d channels", groups, cin);
    LAV_REQUIRE((kernel == 3 && stride == 2) || (kernel == 2 && stride == 2), "lav_deconv_grouped: kernel %d stride %d unsupported (3/2 and 2/2 are)", kernel, stride);
    LAV_REQUIRE(pad >= 0 && pad < kernel && out_pad >= 0 && out_pad < stride, "lav_deconv_grouped: bad padding");
    DeconvArgs a;
    a.x = x; a.w = weight; a.bias = bias; a.y = y;
    a.B = batch; a.cin = cin; a.H = h; a.W = w; a.cin_g = cin / groups; a.pad = pad; a.sigmoid_from = sigmoid_from;
    a.OH = (h - 1) * stride - 2 * pad + kernel + out_pad;
    a.OW = (w - 1) * stride - 2 * pad + kernel + out_pad;
    LAV_REQUIRE(a.OH >= 1 && a.OW >= 1, "lav_deconv_grouped: empty output");
    a.QH = (a.OH - 1 + pad) / stride + 1;
    a.QW = (a.OW - 1 + pad) / stride + 1;
    int nc_max = 0, woff = 0;
    a.cout_off[0] = 0;
    for (int g = 0; g < groups; ++g) {
        LAV_REQUIRE(cout_per_group[g] >= 1 && cout_per_group[g] <= 8, "lav_deconv_grouped: group %d has %d output channels (1..8)", g, cout_per_group[g]);
        a.cout_off[g + 1] = a.cout_off[g] + cout_per_group[g];
        a.w_off[g] = woff;
        woff += a.cin_g * cout_per_group[g] * kernel * kernel;
        nc_max = nc_max > cout_per_group[g] ? nc_max : cout_per_group[g];
    }
    for (int g = groups; g < 8; ++g) { a.cout_off[g + 1] = a.cout_off[groups]; a.w_off[g] = 0; }
    a.cout = a.cout_off[groups];
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("deconv_grouped", st);
    if (kernel == 3) {
        if (nc_max <= 2) launch<3, 2, 2>(a, groups, st);
        else if (nc_max <= 4) launch<3, 2, 4>(a, groups, st);
        else launch<3, 2, 8>(a, groups, st);
    } else {
        if (nc_max <= 2) launch<2, 2, 2>(a, groups, st);
        else if (nc_max <= 4) launch<2, 2, 4>(a, groups, st);
        else launch<2, 2, 8>(a, groups, st);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

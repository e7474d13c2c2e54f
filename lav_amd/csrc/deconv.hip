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

template <int K, int S, int NC>
void launch(const DeconvArgs &a, int groups, hipStream_t st) {
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

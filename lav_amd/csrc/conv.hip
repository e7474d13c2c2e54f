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
#include <vector>

#include "common.hpp"

namespace {
using namespace lav;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_TAPS = 64;
constexpr int MAX_CLASSES = 16;

struct ConvArgs {
    const float *x, *w, *bias, *scale, *shift, *res;
    float *y;
    int in_c_total, in_c_offset, cin, H, W;
    int cout, out_c_total, out_c_offset, OH, OW;
    int QH, QW, in_s, out_s;
    int CK, Wst, ROWS, nclasses, taps_per_class;
    int relu_pre, relu_post, sigmoid;
    int cls_ntaps[MAX_CLASSES], cls_in_oy[MAX_CLASSES], cls_in_ox[MAX_CLASSES];
    int cls_out_oy[MAX_CLASSES], cls_out_ox[MAX_CLASSES], cls_woff[MAX_CLASSES];
    int toff[MAX_TAPS];  // class c, tap t -> toff[c*taps_per_class + t] = dy*Wst + dx
};

template <int MP, int MC>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CO_T = 32 * MC, PIXW = 128 * MP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int cls = blockIdx.z % a.nclasses, n = blockIdx.z / a.nclasses;
    const int cb = blockIdx.y * CO_T;
    const int Q = a.QH * a.QW;
    const int q0 = blockIdx.x * PIXW;
    const int CK = a.CK, Wst = a.Wst, ROWS = a.ROWS, plane = ROWS * Wst;
    float *s_in = smem;                                  // [CK][ROWS][Wst]
    float *s_w = smem + ((CK * plane + 3) & ~3);         // [ntaps][CK][CO_T]
    const int ntaps = a.cls_ntaps[cls];
    const int qy0 = q0 / a.QW;
    const int iy_base = qy0 * a.in_s + a.cls_in_oy[cls];
    const int in_ox = a.cls_in_ox[cls];
    const float *wbase = a.w + a.cls_woff[cls];
    const int *toff = a.toff + cls * a.taps_per_class;

    int base[MP];
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) {
        const int q = min(q0 + (wid * MP + mp) * 32 + l31, Q - 1);
        const int qy = q / a.QW, qx = q - qy * a.QW;
        base[mp] = (qy - qy0) * a.in_s * Wst + qx * a.in_s;
    }
    f32x16 acc[MC][MP];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int mp = 0; mp < MP; ++mp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mc][mp][r] = 0.f;

    const float *xin = a.x + ((long)n * a.in_c_total + a.in_c_offset) * a.H * a.W;
    for (int ci0 = 0; ci0 < a.cin; ci0 += CK) {
        __syncthreads();
        for (int pr = wid; pr < CK * ROWS; pr += 4) {
            const int c = pr / ROWS, rr = pr - c * ROWS;
            const int ci = ci0 + c, iy = iy_base + rr;
            const bool rowok = ci < a.cin && iy >= 0 && iy < a.H;
            const float *src = xin + ((long)ci * a.H + (rowok ? iy : 0)) * a.W;
            float *dst = s_in + c * plane + rr * Wst;
            for (int xx = lane; xx < Wst; xx += 64) {
                const int ix = in_ox + xx;
                dst[xx] = (rowok && ix >= 0 && ix < a.W) ? src[ix] : 0.f;
            }
        }
        for (int pr = wid; pr < ntaps * CK; pr += 4) {
            const int tap = pr / CK, c = pr - tap * CK;
            const int ci = ci0 + c;
            const float *src = wbase + ((long)tap * a.cin + (ci < a.cin ? ci : 0)) * a.cout + cb;
            float *dst = s_w + pr * CO_T;
            for (int j = lane; j < CO_T; j += 64) dst[j] = (ci < a.cin && cb + j < a.cout) ? src[j] : 0.f;
        }
        __syncthreads();
        for (int tap = 0; tap < ntaps; ++tap) {
            const int to = toff[tap];
            const float *wt = s_w + tap * CK * CO_T + l31;
            for (int cp = 0; cp < CK; cp += 2) {
                const int c = cp + half;
                float av[MC], bv[MP];
#pragma unroll
                for (int mc = 0; mc < MC; ++mc) av[mc] = wt[c * CO_T + mc * 32];
#pragma unroll
                for (int mp = 0; mp < MP; ++mp) bv[mp] = s_in[c * plane + base[mp] + to];
#pragma unroll
                for (int mc = 0; mc < MC; ++mc)
#pragma unroll
                    for (int mp = 0; mp < MP; ++mp)
                        acc[mc][mp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mc], bv[mp], acc[mc][mp], 0, 0, 0);
            }
        }
    }

    const int out_oy = a.cls_out_oy[cls], out_ox = a.cls_out_ox[cls];
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) {
        const int q = q0 + (wid * MP + mp) * 32 + l31;
        if (q >= Q) continue;
        const int qy = q / a.QW, qx = q - qy * a.QW;
        const int oy = qy * a.out_s + out_oy, ox = qx * a.out_s + out_ox;
        if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb + mc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co >= a.cout) continue;
                float v = acc[mc][mp][r];
                if (a.bias) v += a.bias[co];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
                const long idx = (((long)n * a.out_c_total + a.out_c_offset + co) * a.OH + oy) * a.OW + ox;
                if (a.res) v += a.res[idx];
                if (a.relu_post) v = v > 0.f ? v : 0.f;
                if (a.sigmoid) v = 1.f / (1.f + expf(-v));
                a.y[idx] = v;
            }
        }
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
    size_t off = 0;
    for (auto &t : p.taps) {
        p.taps_per_class = std::max<int>(p.taps_per_class, (int)t.size());
        for (auto &tp : t) { p.max_dy = std::max(p.max_dy, tp.dy); p.max_dx = std::max(p.max_dx, tp.dx); }
        p.woff.push_back(off);
        off += t.size() * (size_t)c.cin * c.cout;
    }
    p.wfloats = off;
    LAV_REQUIRE(p.taps_per_class * p.nclasses <= MAX_TAPS, "lav_conv: %d taps x %d classes exceed %d", p.taps_per_class, p.nclasses, MAX_TAPS);
    return LAV_OK;
}

template <int MP, int MC>
int launch(const ConvArgs &a, const Plan &p, int batch, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv<MP, MC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int Q = p.QH * p.QW;
    dim3 grid((Q + 128 * MP - 1) / (128 * MP), (a.cout + 32 * MC - 1) / (32 * MC), batch * p.nclasses);
    const int tok = timer_begin("conv2d", st);
    hipLaunchKernelGGL((k_conv<MP, MC>), grid, dim3(256), lds, st, a);
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
    return p.wfloats;
}

extern "C" int lav_conv_pack_weights(const lav_conv *c, const float *h_weight, float *h_packed) {
    LAV_REQUIRE(c && h_weight && h_packed, "lav_conv_pack_weights: null");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    for (int cls = 0; cls < p.nclasses; ++cls) {
        float *dst = h_packed + p.woff[cls];
        const auto &t = p.taps[cls];
        for (size_t ti = 0; ti < t.size(); ++ti)
            for (int ci = 0; ci < c->cin; ++ci)
                for (int co = 0; co < c->cout; ++co) {
                    const size_t src = c->transposed
                                           ? (((size_t)ci * c->cout + co) * c->kh + t[ti].ky) * c->kw + t[ti].kx
                                           : (((size_t)co * c->cin + ci) * c->kh + t[ti].ky) * c->kw + t[ti].kx;
                    dst[(ti * c->cin + ci) * c->cout + co] = h_weight[src];
                }
    }
    return LAV_OK;
}

extern "C" int lav_conv2d(const lav_conv *c, const float *x, const float *w_packed, const float *bias, const float *scale,
                          const float *shift, const float *residual, float *y, void *stream) {
    LAV_REQUIRE(c && x && w_packed && y, "lav_conv2d: null argument");
    LAV_REQUIRE((scale == nullptr) == (shift == nullptr), "lav_conv2d: scale and shift go together");
    Plan p;
    int rc = build_plan(*c, p);
    if (rc) return rc;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.in_c_total = c->in_c_total; a.in_c_offset = c->in_c_offset; a.cin = c->cin; a.H = c->h; a.W = c->w;
    a.cout = c->cout; a.out_c_total = c->out_c_total; a.out_c_offset = c->out_c_offset; a.OH = p.OH; a.OW = p.OW;
    a.QH = p.QH; a.QW = p.QW; a.in_s = p.in_s; a.out_s = p.out_s;
    a.nclasses = p.nclasses; a.taps_per_class = p.taps_per_class;
    a.relu_pre = c->relu_pre; a.relu_post = c->relu_post; a.sigmoid = c->sigmoid;

    // tile shape: the largest tile that still gives the 256 CUs >= 2 workgroups each
    const long Q = (long)p.QH * p.QW;
    auto nwg = [&](int mp, int mc) { return ((Q + 128 * mp - 1) / (128 * mp)) * ((c->cout + 32 * mc - 1) / (32 * mc)) * c->batch * p.nclasses; };
    int MP = 1, MC = c->cout > 32 ? 2 : 1;
    if (MC == 2 && nwg(2, 2) >= 512) MP = 2;
    if (MC == 2 && MP == 1 && nwg(1, 2) < 256 && c->cout % 64 != 0) MC = 1;
    if (MC == 2 && MP == 1 && nwg(1, 2) < 200) MC = 1;
    const int PIXW = 128 * MP, CO_T = 32 * MC;

    a.Wst = (p.QW - 1) * p.in_s + p.max_dx + 1;
    const int span_rows = (int)std::min<long>((PIXW - 1 + p.QW - 1) / p.QW + 1, p.QH);
    a.ROWS = (span_rows - 1) * p.in_s + p.max_dy + 1;
    // cin chunk: as large as fits ~64 KB of LDS (2 workgroups per CU), at least 2
    const int cin_even = (c->cin + 1) & ~1;
    int CK = 16;
    auto lds_bytes = [&](int ck) { return (size_t)((((size_t)ck * a.ROWS * a.Wst + 3) & ~(size_t)3) + (size_t)p.taps_per_class * ck * CO_T) * 4; };
    while (CK > 2 && (lds_bytes(CK) > 64 * 1024 || CK > cin_even)) CK >>= 1;
    LAV_REQUIRE(lds_bytes(CK) <= 160 * 1024, "lav_conv2d: tile needs %zu bytes of LDS", lds_bytes(CK));
    a.CK = CK;
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

    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = lds_bytes(CK);
    if (MP == 2 && MC == 2) return launch<2, 2>(a, p, c->batch, lds, st);
    if (MP == 1 && MC == 2) return launch<1, 2>(a, p, c->batch, lds, st);
    return launch<1, 1>(a, p, c->batch, lds, st);
}

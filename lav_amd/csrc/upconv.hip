// Transposed convolutions whose kernel size equals their stride ("pointwise" up-convolutions): every output pixel has exactly ONE
// contributing input pixel and tap,  y[co][s iy + ky - pad][s ix + kx - pad] = sum_ci W[ci][co][ky][kx] x[ci][iy][ix],  i.e. k*k independent
// [cout x cin] x [cin x pixels] matrix products whose results interleave in the output.  Two of the BEV backbone's three up-convolutions
// are of this kind (team_code_v2/models/lidar.py:114-131: upconv1 = ConvTranspose2d(64, 128, 1, 1), upconv3 = ConvTranspose2d(128, 128,
// 4, 4, 1, 2), each followed by ReLU and an eval BatchNorm, written into their slice of the 384-channel feature map).
//
// The implicit-GEMM kernels are the wrong tool for them (round 6 trace of the lidar graph, tools/graph_replay.py): a K loop of 4-8 steps
// never fills their loader / compute pipeline - 21.5 us for the 1x1 layer on the direct fp32 kernel, 31.1 us for the 4x4 stride-4 layer
// on the split kernel, for 0.42 / 0.84 GFLOP and 13 MB of output each.  This kernel has no pipeline to fill:
//   wave        = 32 input pixels x FOUR 32x32 accumulator tiles: the four kx taps of one ky and one 32-cout tile (stride 4), or four
//                 32-cout tiles (1x1).  Its whole B operand - 32 pixels x CIN channels - is loaded once into CIN/2 registers per lane
//                 (coalesced: a channel plane's 32 consecutive pixels per half wave), all loads in flight at once
//   workgroup   = 4 waves on 4 pixel groups sharing one (ky, cout tile): its 4 x CIN x 32 weights (32 / 64 KB, contiguous in the packed
//                 buffer) go to LDS once; the A fragments are conflict-free ds_read_b32 (32 consecutive couts per half wave)
//   loop        = CIN/2 steps of 4 v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (no operand splitting, no scales)
//   epilogue    = ReLU -> x scale + shift (eval BatchNorm) -> NCHW stores into the channel slice; the workgroup's largest finite
//                 |value| goes to amax_out[workgroup] (the feature map's bound for LAV_CONV_F16X3 readers, lav_conv2d_amax)
// Padding: out-of-range output positions (4 iy + ky - 1 = -1) are dropped; output positions past the last input pixel (output_padding:
// row / column 159 of the 160 x 160 map) have no contribution and hold epilogue(0) - the kernel walks an input grid extended by one
// row and column of zeros, so they are written like every other pixel.
// Bound: the output write (13 MB per layer) and 6.8 / 3.4 us of fp32 matrix work per wave at the fp32 MFMA rate.
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace {
using namespace lav;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct UpArgs {
    const float *__restrict__ x, *__restrict__ w, *__restrict__ scale, *__restrict__ shift;
    float *__restrict__ y, *__restrict__ amax_out;
    int B, COUT, IH, IW, OH, OW, EH, EW, pad, relu_pre, out_c_total, out_c_offset;
};

template <int S, int CIN, int PAD>
__global__ __launch_bounds__(256) void k_upconv_pointwise(UpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // [4 slots][CIN][32 couts]
    __shared__ float s_max[4];
    __shared__ float s_ss[2][128];   // scale / shift of the workgroup's couts (through LDS: the epilogue's stores would otherwise order its loads)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, kh = lane >> 5;
    const int ct = blockIdx.y;                           // S = 4: 32-cout tile; S = 1: 128-cout tile
    const int n = blockIdx.z / S, ky = blockIdx.z - n * S;
    const int e = (blockIdx.x * 4 + wid) * 32 + l31;     // pixel of the extended input grid
    const int iy = e / a.EW, ix = e - iy * a.EW;
    const bool in_grid = e < a.EH * a.EW, in_ok = in_grid && iy < a.IH && ix < a.IW;
    const long plane = (long)a.IH * a.IW;
    // the wave's B operand: channel 2 kk + kh of its 32 pixels (every load issued before anything is waited for)
    const float *xp = a.x + ((long)n * CIN + kh) * plane + (in_ok ? iy * a.IW + ix : 0);
    float b[CIN / 2];
#pragma unroll
    for (int kk = 0; kk < CIN / 2; ++kk) b[kk] = xp[(long)(2 * kk) * plane];
    // the workgroup's weights: one contiguous block of the packed buffer
    {
        const float4 *wb = reinterpret_cast<const float4 *>(a.w + ((long)ky * gridDim.y + ct) * (4 * CIN * 32));
        float4 *dst = reinterpret_cast<float4 *>(s_w);
#pragma unroll
        for (int i = 0; i < CIN * 32 / 256; ++i) dst[tid + 256 * i] = wb[tid + 256 * i];
    }
    {
        constexpr int NCO = S == 1 ? 128 : 32;
        if (tid < NCO) { s_ss[0][tid] = a.scale[blockIdx.y * NCO + tid]; s_ss[1][tid] = a.shift[blockIdx.y * NCO + tid]; }
    }
#pragma unroll
    for (int kk = 0; kk < CIN / 2; ++kk) b[kk] = in_ok ? b[kk] : 0.f;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float *wl = s_w + kh * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < CIN / 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(j * CIN + 2 * kk) * 32], b[kk], acc[j], 0, 0, 0);
    }
    // epilogue: this lane holds pixel l31 of the group, couts (r & 3) + 8 (r >> 2) + 4 kh of each slot's tile
    float m = 0.f;
    const long oplane = (long)a.OH * a.OW;
    float *yn = a.y + ((long)n * a.out_c_total + a.out_c_offset) * oplane;
    if constexpr (S == 1) {
        const long pix = (long)iy * a.OW + ix;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, co = ct * 128 + cl;
                float v = acc[j][r];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                v = fmaf(v, s_ss[0][cl], s_ss[1][cl]);
                if (in_ok) {
                    yn[co * oplane + pix] = v;
                    m = fmaxf(m, finite_abs(v));
                }
            }
    } else {
        // Slot j = tap kx: output column S ix + j - PAD.  The aligned group of four columns [S ix, S ix + 3] is this pixel's taps PAD..3 and the
        // NEXT pixel's taps 0..PAD-1 (the next lane's, same row): one 16-byte store per lane and cout instead of four 4-byte ones at a
        // 16-byte pitch.  Lanes without a right-hand neighbour in the wave / the row, and groups that cross the map's edge, store their
        // own values one by one; taps 0..PAD-1 of a lane whose left-hand neighbour did not take them are stored by the lane itself.
        const int oy = S * iy + ky - PAD, ox0 = S * ix - PAD;
        const bool row_ok = in_grid && oy >= 0 && oy < a.OH;
        const bool vec = row_ok && l31 < 31 && ix + 1 < a.EW && e + 1 < a.EH * a.EW && S * ix + 3 < a.OW && (a.OW & 3) == 0;
        const bool prev_vec = __shfl_up((int)vec, 1, 32) != 0 && l31 > 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * kh, co = ct * 32 + cl;
            const float sc = s_ss[0][cl], sh = s_ss[1][cl];
            float *row = yn + co * oplane + (long)oy * a.OW;
            float v[4], nx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[j][r];
                if (a.relu_pre) v[j] = v[j] > 0.f ? v[j] : 0.f;
                v[j] = fmaf(v[j], sc, sh);
                if (row_ok && ox0 + j >= 0 && ox0 + j < a.OW) m = fmaxf(m, finite_abs(v[j]));
                nx[j] = j < PAD ? __shfl_down(v[j], 1, 32) : 0.f;
            }
            if (vec) {
                float4 g;
                g.x = 0 < 4 - PAD ? v[(PAD + 0) & 3] : nx[(0 + PAD) & 3];
                g.y = 1 < 4 - PAD ? v[(PAD + 1) & 3] : nx[(1 + PAD) & 3];
                g.z = 2 < 4 - PAD ? v[(PAD + 2) & 3] : nx[(2 + PAD) & 3];
                g.w = 3 < 4 - PAD ? v[(PAD + 3) & 3] : nx[(3 + PAD) & 3];
                *reinterpret_cast<float4 *>(row + S * ix) = g;
            } else if (row_ok) {
#pragma unroll
                for (int j = PAD; j < 4; ++j)
                    if (ox0 + j < a.OW) row[ox0 + j] = v[j];
            }
            if (!prev_vec && row_ok) {
#pragma unroll
                for (int j = 0; j < PAD; ++j)
                    if (ox0 + j >= 0 && ox0 + j < a.OW) row[ox0 + j] = v[j];
            }
        }
    }
    if (a.amax_out) {
        m = wave_finite_absmax(m);
        if (lane == 0) s_max[wid] = m;
        __syncthreads();
        if (tid == 0) a.amax_out[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
}

struct UpGeom {
    int S, OH, OW, EH, EW, gx, gy, gz;
};
inline bool up_geom(int batch, int cin, int cout, int ih, int iw, int k, int pad, int out_pad, UpGeom &g) {
    if (!(k == 1 || k == 4) || !(cin == 64 || cin == 128) || batch < 1 || ih < 1 || iw < 1 || pad < 0 || pad >= k || out_pad < 0 || out_pad >= std::max(k, 2)) return false;
    if (k == 1 ? (cout < 128 || cout % 128 || pad || out_pad) : (cout < 32 || cout % 32)) return false;
    g.S = k;
    g.OH = (ih - 1) * k - 2 * pad + k + out_pad; g.OW = (iw - 1) * k - 2 * pad + k + out_pad;
    if (g.OH < 1 || g.OW < 1) return false;
    // the input grid extended so that every output position has an owner: iy up to (OH - 1 + pad) / k
    g.EH = (g.OH - 1 + pad) / k + 1; g.EW = (g.OW - 1 + pad) / k + 1;
    g.gx = (g.EH * g.EW + 127) / 128;
    g.gy = k == 1 ? cout / 128 : cout / 32;
    g.gz = batch * k;
    return true;
}
}  // namespace

extern "C" size_t lav_upconv_pointwise_packed_floats(int cin, int cout, int k) {
    return (k == 1 || k == 4) && cin > 0 && cout > 0 ? (size_t)k * k * cin * cout : 0;
}

// PyTorch ConvTranspose2d weight [cin][cout][k][k] -> blocks of [4 slots][cin][32 couts]:
// k = 4: block (ky, ct), slot j = kx;  k = 1: block ct (128 couts), slot j = 32-cout tile
extern "C" int lav_upconv_pointwise_pack(int cin, int cout, int k, const float *h_weight, float *h_packed) {
    LAV_REQUIRE(h_weight && h_packed && (k == 1 || k == 4) && cin >= 1 && cout >= 1 && cout % (k == 1 ? 128 : 32) == 0,
                "lav_upconv_pointwise_pack: kernel 1 (cout a multiple of 128) or 4 (cout a multiple of 32)");
    const int nct = k == 1 ? cout / 128 : cout / 32;
    for (int ky = 0; ky < k; ++ky)
        for (int ct = 0; ct < nct; ++ct)
            for (int j = 0; j < 4; ++j)
                for (int ci = 0; ci < cin; ++ci)
                    for (int c = 0; c < 32; ++c) {
                        const int co = k == 1 ? (ct * 4 + j) * 32 + c : ct * 32 + c, kx = k == 1 ? 0 : j;
                        h_packed[((((size_t)ky * nct + ct) * 4 + j) * cin + ci) * 32 + c] = h_weight[(((size_t)ci * cout + co) * k + ky) * k + kx];
                    }
    return LAV_OK;
}

extern "C" int lav_upconv_pointwise_parts(int batch, int cin, int cout, int ih, int iw, int k, int pad, int out_pad) {
    UpGeom g;
    return up_geom(batch, cin, cout, ih, iw, k, pad, out_pad, g) ? g.gx * g.gy * g.gz : 0;
}

extern "C" int lav_upconv_pointwise(int batch, int cin, int cout, int ih, int iw, int k, int pad, int out_pad, const float *x, const float *w_packed,
                                    const float *scale, const float *shift, int relu_pre, int out_c_total, int out_c_offset, float *y,
                                    float *amax_out, void *stream) {
    UpGeom g;
    LAV_REQUIRE(up_geom(batch, cin, cout, ih, iw, k, pad, out_pad, g),
                "lav_upconv_pointwise: kernel == stride in {1, 4}, cin in {64, 128}, cout a multiple of 128 (kernel 1) / 32 (kernel 4)");
    LAV_REQUIRE(x && w_packed && scale && shift && y, "lav_upconv_pointwise: null argument");
    LAV_REQUIRE(out_c_total >= cout && out_c_offset >= 0 && out_c_offset + cout <= out_c_total, "lav_upconv_pointwise: channel slice outside the output");
    LAV_REQUIRE((long)cout * g.OH * g.OW < (1l << 31) && (long)ih * iw < (1l << 30), "lav_upconv_pointwise: map too large");
    UpArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y; a.amax_out = amax_out;
    a.B = batch; a.COUT = cout; a.IH = ih; a.IW = iw; a.OH = g.OH; a.OW = g.OW; a.EH = g.EH; a.EW = g.EW; a.pad = pad; a.relu_pre = relu_pre ? 1 : 0;
    a.out_c_total = out_c_total; a.out_c_offset = out_c_offset;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(g.gx, g.gy, g.gz);
    const size_t lds = (size_t)4 * cin * 32 * sizeof(float);
    const int tok = timer_begin("upconv_pointwise", st);
#define LAV_UP_CASE(S_, CIN_, PAD_) if (k == S_ && cin == CIN_ && pad == PAD_) { \
        static bool attr = false; \
        if (!attr) { LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_upconv_pointwise<S_, CIN_, PAD_>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr = true; } \
        hipLaunchKernelGGL((k_upconv_pointwise<S_, CIN_, PAD_>), grid, dim3(256), lds, st, a); }
    LAV_UP_CASE(1, 64, 0) LAV_UP_CASE(1, 128, 0)
    LAV_UP_CASE(4, 64, 0) LAV_UP_CASE(4, 64, 1) LAV_UP_CASE(4, 64, 2) LAV_UP_CASE(4, 64, 3)
    LAV_UP_CASE(4, 128, 0) LAV_UP_CASE(4, 128, 1) LAV_UP_CASE(4, 128, 2) LAV_UP_CASE(4, 128, 3)
#undef LAV_UP_CASE
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

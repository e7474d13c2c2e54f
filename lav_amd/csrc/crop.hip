// Rotated crop of the BEV feature map around each actor: affine_grid + bilinear grid_sample in one kernel.
//
// Replaces crop_feature of the reference (team_code_v2/model_inference.py:204-238, uniplanner.py:310-352), which
// materialises an (N, 96, 96, 2) sampling grid with a batched matmul and then runs torch's generic grid sampler
// (measured 370-400 us per call here).  One thread computes the source position of one output pixel once and then
// walks the 384 channels: 4 gathers + 1 coalesced store per channel.
//
//   theta = [[k cos, -k sin, tx], [k sin, k cos, ty]],  k = crop/H,
//   tx = -k ox cos + k oy sin + ox + loc_x * ppm/(H/2),   ty = -k ox sin - k oy cos + oy + loc_y * ppm/(W/2)
//   grid (align_corners=True): xs = linspace(-1, 1, crop)[x], ys likewise;  gx = t00 xs + t01 ys + t02 ...
//   sample (align_corners=True): ix = (gx + 1)/2 * (W-1), bilinear, zeros outside.
#include "common.hpp"

namespace {
using namespace lav;

__device__ __forceinline__ float lin(int i, int n) {
    // torch.linspace(-1, 1, n): start + step*i in the first half, end - step*(n-1-i) in the second
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// BACKWARD=false: out[n][c][y][x] = bilinear(feat[map(n)][c]);  BACKWARD=true: feat_or_grad[map(n)][c] += weights * out[n][c][y][x]
// (`out` is then the incoming gradient, read only).  map(n) = map_index[n] when given, n when there is one map per crop, else 0.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_crop_rotate(float *__restrict__ feat, int feat_batch, const int *__restrict__ map_index, int C,
                                                     int H, int W, const float *__restrict__ locs, const float *__restrict__ oris,
                                                     float ppm, int crop, float ox, float oy, int c_per_block,
                                                     float *__restrict__ out, const int *__restrict__ n_valid) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.z;
    if (n_valid && n >= *n_valid) return;   // lav_batch_limit
    if (pix >= crop * crop) return;
    const int y = pix / crop, x = pix - y * crop;
    const float o = oris[n];
    const float cs = cosf(o), sn = sinf(o);
    const float k = (float)crop / (float)H;
    const float rx = locs[n * 2 + 0] * ppm / ((float)H / 2.f);
    const float ry = locs[n * 2 + 1] * ppm / ((float)W / 2.f);
    const float t02 = -k * ox * cs + k * oy * sn + ox + rx;
    const float t12 = -k * ox * sn - k * oy * cs + oy + ry;
    const float xs = lin(x, crop), ys = lin(y, crop);
    const float gx = k * cs * xs + (k * -sn) * ys + t02;
    const float gy = k * sn * xs + k * cs * ys + t12;
    const float ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    const float iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    // grid_sampler's weights: nw = (x1-ix)(y1-iy), ne = (ix-x0)(y1-iy), sw = (x1-ix)(iy-y0), se = (ix-x0)(iy-y0)
    const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    const float w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const long plane = (long)H * W;
    const int m = map_index ? map_index[n] : (feat_batch > 1 ? n : 0);
    float *f = feat + (long)m * C * plane;
    const int c_lo = blockIdx.y * c_per_block, c_hi = min(C, c_lo + c_per_block);
    float *o_ = out + ((long)n * C) * crop * crop + pix;
    if constexpr (!BACKWARD) {
#pragma unroll 8   // 32 independent gathers in flight per thread: the loop is latency bound, not bandwidth bound
        for (int c = c_lo; c < c_hi; ++c) {
            const float *p = f + c * plane;
            const float v = p[cy0 * W + cx0] * w00 + p[cy0 * W + cx1] * w01 + p[cy1 * W + cx0] * w10 + p[cy1 * W + cx1] * w11;
            o_[(long)c * crop * crop] = v;
        }
    } else {
        // transpose of the gather: every output-pixel gradient is spread over its four source pixels (crops overlap and
        // share maps, hence atomics - the same choice torch's grid_sampler backward makes)
        for (int c = c_lo; c < c_hi; ++c) {
            float *p = f + c * plane;
            const float g = o_[(long)c * crop * crop];
            if (w00 != 0.f) atomicAdd(p + cy0 * W + cx0, g * w00);
            if (w01 != 0.f) atomicAdd(p + cy0 * W + cx1, g * w01);
            if (w10 != 0.f) atomicAdd(p + cy1 * W + cx0, g * w10);
            if (w11 != 0.f) atomicAdd(p + cy1 * W + cx1, g * w11);
        }
    }
}
}  // namespace

namespace {
int crop_launch(bool backward, float *feat, int nmaps, const int *map_index, int C, int H, int W, const float *locs, const float *oris,
                int n, float ppm, int crop, float ox, float oy, float *out, hipStream_t st, const char *what) {
    const int c_per_block = 32;
    dim3 grid((crop * crop + 255) / 256, (C + c_per_block - 1) / c_per_block, n);
    const int tok = timer_begin(what, st);
    if (backward)
        hipLaunchKernelGGL(k_crop_rotate<true>, grid, dim3(256), 0, st, feat, nmaps, map_index, C, H, W, locs, oris, ppm, crop, ox, oy, c_per_block, out, backward ? nullptr : lav::batch_limit());
    else
        hipLaunchKernelGGL(k_crop_rotate<false>, grid, dim3(256), 0, st, feat, nmaps, map_index, C, H, W, locs, oris, ppm, crop, ox, oy, c_per_block, out, backward ? nullptr : lav::batch_limit());
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}
}  // namespace

extern "C" int lav_crop_rotate(const float *feat, int feat_batch, int C, int H, int W, const float *locs, const float *oris,
                               int n, float pixels_per_meter, int crop, float offset_x, float offset_y, float *out,
                               void *stream) {
    LAV_REQUIRE(n >= 0 && C > 0 && H > 1 && W > 1 && crop > 1, "lav_crop_rotate: bad sizes");
    if (n == 0) return LAV_OK;
    LAV_REQUIRE(feat && locs && oris && out, "lav_crop_rotate: null argument");
    LAV_REQUIRE(feat_batch == 1 || feat_batch == n, "lav_crop_rotate: feat_batch must be 1 or n");
    return crop_launch(false, const_cast<float *>(feat), feat_batch, nullptr, C, H, W, locs, oris, n, pixels_per_meter, crop, offset_x,
                       offset_y, out, static_cast<hipStream_t>(stream), "crop_rotate");
}

extern "C" int lav_crop_rotate_indexed(const float *feat, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                                       const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                                       float *out, void *stream) {
    LAV_REQUIRE(n >= 0 && num_maps >= 1 && C > 0 && H > 1 && W > 1 && crop > 1, "lav_crop_rotate_indexed: bad sizes");
    if (n == 0) return LAV_OK;
    LAV_REQUIRE(feat && map_index && locs && oris && out, "lav_crop_rotate_indexed: null argument");
    return crop_launch(false, const_cast<float *>(feat), num_maps, map_index, C, H, W, locs, oris, n, pixels_per_meter, crop, offset_x,
                       offset_y, out, static_cast<hipStream_t>(stream), "crop_rotate");
}

extern "C" int lav_crop_rotate_backward(const float *grad_out, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                                        const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                                        float *grad_feat, void *stream) {
    LAV_REQUIRE(n >= 0 && num_maps >= 1 && C > 0 && H > 1 && W > 1 && crop > 1, "lav_crop_rotate_backward: bad sizes");
    LAV_REQUIRE(grad_feat, "lav_crop_rotate_backward: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LAV_HIP(hipMemsetAsync(grad_feat, 0, (size_t)num_maps * C * H * W * sizeof(float), st));
    if (n == 0) return LAV_OK;
    LAV_REQUIRE(grad_out && map_index && locs && oris, "lav_crop_rotate_backward: null argument");
    return crop_launch(true, grad_feat, num_maps, map_index, C, H, W, locs, oris, n, pixels_per_meter, crop, offset_x, offset_y,
                       const_cast<float *>(grad_out), st, "crop_rotate_backward");
}

// Rotated crop of the BEV feature map around each actor: affine_grid + bilinear grid_sample in one kernel.
//
// Replaces crop_feature of the reference (team_code_v2/model_inference.py:204-238, uniplanner.py:310-352), which
// materialises an (N, 96, 96, 2) sampling grid with a batched matmul and then runs torch's generic grid sampler
// (measured 370-400 us per call here).  One thread computes the source position of one output pixel once and then
// walks the 384 channels: 4 corner reads + 1 coalesced store per channel.  Two kernels with identical arithmetic (and bits):
// k_crop_rotate_staged - 16 x 16 output tiles, the touched box of the map staged through LDS (the common geometry, map pitch
// ~ output pitch) - and k_crop_rotate, which gathers the corners from L2 (any geometry).  Training: lav_crop_rotate_indexed
// (crop i from map map_index[i]) and lav_crop_rotate_backward (gather form, below).
//
//   theta = [[k cos, -k sin, tx], [k sin, k cos, ty]],  k = crop/H,
//   tx = -k ox cos + k oy sin + ox + loc_x * ppm/(H/2),   ty = -k ox sin - k oy cos + oy + loc_y * ppm/(W/2)
//   grid (align_corners=True): xs = linspace(-1, 1, crop)[x], ys likewise;  gx = t00 xs + t01 ys + t02 ...
//   sample (align_corners=True): ix = (gx + 1)/2 * (W-1), bilinear, zeros outside.
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace {
using namespace lav;

__device__ __forceinline__ float lin(int i, int n) {
    // torch.linspace(-1, 1, n): start + step*i in the first half, end - step*(n-1-i) in the second
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// out[n][c][y][x] = bilinear(feat[map(n)][c]);  map(n) = map_index[n] when given, n when there is one map per crop, else 0.
__global__ __launch_bounds__(256) void k_crop_rotate(const float *__restrict__ feat, int feat_batch, const int *__restrict__ map_index, int C,
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
    const float *f = feat + (long)m * C * plane;
    const int c_lo = blockIdx.y * c_per_block, c_hi = min(C, c_lo + c_per_block);
    float *o_ = out + ((long)n * C) * crop * crop + pix;
    {
#pragma unroll 8   // 32 independent gathers in flight per thread: the loop is latency bound, not bandwidth bound
        for (int c = c_lo; c < c_hi; ++c) {
            const float *p = f + c * plane;
            const float v = p[cy0 * W + cx0] * w00 + p[cy0 * W + cx1] * w01 + p[cy1 * W + cx0] * w10 + p[cy1 * W + cx1] * w11;
            o_[(long)c * crop * crop] = v;
        }
    }
}

// The same crop with the source pixels staged through LDS.  A workgroup owns a 16 x 16 tile of one crop's output; the map
// pixels its samples touch lie in a (rotated) box of at most ~27 x 27 pixels, read with coalesced row loads eight channels at
// a time, so the four corner reads of every output pixel hit LDS instead of 8+ L1 lines per wave instruction.  Corners,
// weights and the order of the four products are the unstaged kernel's: the outputs are bit-identical.
constexpr int FWD_TILE = 16;
constexpr int FWD_SUB = 8;       // channels staged at a time
constexpr int FWD_CAP = 1024;    // floats per staged channel
constexpr int FWD_CPB = 32;      // channels per workgroup

struct SamplePos {
    float w00, w01, w10, w11;
    int cx0, cx1, cy0, cy1;
};

__device__ __forceinline__ SamplePos sample_pos(int x, int y, int crop, int H, int W, float k, float cs, float sn, float t02, float t12,
                                                float &ix, float &iy) {
    const float xs = lin(x, crop), ys = lin(y, crop);
    const float gx = k * cs * xs + (k * -sn) * ys + t02;
    const float gy = k * sn * xs + k * cs * ys + t12;
    ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    SamplePos p;
    p.w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f; p.w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    p.w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f; p.w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    p.cx0 = min(max(x0, 0), W - 1); p.cx1 = min(max(x1, 0), W - 1);
    p.cy0 = min(max(y0, 0), H - 1); p.cy1 = min(max(y1, 0), H - 1);
    return p;
}

__global__ __launch_bounds__(256, 3) void k_crop_rotate_staged(const float *__restrict__ feat, int feat_batch, const int *__restrict__ map_index,
                                                               int C, int H, int W, const float *__restrict__ locs,
                                                               const float *__restrict__ oris, float ppm, int crop, float ox, float oy,
                                                               float *__restrict__ out, const int *__restrict__ n_valid, int cpb, int nslab_crops) {
    __shared__ float s_f[FWD_SUB][FWD_CAP];
    // Workgroup -> (crop, channel block, tile), XCD aware (round 6): consecutive workgroups go to the eight XCDs in turn, and the tiles of
    // one (crop, channel block) slab read overlapping boxes of the same 32 map planes - dealt out in launch order every XCD fetched its
    // own copy of every halo from the memory side (2.85x the slab).  Here a slab's tiles all land on ONE XCD (slab = 8 (l / (8 tiles)) +
    // l % 8, tile = (l / 8) % tiles): the overlaps are hits in that XCD's L2.
    const int tiles_x = (crop + FWD_TILE - 1) / FWD_TILE, tiles = tiles_x * tiles_x;
    const int cblocks = (C + cpb - 1) / cpb;
    const int l = blockIdx.x;
    const int slab = 8 * (l / (8 * tiles)) + (l & 7), tile = (l >> 3) % tiles;
    const int n = slab / cblocks, cblk = slab - n * cblocks;
    if (n >= nslab_crops) return;                          // (the launch is padded to whole groups of eight slabs)
    if (n_valid && n >= *n_valid) return;   // lav_batch_limit (workgroup-uniform)
    const int tid = threadIdx.x;
    const int ty0 = (tile / tiles_x) * FWD_TILE, tx0 = (tile % tiles_x) * FWD_TILE;
    const int x = tx0 + (tid & (FWD_TILE - 1)), y = ty0 + tid / FWD_TILE;
    const bool live = x < crop && y < crop;
    const float o = oris[n];
    const float cs = cosf(o), sn = sinf(o);
    const float k = (float)crop / (float)H;
    const float rx = locs[n * 2 + 0] * ppm / ((float)H / 2.f);
    const float ry = locs[n * 2 + 1] * ppm / ((float)W / 2.f);
    const float t02 = -k * ox * cs + k * oy * sn + ox + rx;
    const float t12 = -k * ox * sn - k * oy * cs + oy + ry;
    float ix, iy;
    const SamplePos sp = sample_pos(min(x, crop - 1), min(y, crop - 1), crop, H, W, k, cs, sn, t02, t12, ix, iy);
    // box of the tile's samples: the sample position is affine in (x, y), so its extremes are at the tile's corners (one pixel of
    // margin for rounding); workgroup-uniform
    float ex[4], ey[4];
    const int xe = min(tx0 + FWD_TILE - 1, crop - 1), ye = min(ty0 + FWD_TILE - 1, crop - 1);
    sample_pos(tx0, ty0, crop, H, W, k, cs, sn, t02, t12, ex[0], ey[0]);
    sample_pos(xe, ty0, crop, H, W, k, cs, sn, t02, t12, ex[1], ey[1]);
    sample_pos(tx0, ye, crop, H, W, k, cs, sn, t02, t12, ex[2], ey[2]);
    sample_pos(xe, ye, crop, H, W, k, cs, sn, t02, t12, ex[3], ey[3]);
    const float fx_lo = floorf(fminf(fminf(ex[0], ex[1]), fminf(ex[2], ex[3]))), fx_hi = floorf(fmaxf(fmaxf(ex[0], ex[1]), fmaxf(ex[2], ex[3])));
    const float fy_lo = floorf(fminf(fminf(ey[0], ey[1]), fminf(ey[2], ey[3]))), fy_hi = floorf(fmaxf(fmaxf(ey[0], ey[1]), fmaxf(ey[2], ey[3])));
    // (float clamps first: positions far outside the map must not overflow the integer conversion)
    const int bx0 = (int)fminf(fmaxf(fx_lo - 1.f, 0.f), (float)(W - 1)), bx1 = (int)fminf(fmaxf(fx_hi + 2.f, 0.f), (float)(W - 1));
    const int by0 = (int)fminf(fmaxf(fy_lo - 1.f, 0.f), (float)(H - 1)), by1 = (int)fminf(fmaxf(fy_hi + 2.f, 0.f), (float)(H - 1));
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    if (bw * bh > FWD_CAP || bw > 32) __builtin_trap();   // the host only picks this kernel when the box always fits (crop_fwd_staged_ok)
    // this thread's four corners inside the box; a corner outside it (cannot happen by construction) is read from the map itself
    const bool boxed = sp.cx0 >= bx0 && sp.cx1 <= bx1 && sp.cy0 >= by0 && sp.cy1 <= by1;
    const int o00 = (sp.cy0 - by0) * bw + (sp.cx0 - bx0), o01 = (sp.cy0 - by0) * bw + (sp.cx1 - bx0);
    const int o10 = (sp.cy1 - by0) * bw + (sp.cx0 - bx0), o11 = (sp.cy1 - by0) * bw + (sp.cx1 - bx0);
    const long plane = (long)H * W, cc = (long)crop * crop;
    const int m = map_index ? map_index[n] : (feat_batch > 1 ? n : 0);
    const int c_lo = cblk * cpb, nch = min(cpb, C - c_lo);   // cpb: channels per workgroup (8, 16 or 32: crop_launch)
    const float *f = feat + ((long)m * C + c_lo) * plane;
    float *o_ = out + ((long)n * C + c_lo) * cc + (long)y * crop + x;
    // Staging (round 6).  Thread (c = tid / 32, column = tid % 32) fetches column `column` of channel c's box: at most 32 rows, ALL
    // requested before the first one is used, from clamped addresses (a load under an exec-mask branch is waited for inside the
    // branch).  The first version walked the box in seven batches of four loads, each batch a dependent L2 round trip: 5 us of
    // latency per eight channels, which is what the kernel's 86 us for seven crops were.  The next eight channels are requested
    // before this group's samples are computed, so their round trip runs under the LDS reads and the output stores.
    const int l32 = tid & 31, sc = tid >> 5;
    if (bh > 32) __builtin_trap();   // (crop_fwd_staged_ok: the box of a 16 x 16 tile is at most 28 x 28)
    const float *fcol = f + (long)by0 * W + bx0 + min(l32, bw - 1);
    float stg[32];
    auto request = [&](int cs0) __attribute__((always_inline)) {
        const int nsub = min(FWD_SUB, nch - cs0);
        const float *src = fcol + (long)(cs0 + min(sc, nsub - 1)) * plane;
#pragma unroll
        for (int yy = 0; yy < 32; ++yy) stg[yy] = src[(long)min(yy, bh - 1) * W];
    };
    request(0);
    for (int cs0 = 0; cs0 < nch; cs0 += FWD_SUB) {
        const int nsub = min(FWD_SUB, nch - cs0);
        if (sc < nsub && l32 < bw) {
#pragma unroll
            for (int yy = 0; yy < 32; ++yy)
                if (yy < bh) s_f[sc][yy * bw + l32] = stg[yy];
        }
        __syncthreads();
        if (cs0 + FWD_SUB < nch) request(cs0 + FWD_SUB);
        if (live) {
#pragma unroll
            for (int c = 0; c < FWD_SUB; ++c) {
                if (c < nsub) {
                    float v;
                    if (boxed) {
                        v = s_f[c][o00] * sp.w00 + s_f[c][o01] * sp.w01 + s_f[c][o10] * sp.w10 + s_f[c][o11] * sp.w11;
                    } else {
                        const float *p = f + (long)(cs0 + c) * plane;
                        v = p[sp.cy0 * W + sp.cx0] * sp.w00 + p[sp.cy0 * W + sp.cx1] * sp.w01 + p[sp.cy1 * W + sp.cx0] * sp.w10 + p[sp.cy1 * W + sp.cx1] * sp.w11;
                    }
                    o_[(long)(cs0 + c) * cc] = v;
                }
            }
        }
        __syncthreads();
    }
}

bool crop_fwd_staged_ok(int H, int W, int crop) {
    const float k = (float)crop / (float)H, step = 2.f / (float)(crop - 1);
    const float pitch_max = std::fmax(k * step * 0.5f * (float)(W - 1), k * step * 0.5f * (float)(H - 1));   // map pixels per output pixel
    return (float)(FWD_TILE - 1) * 1.41421357f * pitch_max + 5.f <= 32.f;
}

// Backward in GATHER form: one thread per pixel of the map gradient, no atomics, no memset, bit-reproducible.
//
// The scatter form (every output-pixel gradient added to its four source pixels with fp32 atomics - what torch's grid_sampler
// backward does) cost 20 ms per call at train_full's sizes: 64+ crops x 384 channels x 96 x 96 x 4 = 0.9 G atomics.  The
// sampling grid is a rotation at (almost exactly) the map's own pixel pitch, so a map pixel (sy, sx) is a bilinear corner of
// only the few output pixels whose sample position falls inside the 2 x 2 square around it: the thread inverts the affine
// map, visits the integer output positions within sqrt(2)/pitch of the pre-image, re-derives each one's corners and
// weights with EXACTLY the forward's arithmetic (so this is the transpose of the forward, not an approximation of it) and
// accumulates weight x gradient over its channels in registers.  Crops that share a map are summed in index order.
// A workgroup owns a 16 x 16 tile of the map: the output gradients it can touch lie in a (rotated) box of at most 27 x 27
// pixels, which is staged in LDS eight channels at a time with coalesced row reads, so the per-pixel gathers hit LDS and
// not 8+ L1 lines per wave instruction.
constexpr int BWD_TW = 16, BWD_TH = 16;  // map pixels per workgroup: a square tile keeps its rotated pre-image compact
constexpr int BWD_CPB = 32;              // channels per thread (accumulators in registers)
constexpr int BWD_SUB = 8;               // channels staged in LDS at a time
constexpr int BWD_CAP = 1024;            // floats per staged channel: bounding box of the tile's pre-image (<= 27 x 27 at pitch 1)
constexpr int BWD_MAXSPAN = 3;           // candidates per axis the staged path holds in registers

struct CropGeom {
    int n;
    float cs, sn, t02, t12;
};

// Weight with which output pixel (y, x) of a crop samples map pixel (sy, sx): the forward's arithmetic, verbatim (0 when
// (sy, sx) is not one of its four corners).  (sy, sx) is inside the map, so a corner that equals it is a valid corner.
__device__ __forceinline__ float corner_weight(const CropGeom &cg, float k, int x, int y, int sx, int sy, int W, int H, int crop) {
    const float xs = lin(x, crop), ys = lin(y, crop);
    const float gx = k * cg.cs * xs + (k * -cg.sn) * ys + cg.t02;
    const float gy = k * cg.sn * xs + k * cg.cs * ys + cg.t12;
    const float ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    const float iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float wx = sx == x0 ? wx0 : (sx == x0 + 1 ? wx1 : 0.f);
    const float wy = sy == y0 ? wy0 : (sy == y0 + 1 ? wy1 : 0.f);
    return wx * wy;
}

// Output-pixel coordinates (continuous) whose sample position is the centre of map pixel (sx, sy).
__device__ __forceinline__ void preimage(const CropGeom &cg, float k, float step, float sx, float sy, int W, int H, float &px, float &py) {
    const float u = (2.f * sx / (float)(W - 1) - 1.f - cg.t02) / k;
    const float v = (2.f * sy / (float)(H - 1) - 1.f - cg.t12) / k;
    px = ((cg.cs * u + cg.sn * v) + 1.f) / step;
    py = ((-cg.sn * u + cg.cs * v) + 1.f) / step;
}

__global__ __launch_bounds__(256, 3) void k_crop_rotate_bwd(const float *__restrict__ g, int n, const int *__restrict__ map_index, int C, int H,
                                                         int W, const float *__restrict__ locs, const float *__restrict__ oris, float ppm,
                                                         int crop, float ox, float oy, float *__restrict__ grad_feat, int num_maps) {
    __shared__ CropGeom s_crop[256];
    __shared__ int s_wave_cnt[4];
    __shared__ float s_g[BWD_SUB][BWD_CAP];   // the output gradients the tile can touch, BWD_SUB channels at a time
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // workgroup -> (map, channel block, tile), XCD aware like the forward kernel: the tiles of one (map, channel block) slab stage
    // overlapping boxes of the same crops' gradient planes - all of them on one XCD (slab = 8 (l / (8 tiles)) + l % 8)
    const int tiles_x = (W + BWD_TW - 1) / BWD_TW, tiles = tiles_x * ((H + BWD_TH - 1) / BWD_TH);
    const int cblocks = (C + BWD_CPB - 1) / BWD_CPB;
    const int l = blockIdx.x;
    const int slab = 8 * (l / (8 * tiles)) + (l & 7), tile = (l >> 3) % tiles;
    const int m = slab / cblocks, cblk = slab - m * cblocks;
    if (m >= num_maps) return;     // (the launch is padded to whole groups of eight slabs; workgroup-uniform)
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int tx0 = tile_x * BWD_TW, ty0 = tile_y * BWD_TH;
    const int sx = tx0 + (tid & (BWD_TW - 1)), sy = ty0 + tid / BWD_TW;
    const int c_lo = cblk * BWD_CPB;
    const int nch = min(BWD_CPB, C - c_lo);
    const bool inside_map = sx < W && sy < H;
    const float k = (float)crop / (float)H;
    const float step = 2.f / (float)(crop - 1);
    // output pixels per map pixel along each axis of the (rotated) grid; the pre-image of the 2 x 2 square around a map pixel
    // lies within `reach` output pixels of the pre-image of its centre
    const float pitch_min = fminf(k * step * 0.5f * (float)(W - 1), k * step * 0.5f * (float)(H - 1));
    const float reach = 1.41421357f / pitch_min + 0.01f;
    const int span = (int)floorf(2.f * reach) + 1;
    const long cc = (long)crop * crop;
    float acc[BWD_CPB];
#pragma unroll
    for (int c = 0; c < BWD_CPB; ++c) acc[c] = 0.f;

    for (int base = 0; base < n; base += 256) {
        // the crops of this pass that sample map m, in index order (ordered compaction: the sum below must not depend on timing)
        const int i = base + tid;
        const bool mine = i < n && map_index[i] == m;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) s_wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wid) pos += s_wave_cnt[w];
            total += s_wave_cnt[w];
        }
        if (mine) {
            const float o = oris[i];
            const float cs = cosf(o), sn = sinf(o);
            const float rx = locs[i * 2 + 0] * ppm / ((float)H / 2.f);
            const float ry = locs[i * 2 + 1] * ppm / ((float)W / 2.f);
            s_crop[pos] = CropGeom{i, cs, sn, -k * ox * cs + k * oy * sn + ox + rx, -k * ox * sn - k * oy * cs + oy + ry};
        }
        __syncthreads();
        for (int q = 0; q < total; ++q) {
            const CropGeom cg = s_crop[q];
            // bounding box (in output pixels) of everything the tile can be a corner of: the map is affine, so the extremes
            // of the pre-image are at the tile's corners.  Workgroup-uniform.
            float cx[4], cy[4];
            preimage(cg, k, step, (float)tx0, (float)ty0, W, H, cx[0], cy[0]);
            preimage(cg, k, step, (float)(tx0 + BWD_TW - 1), (float)ty0, W, H, cx[1], cy[1]);
            preimage(cg, k, step, (float)tx0, (float)(ty0 + BWD_TH - 1), W, H, cx[2], cy[2]);
            preimage(cg, k, step, (float)(tx0 + BWD_TW - 1), (float)(ty0 + BWD_TH - 1), W, H, cx[3], cy[3]);
            const int bx0 = max((int)ceilf(fminf(fminf(cx[0], cx[1]), fminf(cx[2], cx[3])) - reach) - 1, 0);
            const int bx1 = min((int)floorf(fmaxf(fmaxf(cx[0], cx[1]), fmaxf(cx[2], cx[3])) + reach) + 1, crop - 1);
            const int by0 = max((int)ceilf(fminf(fminf(cy[0], cy[1]), fminf(cy[2], cy[3])) - reach) - 1, 0);
            const int by1 = min((int)floorf(fmaxf(fmaxf(cy[0], cy[1]), fmaxf(cy[2], cy[3])) + reach) + 1, crop - 1);
            if (bx0 > bx1 || by0 > by1) continue;   // the crop does not reach this tile
            const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            // this thread's candidates
            float px, py;
            preimage(cg, k, step, (float)sx, (float)sy, W, H, px, py);
            const int xlo = max((int)ceilf(px - reach), 0), xhi = min((int)floorf(px + reach), crop - 1);
            const int ylo = max((int)ceilf(py - reach), 0), yhi = min((int)floorf(py + reach), crop - 1);
            const bool any_here = inside_map && xlo <= xhi && ylo <= yhi;
            const float *gq = g + ((long)cg.n * C + c_lo) * cc;
            if (bw * bh > BWD_CAP || bw > 32) __builtin_trap();   // the host only picks this kernel when the box always fits (crop_bwd_staged_ok)
            {
                // staged path: weights and LDS offsets of the (up to) 3 x 3 candidates in registers
                float wgt[BWD_MAXSPAN * BWD_MAXSPAN];
                int off[BWD_MAXSPAN * BWD_MAXSPAN];
#pragma unroll
                for (int dy = 0; dy < BWD_MAXSPAN; ++dy)
#pragma unroll
                    for (int dx = 0; dx < BWD_MAXSPAN; ++dx) {
                        const int x = xlo + dx, y = ylo + dy;
                        const bool live = any_here && dx < span && dy < span && x <= xhi && y <= yhi;
                        const float w_ = live ? corner_weight(cg, k, x, y, sx, sy, W, H, crop) : 0.f;
                        wgt[dy * BWD_MAXSPAN + dx] = w_;
                        // a candidate with a weight lies inside the box by construction; the others read element 0
                        off[dy * BWD_MAXSPAN + dx] = w_ != 0.f ? (y - by0) * bw + (x - bx0) : 0;
                        __builtin_amdgcn_sched_barrier(0);   // one candidate at a time: nine interleaved copies of the weight arithmetic cost 100+ registers
                    }
                // Staging (round 6, as in the forward kernel): thread (channel tid / 32, column tid % 32) fetches its column of the
                // box - at most 32 rows, all requested before the first is used, from clamped addresses - and the next eight channels
                // are requested before this group's products are accumulated.  The first version walked the box in seven dependent
                // batches of four loads.
                if (bh > 32) __builtin_trap();
                const int l32 = tid & 31, sc = tid >> 5;
                const float *gcol = gq + (long)by0 * crop + bx0 + min(l32, bw - 1);
                float stg[32];
                auto request = [&](int cs0) __attribute__((always_inline)) {
                    const int nsub = min(BWD_SUB, nch - cs0);
                    const float *src = gcol + (long)(cs0 + min(sc, nsub - 1)) * cc;
#pragma unroll
                    for (int yy = 0; yy < 32; ++yy) stg[yy] = src[(long)min(yy, bh - 1) * crop];
                };
                request(0);
#pragma unroll
                for (int cs0 = 0; cs0 < BWD_CPB; cs0 += BWD_SUB) {
                    if (cs0 < nch) {   // workgroup-uniform
                        if (sc < min(BWD_SUB, nch - cs0) && l32 < bw) {
#pragma unroll
                            for (int yy = 0; yy < 32; ++yy)
                                if (yy < bh) s_g[sc][yy * bw + l32] = stg[yy];
                        }
                        __syncthreads();
                        if (cs0 + BWD_SUB < nch) request(cs0 + BWD_SUB);
#pragma unroll
                        for (int c = 0; c < BWD_SUB; ++c) {
#pragma unroll
                            for (int j = 0; j < BWD_MAXSPAN * BWD_MAXSPAN; ++j) acc[cs0 + c] = fmaf(wgt[j], s_g[c][off[j]], acc[cs0 + c]);
                            if (c & 1) __builtin_amdgcn_sched_barrier(0);   // 18 LDS reads in flight, not all 72 (registers)
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
    }
    if (inside_map) {
        float *o = grad_feat + ((long)m * C + c_lo) * H * W + (long)sy * W + sx;
#pragma unroll
        for (int c = 0; c < BWD_CPB; ++c)
            if (c_lo + c < C) o[(long)c * H * W] = acc[c];
    }
}

// The same gradient without the LDS stage, for geometries whose boxes do not fit it (a map much larger or smaller than the
// crop: pitch far from 1): 32 x 8 pixel tiles, candidates gathered from L2.
__global__ __launch_bounds__(256) void k_crop_rotate_bwd_general(const float *__restrict__ g, int n, const int *__restrict__ map_index, int C,
                                                                 int H, int W, const float *__restrict__ locs, const float *__restrict__ oris,
                                                                 float ppm, int crop, float ox, float oy, float *__restrict__ grad_feat) {
    __shared__ CropGeom s_crop[256];
    __shared__ int s_wave_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_x = (W + 31) / 32;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int sx = tile_x * 32 + (tid & 31), sy = tile_y * 8 + (tid >> 5);
    const int m = blockIdx.z;
    const int c_lo = blockIdx.y * BWD_CPB;
    const int nch = min(BWD_CPB, C - c_lo);
    const bool inside_map = sx < W && sy < H;
    const float k = (float)crop / (float)H;
    const float step = 2.f / (float)(crop - 1);
    const float pitch_min = fminf(k * step * 0.5f * (float)(W - 1), k * step * 0.5f * (float)(H - 1));
    const float reach = 1.41421357f / pitch_min + 0.01f;
    const int span = (int)floorf(2.f * reach) + 1;
    const long cc = (long)crop * crop;
    float acc[BWD_CPB];
#pragma unroll
    for (int c = 0; c < BWD_CPB; ++c) acc[c] = 0.f;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const bool mine = i < n && map_index[i] == m;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) s_wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wid) pos += s_wave_cnt[w];
            total += s_wave_cnt[w];
        }
        if (mine) {
            const float o = oris[i];
            const float cs = cosf(o), sn = sinf(o);
            const float rx = locs[i * 2 + 0] * ppm / ((float)H / 2.f);
            const float ry = locs[i * 2 + 1] * ppm / ((float)W / 2.f);
            s_crop[pos] = CropGeom{i, cs, sn, -k * ox * cs + k * oy * sn + ox + rx, -k * ox * sn - k * oy * cs + oy + ry};
        }
        __syncthreads();
        for (int q = 0; q < total; ++q) {
            const CropGeom cg = s_crop[q];
            float px, py;
            preimage(cg, k, step, (float)sx, (float)sy, W, H, px, py);
            const int xlo = max((int)ceilf(px - reach), 0), xhi = min((int)floorf(px + reach), crop - 1);
            const int ylo = max((int)ceilf(py - reach), 0), yhi = min((int)floorf(py + reach), crop - 1);
            const bool any_here = inside_map && xlo <= xhi && ylo <= yhi;
            if (!__any(any_here)) continue;
            const float *gq = g + ((long)cg.n * C + c_lo) * cc;
            for (int dy = 0; dy < span; ++dy)
                for (int dx = 0; dx < span; ++dx) {
                    const int x = xlo + dx, y = ylo + dy;
                    const float w_ = (any_here && x <= xhi && y <= yhi) ? corner_weight(cg, k, x, y, sx, sy, W, H, crop) : 0.f;
                    if (w_ != 0.f) {
                        const float *gp = gq + (long)y * crop + x;
                        if (nch == BWD_CPB) {
#pragma unroll
                            for (int c = 0; c < BWD_CPB; ++c) acc[c] = fmaf(w_, gp[c * cc], acc[c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < BWD_CPB; ++c)
                                if (c < nch) acc[c] = fmaf(w_, gp[c * cc], acc[c]);
                        }
                    }
                }
        }
        __syncthreads();
    }
    if (inside_map) {
        float *o = grad_feat + ((long)m * C + c_lo) * H * W + (long)sy * W + sx;
#pragma unroll
        for (int c = 0; c < BWD_CPB; ++c)
            if (c_lo + c < C) o[(long)c * H * W] = acc[c];
    }
}

// Host side of the choice: the staged kernel holds 3 x 3 candidates per pixel and a box of at most 32 x 32 output pixels per tile.
bool crop_bwd_staged_ok(int H, int W, int crop) {
    const float k = (float)crop / (float)H, step = 2.f / (float)(crop - 1);
    const float pitch_min = std::fmin(k * step * 0.5f * (float)(W - 1), k * step * 0.5f * (float)(H - 1));
    const float reach = 1.41421357f / pitch_min + 0.01f;
    const int span = (int)std::floor(2.f * reach) + 1;
    const float box = (float)(BWD_TW - 1) * 1.41421357f / pitch_min + 2.f * reach + 3.f;   // upper bound of a tile's box side
    return span <= BWD_MAXSPAN && box <= 32.f;
}
}  // namespace

namespace {
int crop_launch(const float *feat, int nmaps, const int *map_index, int C, int H, int W, const float *locs, const float *oris, int n,
                float ppm, int crop, float ox, float oy, float *out, hipStream_t st) {
    const int c_per_block = 32;
    const int tok = timer_begin("crop_rotate", st);
    if (crop_fwd_staged_ok(H, W, crop) && !getenv("LAV_CROP_FWD_GENERAL")) {   // (A/B knob)
        const int tiles = (crop + FWD_TILE - 1) / FWD_TILE;
        // channels per workgroup: 32; 16 for a single crop (432 workgroups of 32 channels are 1.7 per CU, each a serial chain of
        // stage -> sample -> store groups of eight channels: 19.7 vs 20.7 us; from two crops on 32 wins, 8 always loses - measured,
        // profiles/r06_crop_probe.txt)
        static const int cpb_env = [] { const char *e = getenv("LAV_CROP_CPB"); return e ? atoi(e) : 0; }();   // (A/B knob: 8, 16 or 32)
        const int cpb = cpb_env == 8 || cpb_env == 16 || cpb_env == 32 ? cpb_env : (n <= 1 ? 16 : FWD_CPB);
        const long slabs = (long)n * ((C + cpb - 1) / cpb), groups = (slabs + 7) / 8;
        hipLaunchKernelGGL(k_crop_rotate_staged, dim3((unsigned)(groups * 8 * tiles * tiles)), dim3(256), 0, st, feat, nmaps, map_index, C,
                           H, W, locs, oris, ppm, crop, ox, oy, out, lav::batch_limit(), cpb, n);
    } else {
        dim3 grid((crop * crop + 255) / 256, (C + c_per_block - 1) / c_per_block, n);
        hipLaunchKernelGGL(k_crop_rotate, grid, dim3(256), 0, st, feat, nmaps, map_index, C, H, W, locs, oris, ppm, crop, ox, oy, c_per_block,
                           out, lav::batch_limit());
    }
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
    return crop_launch(feat, feat_batch, nullptr, C, H, W, locs, oris, n, pixels_per_meter, crop, offset_x, offset_y, out,
                       static_cast<hipStream_t>(stream));
}

extern "C" int lav_crop_rotate_indexed(const float *feat, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                                       const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                                       float *out, void *stream) {
    LAV_REQUIRE(n >= 0 && num_maps >= 1 && C > 0 && H > 1 && W > 1 && crop > 1, "lav_crop_rotate_indexed: bad sizes");
    if (n == 0) return LAV_OK;
    LAV_REQUIRE(feat && map_index && locs && oris && out, "lav_crop_rotate_indexed: null argument");
    return crop_launch(feat, num_maps, map_index, C, H, W, locs, oris, n, pixels_per_meter, crop, offset_x, offset_y, out,
                       static_cast<hipStream_t>(stream));
}

extern "C" int lav_crop_rotate_backward(const float *grad_out, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                                        const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                                        float *grad_feat, void *stream) {
    LAV_REQUIRE(n >= 0 && num_maps >= 1 && C > 0 && H > 1 && W > 1 && crop > 1, "lav_crop_rotate_backward: bad sizes");
    LAV_REQUIRE(grad_feat, "lav_crop_rotate_backward: null argument");
    LAV_REQUIRE(num_maps <= 65535, "lav_crop_rotate_backward: too many maps");
    LAV_REQUIRE(n == 0 || (grad_out && map_index && locs && oris), "lav_crop_rotate_backward: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // every pixel of grad_feat is written (zeros where no crop samples it): no memset, no atomics
    const bool staged = crop_bwd_staged_ok(H, W, crop) && !getenv("LAV_CROP_BWD_GENERAL");   // (A/B knob)
    const int tw = staged ? BWD_TW : 32, th = staged ? BWD_TH : 8;
    dim3 grid(((W + tw - 1) / tw) * ((H + th - 1) / th), (C + BWD_CPB - 1) / BWD_CPB, num_maps);
    const int tok = timer_begin("crop_rotate_backward", st);
    if (staged) {
        const long slabs = (long)num_maps * grid.y, groups = (slabs + 7) / 8;
        hipLaunchKernelGGL(k_crop_rotate_bwd, dim3((unsigned)(groups * 8 * grid.x)), dim3(256), 0, st, grad_out, n, map_index, C, H, W, locs, oris,
                           pixels_per_meter, crop, offset_x, offset_y, grad_feat, num_maps);
    }
    else
        hipLaunchKernelGGL(k_crop_rotate_bwd_general, grid, dim3(256), 0, st, grad_out, n, map_index, C, H, W, locs, oris, pixels_per_meter,
                           crop, offset_x, offset_y, grad_feat);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

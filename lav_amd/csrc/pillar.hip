// PointPillars dynamic voxelisation -> decoration -> PointNet -> scatter-max -> dense BEV canvas.
//
// Replaces PointPillarNet.forward of the reference (lav/models/point_pillar.py:92-116) including the two
// torch_scatter calls (:33, :62) and coords.unique(dim=0) (:82).  gfx950 only.
//
// Data flow (all buffers in HBM, sizes for the v2 agent: N ~ 196k points x 11 floats, 320x320 cells, C = 64):
//
//   k_key_count   1 thread / point   reads x,y                -> key[i] (cell id or -1), slot[i] = arrival
//                                                                 number inside its cell (int atomic, order free)
//   scan          exclusive prefix sum of the per-cell counts -> cell_offset[cells+1]   (counting sort)
//   k_place       1 thread / point                            -> sorted_idx[cell_offset[key]+slot] = i
//   k_pointnet_scatter  one workgroup per (cloud, canvas row, column tile):
//        the cells of one canvas row are contiguous in key order, hence so are their points in sorted_idx;
//        the workgroup  (a) sums xyz per cell in LDS with 64-bit fixed-point atomics (order independent =>
//        run-to-run deterministic; exact to 2^-32 m),  (b) decorates each point (16 features) and runs the
//        2-layer PointNet,  (c) max-reduces into an LDS tile [C][tile_w] (values >= 0 after ReLU, so an
//        unsigned integer max on the float bits is exact),  (d) streams the tile - zeros for empty cells
//        included - to the NCHW canvas with 16-byte coalesced stores.  The canvas is written exactly once
//        and never read or memset: algorithmic traffic 4*(N*D + C*ny*nx) bytes.
//
// Index outputs (unique_coords / inverse, the "bit-exact pillar indices" of the parity contract) are produced
// by extra scans only when requested.
//
// Cell-id arithmetic is float32 exactly as the reference's: (x - min_x) * ppm, subtraction and multiplication
// rounded separately, then truncation.  Because the product can round up to exactly nx (resp. ny) - e.g.
// y = nextafter(40,0) with min_y=-40 gives yi = 320 - the key space has one extra row and column; the canvas
// write clamps them like the reference (:89) with "later pillar in unique order wins".
#include <cstdlib>

#include "common.hpp"

// parity-critical float32 arithmetic: no fused multiply-add contraction anywhere in this file
// (HIP's __fadd_rn/__fmul_rn are plain operators that clang would otherwise fuse)
#pragma clang fp contract(off)

namespace {
using namespace lav;

constexpr int C = 64;            // PointNet width (config num_features [64,64])
constexpr int MAX_BATCH = 64;    // per-call limit on clouds (kernel-argument table)
constexpr int TILE_MAX_W = 176;  // canvas columns per workgroup tile (LDS tile [64][tile_w|1] floats)
constexpr double FIX_SCALE = 4294967296.0;  // 2^32 fixed-point scale of the per-cell coordinate sums

struct PillarArgs {
    const float *points;
    int batch, max_points, D;
    int n[MAX_BATCH];
    float min_x, max_x, min_y, max_y, ppm;
    int nx, ny;  // nx = number of xi cells = canvas columns count; ny = number of yi cells = canvas rows count
    int KX, KY;  // key space (nx+1) x (ny+1)
};

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_key_count(PillarArgs a, int *__restrict__ key, int *__restrict__ slot,
                                                   int *__restrict__ cell_count) {
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / a.max_points);
    const int i = (int)(gid - (long)b * a.max_points);
    int k = -1;
    if (i < a.n[b]) {
        const float *pt = a.points + gid * a.D;
        const float x = pt[0], y = pt[1];
        // NaN fails every comparison and is dropped, as in torch
        if (x >= a.min_x && x < a.max_x && y >= a.min_y && y < a.max_y) {
            const float fx = __fmul_rn(__fsub_rn(x, a.min_x), a.ppm);
            const float fy = __fmul_rn(__fsub_rn(y, a.min_y), a.ppm);
            const int xi = (int)fx, yi = (int)fy;  // truncation; both are in [0, nx] x [0, ny]
            k = (b * a.KX + xi) * a.KY + yi;
        }
    }
    key[gid] = k;
    if (k >= 0) slot[gid] = atomicAdd(&cell_count[k], 1);
}

__global__ __launch_bounds__(256) void k_place(long total, const int *__restrict__ key, const int *__restrict__ slot,
                                               const int *__restrict__ cell_offset, int *__restrict__ sorted_idx) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k >= 0) sorted_idx[cell_offset[k] + slot[gid]] = (int)gid;
}

// ---------------------------------------------------------------------------------------------------------
// Exclusive scan of f(in[i]) over n ints, 3 launches (block sums / scan of sums / apply). out has n+1 entries.
// mode 0: f = v;  mode 1: f = (v > 0);  mode 2: f = (v >= 0)
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ int scan_f(int v, int mode) { return mode == 0 ? v : (mode == 1 ? (v > 0) : (v >= 0)); }

__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int &total) {
    // 256 threads: wave-level inclusive scan via shuffles, then across the 4 waves through LDS
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        int s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(const int *__restrict__ in, long n, int mode,
                                                          int *__restrict__ block_sums) {
    __shared__ int lds[SCAN_BLOCK / 64];
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < n) s += scan_f(in[base + j], mode);
    int tot;
    block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocks(int *__restrict__ block_sums, int nblocks,
                                                            int *__restrict__ grand_total) {
    __shared__ int lds[SCAN_BLOCK / 64];
    int carry = 0;
    for (int base = 0; base < nblocks; base += SCAN_BLOCK) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_sums[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, lds, tot);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(const int *__restrict__ in, long n, int mode,
                                                           const int *__restrict__ block_sums, int *__restrict__ out) {
    __shared__ int lds[SCAN_BLOCK / 64];
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = base + j < n ? scan_f(in[base + j], mode) : 0;
        s += v[j];
    }
    int tot;
    int ex = block_exclusive_scan(s, lds, tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
    // out[n] = grand total: written by the thread that owns element n-1
    if (base <= n - 1 && n - 1 < base + SCAN_ITEMS) out[n] = ex;
}

int exclusive_scan(const int *in, long n, int mode, int *out, int *block_sums, int *grand_total, hipStream_t st) {
    const int nblocks = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_sums, dim3(nblocks), dim3(SCAN_BLOCK), 0, st, in, n, mode, block_sums);
    LAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_BLOCK), 0, st, block_sums, nblocks, grand_total);
    LAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(SCAN_BLOCK), 0, st, in, n, mode, block_sums, out);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

// ---------------------------------------------------------------------------------------------------------
// PointNet + scatter-max + canvas tile.
//
// Workgroup = 4 waves = one (cloud, canvas row, column tile).  Per "layer" (see the override rules above):
//   (a) per-cell xyz sums in LDS (64-bit fixed point, order independent), means
//   (b) every wave takes passes of 32 sorted points and runs BOTH PointNet layers on the matrix cores with all
//       activations in registers (v_mfma_f32_32x32x2_f32, exact fp32):
//         layer 1 (transposed)  D1[c][p]  = sum_k W1[k][c] * F[k][p]     A = weights, B = point features
//             lane l supplies feature 2s+(l>>5) of point l&31 at k-step s; the bias rides as feature 16 (=1).
//             D1 leaves lane (p, half) holding channels c = 32*mt + (r&3) + 8*(r>>2) + 4*half  (r = 0..15)
//         layer 2               D2[p][c2] = sum_c H1[p][c] * W2[c][c2]   A = relu(D1) AS IT SITS, B = weights
//             k-step (mt, r) uses k = 32*mt + (r&3) + 8*(r>>2) + 4*half - a permutation of 0..63, which a sum
//             does not care about - so no lane shuffles or LDS round trip between the layers.
//             D2 leaves lane (c2, half) holding 16 points.  Sorted point j of the pass is given to MFMA row
//             p(j) = (j&3) + 8*((j&15)>>2) + 4*(j>>4), so those 16 points are CONSECUTIVE sorted points and
//             equal-cell runs are folded in registers before one LDS max per run.
//   (c) integer max of the float bits into the LDS tile [C][tile_w|1] (odd stride: conflict-free)
//   (d) the tile - zeros included - streams to the NCHW canvas, 256 B per wave-instruction.
struct TileGeo {
    int tiles_per_row;  // T
    int TW;             // columns per tile
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// shared prologue of a layer: zero sums, accumulate, means, override-zeroing.  Returns with a barrier done.
template <int D>
__device__ __forceinline__ void layer_means(const PillarArgs &a, const int *__restrict__ key, const int *__restrict__ sorted_idx,
                                            const int *__restrict__ cell_offset, int cell0, int ncell, int p0, int p1,
                                            int yi0, int c0, bool overriding, int TWP, float *tile,
                                            unsigned long long *sums, float *means, int *occupied) {
    const int tid = threadIdx.x;
    __syncthreads();
    for (int j = tid; j < ncell * 3; j += 256) sums[j] = 0ull;
    __syncthreads();
    for (int j = p0 + tid; j < p1; j += 256) {
        const int idx = sorted_idx[j];
        const int cell = key[idx] - cell0;
        const float *pt = a.points + (long)idx * D;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const long long q = __double2ll_rn((double)pt[d] * FIX_SCALE);
            atomicAdd(&sums[cell * 3 + d], (unsigned long long)q);
        }
    }
    __syncthreads();
    for (int j = tid; j < ncell; j += 256) {
        const int cnt = cell_offset[cell0 + j + 1] - cell_offset[cell0 + j];
        occupied[j] = cnt > 0;
        if (cnt > 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
                means[j * 3 + d] = (float)((double)(long long)sums[j * 3 + d] / ((double)cnt * FIX_SCALE));
        }
    }
    __syncthreads();
    if (overriding) {  // a later pillar replaces whatever an earlier one put on the same canvas cell
        for (int i = tid; i < C * ncell; i += 256) {
            const int ch = i / ncell, j = i - ch * ncell;
            const int col = min(yi0 + j, a.nx - 1) - c0;
            if (occupied[j]) tile[ch * TWP + col] = 0.f;
        }
        __syncthreads();
    }
}

template <int D>
__device__ __forceinline__ void decorate(const PillarArgs &a, const float *pt, const float *mean3, int xi, int yi, float *f) {
#pragma unroll
    for (int d = 0; d < D; ++d) f[d] = pt[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) f[D + d] = __fsub_rn(f[d], mean3[d]);
    // reference decorate(): x - (yi/ppm + min_x), y - (xi/ppm + min_y)  (sic: swapped, un-centred; :57-58)
    f[D + 3] = __fsub_rn(f[0], __fadd_rn(__fdiv_rn((float)yi, a.ppm), a.min_x));
    f[D + 4] = __fsub_rn(f[1], __fadd_rn(__fdiv_rn((float)xi, a.ppm), a.min_y));
}

template <int D, bool USE_MFMA>
__global__ __launch_bounds__(256, 2) void k_pointnet_scatter(PillarArgs a, TileGeo tg, const int *__restrict__ key,
                                                          const int *__restrict__ sorted_idx,
                                                          const int *__restrict__ cell_offset,
                                                          const float *__restrict__ w1, const float *__restrict__ b1,
                                                          const float *__restrict__ w2, const float *__restrict__ b2,
                                                          float *__restrict__ canvas) {
    constexpr int K1 = D + 5;             // decorated features
    constexpr int KS1 = (K1 + 2) / 2;     // layer-1 k-steps incl. the bias feature (K1=16 -> 9)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TW = tg.TW, TWP = TW | 1;
    float *tile = reinterpret_cast<float *>(smem);                                                     // [C][TWP]
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(smem + ((C * TWP * 4 + 15) & ~15));  // [TW][3]
    float *means = reinterpret_cast<float *>(sums + TW * 3);                                           // [TW][3]
    int *occupied = reinterpret_cast<int *>(means + TW * 3);                                           // [TW]
    float *w2s = reinterpret_cast<float *>(occupied + ((TW + 3) & ~3));                                // [C][C] layer-2 weights

    int wg = blockIdx.x;
    const int t = wg % tg.tiles_per_row;
    wg /= tg.tiles_per_row;
    const int r = wg % a.ny;  // canvas row
    const int b = wg / a.ny;
    const int c0 = t * TW;
    const int c1 = min(a.nx, c0 + TW);
    const int tw = c1 - c0;  // live columns in this tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;

    for (int i = tid; i < C * TWP; i += 256) tile[i] = 0.f;

    // weights in registers (MFMA path), loaded lazily by waves that have work
    float a1[2][KS1];   // layer-1 A operand: W1[2s+half][32*mt + l31], bias as k = K1
    float b2v[2];
    bool weights_loaded = false, w2_staged = false;

    // key rows that land on canvas row r (reference clamp semantics, point_pillar.py:89)
    int xi_lo, xi_hi;
    if (r > 0) {
        xi_lo = xi_hi = a.ny - 1 - r;
    } else {
        xi_lo = max(a.ny - 1, 0);
        xi_hi = a.nx;
    }
    for (int xi = xi_lo; xi <= min(xi_hi, a.nx); ++xi) {
        // layers: first the cells that map 1:1 onto the tile's columns, then (last tile only) the overflow
        // cells yi in [nx, ny] which all clamp onto column nx-1
        const int yi_direct_hi = min(c1 - 1, a.ny);
        const int n_over = (c1 == a.nx) ? max(0, a.ny - a.nx + 1) : 0;
        for (int layer = 0; layer <= n_over; ++layer) {
            int yi0, ncell;
            if (layer == 0) {
                yi0 = c0;
                ncell = yi_direct_hi - c0 + 1;
            } else {
                yi0 = a.nx + layer - 1;
                ncell = 1;
            }
            if (ncell <= 0) continue;
            const int cell0 = (b * a.KX + xi) * a.KY + yi0;
            const int p0 = cell_offset[cell0], p1 = cell_offset[cell0 + ncell];
            if (p0 == p1) continue;  // workgroup-uniform
            const bool overriding = (layer > 0) || (xi > xi_lo);
            if (USE_MFMA && !w2_staged) {  // layer-2 weights -> LDS once per workgroup (B operand of every pass)
                w2_staged = true;
                for (int i = tid; i < C * C; i += 256) w2s[i] = w2[i];
            }
            layer_means<D>(a, key, sorted_idx, cell_offset, cell0, ncell, p0, p1, yi0, c0, overriding, TWP, tile, sums, means, occupied);

            if constexpr (!USE_MFMA) {
                // cross-check path: one point per thread, plain fp32 FMAs (slow; selected by LAV_PILLAR_IMPL=valu)
                for (int j = p0 + tid; j < p1; j += 256) {
                    const int idx = sorted_idx[j];
                    const int cell = key[idx] - cell0;
                    const int yi = yi0 + cell;
                    float f[K1];
                    decorate<D>(a, a.points + (long)idx * D, means + cell * 3, xi, yi, f);
                    float h1[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float acc = b1[c];
#pragma unroll
                        for (int k = 0; k < K1; ++k) acc = fmaf(f[k], w1[k * C + c], acc);
                        h1[c] = acc > 0.f ? acc : 0.f;
                    }
                    const int col = min(yi, a.nx - 1) - c0;
                    for (int c = 0; c < C; ++c) {
                        float acc = b2[c];
#pragma unroll
                        for (int k = 0; k < C; ++k) acc = fmaf(h1[k], w2[k * C + c], acc);
                        const float v = acc > 0.f ? acc : 0.f;
                        atomicMax(reinterpret_cast<unsigned *>(&tile[c * TWP + col]), __float_as_uint(v));
                    }
                }
            } else {
                const int npass = (p1 - p0 + 31) >> 5;
                if (wid < npass && !weights_loaded) {
                    weights_loaded = true;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int s = 0; s < KS1; ++s) {
                            const int k = 2 * s + half;
                            const int c = 32 * mt + l31;
                            a1[mt][s] = k < K1 ? w1[k * C + c] : (k == K1 ? b1[c] : 0.f);
                        }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) b2v[nt] = b2[32 * nt + l31];
                }
                // MFMA row of this lane's point -> position inside the pass's 32 sorted points
                const int sp = 16 * ((l31 >> 2) & 1) + (l31 & 3) + 4 * (l31 >> 3);
                for (int pass = wid; pass < npass; pass += 4) {
                    const int j = p0 + pass * 32 + sp;
                    const bool live = j < p1;
                    const int idx = live ? sorted_idx[j] : 0;
                    const int cell = live ? key[idx] - cell0 : 0;
                    const int yi = yi0 + cell;
                    const int mycol = live ? min(yi, a.nx - 1) - c0 : -1;
                    float f[K1];
                    decorate<D>(a, a.points + (long)idx * D, means + cell * 3, xi, yi, f);
                    float fe[KS1];
#pragma unroll
                    for (int s = 0; s < KS1; ++s) {
                        const float ev = 2 * s < K1 ? f[2 * s < K1 ? 2 * s : 0] : (2 * s == K1 ? 1.f : 0.f);
                        const float od = 2 * s + 1 < K1 ? f[2 * s + 1 < K1 ? 2 * s + 1 : 0] : (2 * s + 1 == K1 ? 1.f : 0.f);
                        fe[s] = live ? (half ? od : ev) : 0.f;
                    }
                    // layer 1
                    f32x16 d1[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) d1[mt][q] = 0.f;
#pragma unroll
                        for (int s = 0; s < KS1; ++s) d1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt][s], fe[s], d1[mt], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 16; ++q) d1[mt][q] = d1[mt][q] > 0.f ? d1[mt][q] : 0.f;
                    }
                    // the 16 points this lane will hold after layer 2 are sorted positions 16*half + q;
                    // their tile columns come from the lanes that loaded them (MFMA row = (q&3)+8*(q>>2)+4*half)
                    int cols[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) cols[q] = __shfl(mycol, (q & 3) + 8 * (q >> 2) + 4 * half, 64);
                    // layer 2 + run-folded max
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        f32x16 d2;
#pragma unroll
                        for (int q = 0; q < 16; ++q) d2[q] = b2v[nt];
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) {
                                // B operand: W2[k][32*nt + l31] with k = 32*mt + (rr&3) + 8*(rr>>2) + 4*half
                                const float wv = w2s[(32 * mt + (rr & 3) + 8 * (rr >> 2) + 4 * half) * C + 32 * nt + l31];
                                d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[mt][rr], wv, d2, 0, 0, 0);
                            }
                        unsigned *trow = reinterpret_cast<unsigned *>(tile + (32 * nt + l31) * TWP);
                        int cur = cols[0];
                        float best = d2[0] > 0.f ? d2[0] : 0.f;
#pragma unroll
                        for (int q = 1; q < 16; ++q) {
                            const float v = d2[q] > 0.f ? d2[q] : 0.f;
                            if (cols[q] != cur) {
                                if (cur >= 0) atomicMax(trow + cur, __float_as_uint(best));
                                cur = cols[q];
                                best = v;
                            } else {
                                best = fmaxf(best, v);
                            }
                        }
                        if (cur >= 0) atomicMax(trow + cur, __float_as_uint(best));
                    }
                }
            }
        }
    }
    __syncthreads();
    // (d) stream the tile out; canvas [B][C][ny][nx]
    float *dst = canvas + ((long)b * C * a.ny + r) * a.nx + c0;
    const long cstride = (long)a.ny * a.nx;
    for (int ch = wid; ch < C; ch += 4) {
        const float *src = tile + ch * TWP;
        float *d = dst + ch * cstride;
        for (int j = lane; j < tw; j += 64) d[j] = src[j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Index outputs
__global__ __launch_bounds__(256) void k_unique_coords(PillarArgs a, const int *__restrict__ cell_count,
                                                       const int *__restrict__ cell_rank, long ncells,
                                                       int *__restrict__ unique_coords) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncells || cell_count[k] <= 0) return;
    const int rk = cell_rank[k];
    const int yi = (int)(k % a.KY);
    const long t = k / a.KY;
    unique_coords[rk * 3 + 0] = (int)(t / a.KX);
    unique_coords[rk * 3 + 1] = (int)(t % a.KX);
    unique_coords[rk * 3 + 2] = yi;
}

__global__ __launch_bounds__(256) void k_inverse(long total, const int *__restrict__ key, const int *__restrict__ kept_rank,
                                                 const int *__restrict__ cell_rank, int *__restrict__ inverse) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k >= 0) inverse[kept_rank[gid]] = cell_rank[k];
}

__global__ void k_counts(const int *p_total, const int *kept_total, int *counts) {
    counts[0] = *p_total;
    counts[1] = *kept_total;
}

TileGeo tile_geometry(int nx) {
    TileGeo tg;
    tg.tiles_per_row = (nx + TILE_MAX_W - 1) / TILE_MAX_W;
    int tw = (nx + tg.tiles_per_row - 1) / tg.tiles_per_row;
    tg.TW = (tw + 3) / 4 * 4;
    return tg;
}

struct Workspace {
    int *cell_count, *cell_offset, *key, *slot, *sorted_idx, *block_sums, *cell_rank, *kept_rank, *totals;
};

size_t carve(Arena &ar, Workspace &w, int batch, int max_points, const lav_grid *g) {
    const size_t ncells = (size_t)batch * (g->nx + 1) * (g->ny + 1);
    const size_t total = (size_t)batch * max_points;
    const size_t nmax = ncells > total ? ncells : total;
    w.cell_count = ar.take<int>(ncells);
    w.cell_offset = ar.take<int>(ncells + 1);
    w.key = ar.take<int>(total);
    w.slot = ar.take<int>(total);
    w.sorted_idx = ar.take<int>(total);
    w.block_sums = ar.take<int>((nmax + SCAN_TILE - 1) / SCAN_TILE + 1);
    w.cell_rank = ar.take<int>(ncells + 1);
    w.kept_rank = ar.take<int>(total + 1);
    w.totals = ar.take<int>(4);
    return align_up(ar.used, 256);
}

bool use_valu_impl() {
    const char *e = getenv("LAV_PILLAR_IMPL");
    return e && e[0] == 'v';
}

template <int D>
int launch_pointnet(const PillarArgs &a, const TileGeo &tg, const Workspace &w, const lav_pointnet *net, float *canvas,
                    hipStream_t st) {
    const int TWP = tg.TW | 1;
    const size_t lds = (((size_t)C * TWP * 4 + 15) & ~(size_t)15) + (size_t)tg.TW * 3 * 8 + (size_t)tg.TW * 3 * 4 +
                       (size_t)((tg.TW + 3) & ~3) * 4 + (size_t)C * C * 4;
    const int grid = a.batch * a.ny * tg.tiles_per_row;
    static bool attr_set = false;
    if (!attr_set) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pointnet_scatter<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pointnet_scatter<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    const int tok = timer_begin("pointnet_scatter", st);
    if (use_valu_impl())
        hipLaunchKernelGGL((k_pointnet_scatter<D, false>), dim3(grid), dim3(256), lds, st, a, tg, w.key, w.sorted_idx,
                           w.cell_offset, net->w1, net->b1, net->w2, net->b2, canvas);
    else
        hipLaunchKernelGGL((k_pointnet_scatter<D, true>), dim3(grid), dim3(256), lds, st, a, tg, w.key, w.sorted_idx,
                           w.cell_offset, net->w1, net->b1, net->w2, net->b2, canvas);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

}  // namespace

extern "C" size_t lav_pillar_workspace_bytes(int batch, int max_points, const lav_grid *grid) {
    if (!grid || batch <= 0 || max_points < 0) return 0;
    Arena ar(nullptr, 0);
    Workspace w;
    return carve(ar, w, batch, max_points, grid);
}

extern "C" int lav_pillar_scatter(const float *points, const int *h_num_points, int batch, int max_points, int D,
                                  const lav_grid *grid, const lav_pointnet *net, float *canvas, int *unique_coords,
                                  int *inverse, int *counts, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(grid && net && canvas && h_num_points, "lav_pillar_scatter: null argument");
    LAV_REQUIRE(batch >= 1 && batch <= MAX_BATCH, "lav_pillar_scatter: batch %d outside [1,%d]", batch, MAX_BATCH);
    LAV_REQUIRE(max_points >= 0 && (points || max_points == 0), "lav_pillar_scatter: bad points");
    LAV_REQUIRE(net->channels == C, "lav_pillar_scatter: PointNet width %d unsupported (built for %d)", net->channels, C);
    LAV_REQUIRE(net->num_input == D + 5, "lav_pillar_scatter: num_input %d != D+5 (D=%d)", net->num_input, D);
    LAV_REQUIRE(grid->nx > 0 && grid->ny > 0, "lav_pillar_scatter: empty grid");
    LAV_REQUIRE((long)batch * (grid->nx + 1) * (grid->ny + 1) < (1l << 30) && (long)batch * max_points < (1l << 30),
                "lav_pillar_scatter: problem too large for 32-bit indices");
    hipStream_t st = static_cast<hipStream_t>(stream);

    Arena ar(workspace, workspace_bytes);
    Workspace w;
    carve(ar, w, batch, max_points, grid);
    if (!workspace || !ar.ok()) return fail(LAV_EWORKSPACE, "lav_pillar_scatter: workspace %zu < %zu bytes", workspace_bytes, ar.used);

    PillarArgs a;
    a.points = points;
    a.batch = batch;
    a.max_points = max_points;
    a.D = D;
    for (int b = 0; b < batch; ++b) {
        LAV_REQUIRE(h_num_points[b] >= 0, "lav_pillar_scatter: negative num_points");
        a.n[b] = h_num_points[b] < max_points ? h_num_points[b] : max_points;
    }
    for (int b = batch; b < MAX_BATCH; ++b) a.n[b] = 0;
    a.min_x = grid->min_x; a.max_x = grid->max_x; a.min_y = grid->min_y; a.max_y = grid->max_y; a.ppm = grid->ppm;
    a.nx = grid->nx; a.ny = grid->ny; a.KX = grid->nx + 1; a.KY = grid->ny + 1;

    const long ncells = (long)batch * a.KX * a.KY;
    const long total = (long)batch * max_points;
    const int tok_prep = timer_begin("pillar_prep", st);
    LAV_HIP(hipMemsetAsync(w.cell_count, 0, ncells * sizeof(int), st));
    if (total > 0) {
        hipLaunchKernelGGL(k_key_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, w.key, w.slot, w.cell_count);
        LAV_LAUNCH_CHECK();
    }
    int rc = exclusive_scan(w.cell_count, ncells, 0, w.cell_offset, w.block_sums, nullptr, st);
    if (rc) return rc;
    if (total > 0) {
        hipLaunchKernelGGL(k_place, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, w.key, w.slot, w.cell_offset, w.sorted_idx);
        LAV_LAUNCH_CHECK();
    }
    timer_end(tok_prep, st);
    const TileGeo tg = tile_geometry(a.nx);
    switch (D) {
        case 11: rc = launch_pointnet<11>(a, tg, w, net, canvas, st); break;
        case 4: rc = launch_pointnet<4>(a, tg, w, net, canvas, st); break;
        case 5: rc = launch_pointnet<5>(a, tg, w, net, canvas, st); break;
        case 8: rc = launch_pointnet<8>(a, tg, w, net, canvas, st); break;
        default: return fail(LAV_EINVAL, "lav_pillar_scatter: point width D=%d not instantiated (4,5,8,11)", D);
    }
    if (rc) return rc;

    if (unique_coords || inverse || counts) {
        rc = exclusive_scan(w.cell_count, ncells, 1, w.cell_rank, w.block_sums, w.totals + 0, st);
        if (rc) return rc;
        if (total > 0) {
            rc = exclusive_scan(w.key, total, 2, w.kept_rank, w.block_sums, w.totals + 1, st);
            if (rc) return rc;
        } else {
            LAV_HIP(hipMemsetAsync(w.totals + 1, 0, sizeof(int), st));
        }
        if (unique_coords) {
            hipLaunchKernelGGL(k_unique_coords, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, a, w.cell_count, w.cell_rank, ncells, unique_coords);
            LAV_LAUNCH_CHECK();
        }
        if (inverse && total > 0) {
            hipLaunchKernelGGL(k_inverse, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, w.key, w.kept_rank, w.cell_rank, inverse);
            LAV_LAUNCH_CHECK();
        }
        if (counts) {
            hipLaunchKernelGGL(k_counts, dim3(1), dim3(1), 0, st, w.totals + 0, w.totals + 1, counts);
            LAV_LAUNCH_CHECK();
        }
    }
    return LAV_OK;
}

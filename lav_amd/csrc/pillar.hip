// PointPillars dynamic voxelisation -> decoration -> PointNet -> scatter-max -> dense BEV canvas.
//
// Replaces PointPillarNet.forward of the reference (lav/models/point_pillar.py:92-116) including the two
// torch_scatter calls (:33, :62) and coords.unique(dim=0) (:82).  gfx950 only.
//
// Data flow of the canvas path (all buffers in HBM; v2 agent: N ~ 196k points x 11 floats, 320x320 cells, C = 64).
// Points are binned by CANVAS TILE (one canvas row x 160 columns = the unit one workgroup writes), not by cell:
//
//   k_tile_count   1 thread / point : float32 cell id exactly as the reference, tile id, slot = arrival number
//                                     inside the tile (int atomic on 640 counters; order is irrelevant, see below)
//   k_tile_scan    1 workgroup      : exclusive prefix of the tile counters
//   k_tile_place   1 thread / point : writes a contiguous 48-byte record {11 floats, cell key} at
//                                     rec[tile_offset + slot]  -> the big kernel never chases indices
//   k_tile_pointnet  one workgroup per (cloud, canvas row, column tile):
//        (a) per-cell xyz sums and counts in LDS (64-bit fixed-point atomics: order independent, so the result
//            is bit-identical for ANY arrival order / input permutation; exact to 2^-32 m), means
//        (b) every wave takes passes of 32 records and runs BOTH PointNet layers on the matrix cores with all
//            activations in registers (v_mfma_f32_32x32x2_f32, exact fp32):
//              layer 1 (transposed)  D1[c][p]  = sum_k W1[k][c] * F[k][p]     A = weights, B = point features
//                 lane l supplies feature 2s+(l>>5) of point l&31 at k-step s; the bias rides as feature 16 (=1).
//                 D1 leaves lane (p, half) holding channels c = 32*mt + (r&3) + 8*(r>>2) + 4*half  (r = 0..15)
//              layer 2               D2[p][c2] = sum_c H1[p][c] * W2[c][c2]   A = relu(D1) AS IT SITS, B = weights
//                 k-step (mt, r) uses k = 32*mt + (r&3) + 8*(r>>2) + 4*half - a permutation of 0..63, which a
//                 sum does not care about - so no lane shuffles or LDS round trip between the layers.
//        (c) unsigned-integer max of the float bits (values >= 0 after ReLU) into an LDS tile [C][tile_w|1]
//            (odd stride: conflict-free)
//        (d) the tile - zeros for empty cells included - streams to the NCHW canvas, 256 B per wave-instruction.
//            The canvas is written exactly once and never read or memset: algorithmic traffic
//            4*(N*D + C*ny*nx) bytes.  Tiles without points skip (a)-(c) and stream zeros.
//
// Index outputs (unique_coords / inverse, the "bit-exact pillar indices" of the parity contract) come from a
// separate per-cell occupancy path (count, two scans) that only runs when they are requested.
//
// Cell-id arithmetic is float32 exactly as the reference's: (x - min_x) * ppm, subtraction and multiplication
// rounded separately (this file is compiled with FMA contraction off), then truncation.  Because the product can
// round up to exactly nx (resp. ny) - y = nextafter(40,0) with min_y=-40 gives yi = 320 - the key space has one
// extra row and column; the canvas write clamps them like the reference (:89) and lets the LATER pillar in unique
// order win, by processing such "overflow" cells as extra layers after the regular ones.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.hpp"

// parity-critical float32 arithmetic: no fused multiply-add contraction anywhere in this file
// (HIP's __fadd_rn/__fmul_rn are plain operators that clang would otherwise fuse)
#pragma clang fp contract(off)

namespace {
using namespace lav;

constexpr int C = 64;            // PointNet width (config num_features [64,64])
constexpr int MAX_BATCH = 64;    // per-call limit on clouds (kernel-argument table)
constexpr int TILE_MAX_W = 80;   // canvas columns per workgroup tile (LDS tile [64][tile_w|1] floats): 4 workgroups per CU
constexpr int NSUB = 8;          // arrival counters per tile, on different cache lines, to spread the atomic traffic
constexpr int MAX_LAYERS = 64;   // regular + overflow layers a tile may have (2-4 for square grids)
constexpr int REC_MAX = 16;      // dwords per point record the workspace is sized for (D <= 15)
constexpr double FIX_SCALE = 4294967296.0;  // 2^32 fixed-point scale of the per-cell coordinate sums

struct PillarArgs {
    const float *points;
    int batch, max_points, D;
    int n[MAX_BATCH];
    float min_x, max_x, min_y, max_y, ppm;
    int nx, ny;  // nx = number of xi cells = canvas columns; ny = number of yi cells = canvas rows
    int KX, KY;  // key space (nx+1) x (ny+1)
    int T, TW;   // column tiles per canvas row, columns per tile
    unsigned long long *trace;  // debug (LAV_PILLAR_TRACE): [ntiles][8] wall-clock stamps of thread 0, else null
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// cell key of a point, or -1 (grid_locations, point_pillar.py:70-79)
__device__ __forceinline__ int cell_key(const PillarArgs &a, int b, float x, float y) {
    // NaN fails every comparison and is dropped, as in torch
    if (!(x >= a.min_x && x < a.max_x && y >= a.min_y && y < a.max_y)) return -1;
    const float fx = (x - a.min_x) * a.ppm;  // two roundings (contraction is off)
    const float fy = (y - a.min_y) * a.ppm;
    const int xi = (int)fx, yi = (int)fy;    // truncation; both are in [0, nx] x [0, ny]
    return (b * a.KX + xi) * a.KY + yi;
}

// canvas tile that cell (b, xi, yi) lands on (scatter_points clamp, point_pillar.py:89)
__device__ __forceinline__ int tile_of(const PillarArgs &a, int b, int xi, int yi) {
    const int r = min(max(a.ny - 1 - xi, 0), a.ny - 1);
    const int col = min(yi, a.nx - 1);
    return (b * a.ny + r) * a.T + col / a.TW;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_count(PillarArgs a, int *__restrict__ key, int *__restrict__ slot,
                                                    int *__restrict__ tile_count) {
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / a.max_points);
    const int i = (int)(gid - (long)b * a.max_points);
    int k = -1;
    if (i < a.n[b]) {
        const float *pt = a.points + gid * a.D;
        k = cell_key(a, b, pt[0], pt[1]);
    }
    key[gid] = k;
    if (k >= 0) {
        const int cellk = k - b * a.KX * a.KY;
        // counters are laid out [sub][tile]; the sub-bucket only decorrelates concurrent arrivals
        const int sub = (threadIdx.x ^ (threadIdx.x >> 6) ^ blockIdx.x) & (NSUB - 1);
        const int ntiles = a.batch * a.ny * a.T;
        slot[gid] = sub | (atomicAdd(&tile_count[sub * ntiles + tile_of(a, b, cellk / a.KY, cellk % a.KY)], 1) << 3);
    }
}

// Exclusive prefix of the arrival counters in (tile, sub) order - a tile's NSUB buckets end up contiguous - read from
// their [sub][tile] layout, and re-zeroing of nothing: n = ntiles*NSUB is a few thousand, so one workgroup does it with
// a serial run of ceil(n/1024) items per thread and a single 1024-wide block scan.
__global__ __launch_bounds__(1024) void k_tile_scan(const int *__restrict__ count, int n, int *__restrict__ offset,
                                                    int4 *__restrict__ order, int *__restrict__ g_tmp) {
    __shared__ int wsum[16];
    __shared__ int bucket[33];
    extern __shared__ int s_tile[];  // [ntiles] points per tile, then [ntiles] first-record offset per tile
    const int ntiles = n / NSUB;
    // per-tile scratch: LDS for the usual few clouds, the workspace for large batches (one workgroup either way)
    int *s_tot = g_tmp ? g_tmp : s_tile, *s_p0 = s_tot + ntiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int t = tid; t < ntiles; t += 1024) s_tot[t] = 0;
    if (tid < 33) bucket[tid] = 0;
    __syncthreads();
    const int per = (n + 1023) / 1024;
    const int i0 = tid * per;
    int sum = 0;
    for (int j = 0; j < per; ++j) {
        const int i = i0 + j;
        if (i < n) {
            const int c = count[(i % NSUB) * ntiles + i / NSUB];
            sum += c;
            if (c) atomicAdd(&s_tot[i / NSUB], c);
        }
    }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int run = inc - sum;
#pragma unroll
    for (int w = 0; w < 16; ++w)
        if (w < wid) run += wsum[w];
    for (int j = 0; j < per; ++j) {
        const int i = i0 + j;
        if (i < n) {
            offset[i] = run;
            if (i % NSUB == 0) s_p0[i / NSUB] = run;
            run += count[(i % NSUB) * ntiles + i / NSUB];
        }
    }
    if (i0 < n && i0 + per >= n) offset[n] = run;  // the thread owning the last item writes the grand total
    // Dispatch order of the PointNet kernel's tiles: heaviest first (32 buckets of 32 points), so that the second round of
    // workgroups on the chip is made of the light tiles.  Order inside a bucket is arbitrary - tiles are independent.
    // Each entry carries the tile's record range, so a workgroup needs ONE 16-byte load to know its work.
    __syncthreads();   // s_p0 / s_tot complete
    int total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) total += wsum[w];
    if (total < 65536) {   // light clouds (config #2: 32 768 points): every tile is about one pass - keep the canvas order,
                           // which measured 2.4 us faster there (adjacent workgroups write adjacent canvas rows)
        for (int t = tid; t < ntiles; t += 1024) order[t] = make_int4(t, s_p0[t], s_p0[t] + s_tot[t], 0);
        return;
    }
    for (int t = tid; t < ntiles; t += 1024) atomicAdd(&bucket[31 - min(s_tot[t] >> 5, 31)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int b = 0; b < 32; ++b) { const int c = bucket[b]; bucket[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024)
        order[atomicAdd(&bucket[31 - min(s_tot[t] >> 5, 31)], 1)] = make_int4(t, s_p0[t], s_p0[t] + s_tot[t], 0);
}

__global__ __launch_bounds__(256) void k_zero_ints(int *__restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0;
}

template <int D>
__global__ __launch_bounds__(256) void k_tile_place(PillarArgs a, const int *__restrict__ key, const int *__restrict__ slot,
                                                    const int *__restrict__ tile_offset, float *__restrict__ rec) {
    constexpr int RS = D + 1;
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k < 0) return;
    const int b = (int)(gid / a.max_points);
    const int cellk = k - b * a.KX * a.KY;
    const int sl = slot[gid];
    const long j = tile_offset[tile_of(a, b, cellk / a.KY, cellk % a.KY) * NSUB + (sl & (NSUB - 1))] + (sl >> 3);
    const float *pt = a.points + gid * D;
    float v[RS];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = pt[d];
    v[D] = __int_as_float(k);
    float *o = rec + j * RS;
    if constexpr (RS % 4 == 0) {
#pragma unroll
        for (int q = 0; q < RS / 4; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
        for (int d = 0; d < RS; ++d) o[d] = v[d];
    }
}

template <int D>
__device__ __forceinline__ void load_record(const float *__restrict__ rec, long j, float (&v)[D + 1]) {
    constexpr int RS = D + 1;
    const float *p = rec + j * RS;
    if constexpr (RS % 4 == 0) {
#pragma unroll
        for (int q = 0; q < RS / 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(p)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int d = 0; d < RS; ++d) v[d] = p[d];
    }
}

template <int D>
__device__ __forceinline__ void decorate(const PillarArgs &a, const float *pt, const float *mean3, int xi, int yi, float *f) {
#pragma unroll
    for (int d = 0; d < D; ++d) f[d] = pt[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) f[D + d] = pt[d] - mean3[d];
    // reference decorate(): x - (yi/ppm + min_x), y - (xi/ppm + min_y)  (sic: swapped, un-centred; :57-58)
    f[D + 3] = pt[0] - ((float)yi / a.ppm + a.min_x);
    f[D + 4] = pt[1] - ((float)xi / a.ppm + a.min_y);
}

// ---------------------------------------------------------------------------------------------------------
template <int D, bool USE_MFMA, bool TRACE = false>
__global__ __launch_bounds__(256, 3) void k_tile_pointnet(PillarArgs a, const float *__restrict__ rec,
                                                          const int *__restrict__ tile_offset, const int4 *__restrict__ tile_order,
                                                          const float *__restrict__ w1, const float *__restrict__ b1,
                                                          const float *__restrict__ w2, const float *__restrict__ b2,
                                                          float *__restrict__ canvas) {
    constexpr int K1 = D + 5;          // decorated features
    constexpr int KS1 = (K1 + 2) / 2;  // layer-1 k-steps incl. the bias feature (K1=16 -> 9)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TW = a.TW, TWP = TW | 1;
    float *tile = reinterpret_cast<float *>(smem);                                                        // [C][TWP]
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(smem + ((C * TWP * 4 + 15) & ~15));  // [TW][3]
    float *means = reinterpret_cast<float *>(sums + TW * 3);                                              // [TW][3]
    int *cnt = reinterpret_cast<int *>(means + TW * 3);                                                   // [TW]
    int *nl = cnt + ((TW + 3) & ~3);                                                                      // [MAX_LAYERS]
    float *w2s = reinterpret_cast<float *>(nl + MAX_LAYERS);                                              // [C][C]

    // Workgroups are dispatched in blockIdx order and the 1280 tiles need two rounds on the chip: k_tile_scan sorted the
    // tiles heaviest-first, so the second round (and the tail) is made of the sparse far field.
    const int4 work = tile_order[blockIdx.x];   // (tile, first record, end record)
    const int wg = work.x;
    const int t = wg % a.T;
    const int r = (wg / a.T) % a.ny;  // canvas row
    const int b = wg / (a.T * a.ny);
    const int c0 = t * TW;
    const int c1 = min(a.nx, c0 + TW);
    const int tw = c1 - c0;  // live columns in this tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
#define LAV_STAMP(i) do { if constexpr (TRACE) { if (tid == 0) a.trace[(long)wg * 8 + (i)] = wall_clock64(); } } while (0)
    LAV_STAMP(0);
    const int p0 = work.y, p1 = work.z;
    if constexpr (TRACE) { if (tid == 0) a.trace[(long)wg * 8 + 7] = (unsigned long long)(p1 - p0); }
    float *dst = canvas + ((long)b * C * a.ny + r) * a.nx + c0;
    const long cstride = (long)a.ny * a.nx;

    if (p0 == p1) {  // empty tile: stream zeros (workgroup-uniform)
        for (int ch = wid; ch < C; ch += 4)
            for (int j = lane; j < tw; j += 64) dst[ch * cstride + j] = 0.f;
        LAV_STAMP(6);
        return;
    }
    LAV_STAMP(1);
    // Issue every global load that depends only on (p0, p1) before touching LDS, so the workgroup pays ONE memory
    // round trip here instead of one per phase: layer-2 weights, layer-1 weights, this thread's record for the
    // sums sweep and this lane's record for the wave's first PointNet pass.
    constexpr int W2R = C * C / 256;
    float w2r[W2R];
    if (USE_MFMA) {
#pragma unroll
        for (int i = 0; i < W2R; ++i) w2r[i] = w2[tid + 256 * i];
    }
    float a1[2][KS1];  // layer-1 A operand: W1[2s+half][32*mt + l31], bias as k = K1
    float b2v[2];
    const int npass = (p1 - p0 + 31) >> 5;
    if (USE_MFMA && wid < npass) {
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
    float srec[4];  // x, y, z, key of record p0 + tid (first chunk of the sums sweep)
    {
        const float *rp = rec + (long)min(p0 + tid, p1 - 1) * (D + 1);
        srec[0] = rp[0]; srec[1] = rp[1]; srec[2] = rp[2]; srec[3] = rp[D];
    }
    float prec[D + 1];  // record of this lane for the wave's first pass
    if (USE_MFMA) load_record<D>(rec, min(p0 + wid * 32 + l31, p1 - 1), prec);

    for (int i = tid; i < C * TWP; i += 256) tile[i] = 0.f;
    if (tid < MAX_LAYERS) nl[tid] = 0;
    if (USE_MFMA) {
#pragma unroll
        for (int i = 0; i < W2R; ++i) w2s[tid + 256 * i] = w2r[i];
    }

    // key rows that land on canvas row r, and overflow columns of the last tile (reference clamp, :89)
    const int xi_lo = r > 0 ? a.ny - 1 - r : max(a.ny - 1, 0);
    const int xi_hi = r > 0 ? xi_lo : a.nx;
    const int n_over = (c1 == a.nx) ? max(0, a.ny - a.nx + 1) : 0;
    const int nlay_y = n_over + 1;
    const int nlayers = (xi_hi - xi_lo + 1) * nlay_y;
    const int cellbase = b * a.KX * a.KY;
    // record -> (layer, tile column, xi, yi).  Layers are ordered like the reference's sorted unique rows.
    auto classify = [&](int k, int &layer, int &col, int &xi, int &yi) {
        const int cellk = k - cellbase;
        xi = cellk / a.KY;
        yi = cellk - xi * a.KY;
        const int over = yi >= a.nx ? yi - a.nx + 1 : 0;
        layer = (xi - xi_lo) * nlay_y + over;
        col = min(yi, a.nx - 1) - c0;
    };

    for (int L = 0; L < min(nlayers, MAX_LAYERS); ++L) {
        if (L > 0 && nl[L] == 0) continue;  // workgroup-uniform; nl[] is complete after layer 0's first barrier pair
        __syncthreads();
        for (int j = tid; j < TW * 3; j += 256) sums[j] = 0ull;
        for (int j = tid; j < TW; j += 256) cnt[j] = 0;
        __syncthreads();
        if (L == 0) LAV_STAMP(2);
        // (a) per-cell coordinate sums and counts of this layer (first 256 records were prefetched)
        for (int j = p0 + tid; j < p1; j += 256) {
            float x, y, z;
            int k;
            if (j == p0 + tid) {
                x = srec[0]; y = srec[1]; z = srec[2]; k = __float_as_int(srec[3]);
            } else {
                const float *rp = rec + (long)j * (D + 1);
                x = rp[0]; y = rp[1]; z = rp[2]; k = __float_as_int(rp[D]);
            }
            int layer, col, xi, yi;
            classify(k, layer, col, xi, yi);
            if (L == 0 && layer != 0) atomicAdd(&nl[min(layer, MAX_LAYERS - 1)], 1);
            if (layer == L) {
                const float xyz[3] = {x, y, z};
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const long long q = __double2ll_rn((double)xyz[d] * FIX_SCALE);
                    atomicAdd(&sums[col * 3 + d], (unsigned long long)q);
                }
                atomicAdd(&cnt[col], 1);
            }
        }
        __syncthreads();
        for (int j = tid; j < tw; j += 256) {
            const int n = cnt[j];
            if (n > 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    means[j * 3 + d] = (float)((double)(long long)sums[j * 3 + d] / ((double)n * FIX_SCALE));
            }
        }
        if (L > 0) {  // a later pillar replaces whatever an earlier one put on the same canvas cell
            for (int i = tid; i < C * tw; i += 256) {
                const int ch = i / tw, j = i - ch * tw;
                if (cnt[j] > 0) tile[ch * TWP + j] = 0.f;
            }
        }
        __syncthreads();
        if (L == 0) LAV_STAMP(3);

        if constexpr (!USE_MFMA) {
            // cross-check path: one point per thread, plain fp32 FMAs (slow; selected by LAV_PILLAR_IMPL=valu)
            for (int j = p0 + tid; j < p1; j += 256) {
                float v[D + 1];
                load_record<D>(rec, j, v);
                int layer, col, xi, yi;
                classify(__float_as_int(v[D]), layer, col, xi, yi);
                if (layer != L) continue;
                float f[K1];
                decorate<D>(a, v, means + col * 3, xi, yi, f);
                float h1[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float acc = b1[c];
#pragma unroll
                    for (int k = 0; k < K1; ++k) acc = fmaf(f[k], w1[k * C + c], acc);
                    h1[c] = acc > 0.f ? acc : 0.f;
                }
                for (int c = 0; c < C; ++c) {
                    float acc = b2[c];
#pragma unroll
                    for (int k = 0; k < C; ++k) acc = fmaf(h1[k], w2[k * C + c], acc);
                    const float o = acc > 0.f ? acc : 0.f;
                    atomicMax(reinterpret_cast<unsigned *>(&tile[c * TWP + col]), __float_as_uint(o));
                }
            }
        } else {
            for (int pass = wid; pass < npass; pass += 4) {
                const int j = p0 + pass * 32 + l31;
                bool live = j < p1;
                float v[D + 1];
                if (pass == wid) {
#pragma unroll
                    for (int d = 0; d <= D; ++d) v[d] = prec[d];
                } else {
                    load_record<D>(rec, live ? j : p0, v);
                }
                int layer, col, xi, yi;
                classify(__float_as_int(v[D]), layer, col, xi, yi);
                live = live && layer == L;
                const int mycol = live ? col : -1;
                float f[K1];
                decorate<D>(a, v, means + (live ? col : 0) * 3, xi, yi, f);
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
                // after layer 2 this lane holds channel c2 of the 16 points in MFMA rows (q&3)+8*(q>>2)+4*half;
                // their tile columns come from the lanes that loaded them
                int cols[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) cols[q] = __shfl(mycol, (q & 3) + 8 * (q >> 2) + 4 * half, 64);
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
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float o = d2[q] > 0.f ? d2[q] : 0.f;  // also maps -0 and NaN to +0
                        if (cols[q] >= 0) atomicMax(trow + cols[q], __float_as_uint(o));
                    }
                }
            }
        }
    }
    LAV_STAMP(4);
    __syncthreads();
    LAV_STAMP(5);
    // (d) stream the tile out; canvas [B][C][ny][nx]
#pragma unroll 4
    for (int ch = wid; ch < C; ch += 4) {
        const float *src = tile + ch * TWP;
        float *d = dst + ch * cstride;
        for (int j = lane; j < tw; j += 64) d[j] = src[j];
    }
    LAV_STAMP(6);
#undef LAV_STAMP
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
// Index outputs
__global__ __launch_bounds__(256) void k_cell_count(long total, const int *__restrict__ key, int *__restrict__ cell_count) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k >= 0) atomicAdd(&cell_count[k], 1);
}

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

void tile_geometry(int nx, int &T, int &TW) {
    static const int max_w = [] {  // experiment knob: LAV_PILLAR_TW=<columns per tile>
        const char *e = getenv("LAV_PILLAR_TW");
        const int v = e ? atoi(e) : 0;
        return v >= 4 && v <= 320 ? v : TILE_MAX_W;
    }();
    T = (nx + max_w - 1) / max_w;
    const int tw = (nx + T - 1) / T;
    TW = (tw + 3) / 4 * 4;
}

struct Workspace {
    int *tile_count, *tile_offset, *key, *slot;
    int4 *tile_order;
    int *tile_tmp;
    float *rec;
    int *cell_count, *cell_rank, *kept_rank, *block_sums, *totals;
    unsigned long long *cell_sums;  // [min(ncells, points)][3] fixed-point coordinate sums (training entry lav_pillar_decorate)
};

size_t carve(Arena &ar, Workspace &w, int batch, int max_points, const lav_grid *g) {
    int T, TW;
    tile_geometry(g->nx, T, TW);
    const size_t ntiles = (size_t)batch * g->ny * T;
    const size_t ncells = (size_t)batch * (g->nx + 1) * (g->ny + 1);
    const size_t total = (size_t)batch * max_points;
    const size_t nmax = ncells > total ? ncells : total;
    w.tile_count = ar.take<int>(ntiles * NSUB + 1);
    w.tile_offset = ar.take<int>(ntiles * NSUB + 1);
    w.tile_order = ar.take<int4>(ntiles);
    w.tile_tmp = ar.take<int>(2 * ntiles);
    w.key = ar.take<int>(total);
    w.slot = ar.take<int>(total);
    w.rec = ar.take<float>(total * REC_MAX);
    w.cell_count = ar.take<int>(ncells);
    w.cell_rank = ar.take<int>(ncells + 1);
    w.kept_rank = ar.take<int>(total + 1);
    w.block_sums = ar.take<int>((nmax + SCAN_TILE - 1) / SCAN_TILE + 1);
    w.totals = ar.take<int>(4);
    w.cell_sums = ar.take<unsigned long long>((ncells < total ? ncells : total) * 3 + 3);
    return align_up(ar.used, 256);
}

bool use_valu_impl() {
    const char *e = getenv("LAV_PILLAR_IMPL");
    return e && e[0] == 'v';
}

// debug: per-workgroup phase times of one k_tile_pointnet launch (100 MHz wall clock -> us)
void dump_trace(const unsigned long long *d_trace, int ntiles, hipStream_t st) {
    std::vector<unsigned long long> h((size_t)ntiles * 8);
    if (hipStreamSynchronize(st) != hipSuccess) return;
    if (hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < ntiles; ++i) { t0 = std::min(t0, h[i * 8]); tend = std::max(tend, h[i * 8 + 6]); }
    auto us = [&](unsigned long long t) { return (double)(t - t0) / 100.0; };
    fprintf(stderr, "[pillar trace] %d tiles, span %.1f us\n", ntiles, us(tend));
    int nempty = 0; double e_start = 0, e_dur = 0, e_last = 0;
    double ph[6] = {0, 0, 0, 0, 0, 0}, n_start_max = 0, n_end_max = 0; int nn = 0;
    struct Row { double start, end; unsigned long long n; double p[6]; };
    std::vector<Row> rows;
    for (int i = 0; i < ntiles; ++i) {
        const unsigned long long *r = &h[i * 8];
        if (r[7] == 0) { ++nempty; e_start += us(r[0]); e_dur += (double)(r[6] - r[0]) / 100.0; e_last = std::max(e_last, us(r[6])); continue; }
        Row w; w.start = us(r[0]); w.end = us(r[6]); w.n = r[7];
        for (int k = 0; k < 6; ++k) { w.p[k] = (double)(r[k + 1] - r[k]) / 100.0; ph[k] += w.p[k]; }
        n_start_max = std::max(n_start_max, w.start); n_end_max = std::max(n_end_max, w.end); ++nn; rows.push_back(w);
    }
    fprintf(stderr, "  empty tiles %d: mean start %.1f us, mean duration %.2f us, last end %.1f us\n", nempty, nempty ? e_start / nempty : 0, nempty ? e_dur / nempty : 0, e_last);
    fprintf(stderr, "  tiles with points %d: last start %.1f us, last end %.1f us; mean phase us: offsets %.2f | prefetch+init %.2f | sums+means %.2f | pointnet %.2f | barrier %.2f | stream-out issue %.2f\n",
            nn, n_start_max, n_end_max, ph[0] / std::max(nn, 1), ph[1] / std::max(nn, 1), ph[2] / std::max(nn, 1), ph[3] / std::max(nn, 1), ph[4] / std::max(nn, 1), ph[5] / std::max(nn, 1));
    std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.end > y.end; });
    for (size_t i = 0; i < std::min<size_t>(rows.size(), 8); ++i)
        fprintf(stderr, "  late tile: n=%llu start %.1f end %.1f | %.2f %.2f %.2f %.2f %.2f %.2f\n", rows[i].n, rows[i].start, rows[i].end, rows[i].p[0], rows[i].p[1], rows[i].p[2], rows[i].p[3], rows[i].p[4], rows[i].p[5]);
    // start-time histogram (2 us bins)
    std::vector<int> hist(40, 0);
    for (int i = 0; i < ntiles; ++i) hist[std::min<size_t>(39, (size_t)(us(h[i * 8]) / 2.0))]++;
    fprintf(stderr, "  starts per 2 us:");
    for (int i = 0; i < 40; ++i) if (hist[i]) fprintf(stderr, " [%d]=%d", i * 2, hist[i]);
    fprintf(stderr, "\n");
}

template <int D>
int launch_canvas(const PillarArgs &a, const Workspace &w, const lav_pointnet *net, float *canvas, hipStream_t st) {
    const long total = (long)a.batch * a.max_points;
    const int ntiles = a.batch * a.ny * a.T;
    const int tok_prep = timer_begin("pillar_prep", st);
    hipLaunchKernelGGL(k_zero_ints, dim3((ntiles * NSUB + 255) / 256), dim3(256), 0, st, w.tile_count, ntiles * NSUB + 1);
    LAV_LAUNCH_CHECK();
    if (total > 0) {
        hipLaunchKernelGGL(k_tile_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, w.key, w.slot, w.tile_count);
        LAV_LAUNCH_CHECK();
    }
    const bool scan_lds = 2 * (size_t)ntiles * sizeof(int) <= 48 * 1024;
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), scan_lds ? 2 * (size_t)ntiles * sizeof(int) : 0, st, w.tile_count, ntiles * NSUB,
                       w.tile_offset, w.tile_order, scan_lds ? nullptr : w.tile_tmp);
    LAV_LAUNCH_CHECK();
    if (total > 0) {
        hipLaunchKernelGGL((k_tile_place<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, w.key, w.slot, w.tile_offset, w.rec);
        LAV_LAUNCH_CHECK();
    }
    timer_end(tok_prep, st);
    const int TWP = a.TW | 1;
    const size_t lds = (((size_t)C * TWP * 4 + 15) & ~(size_t)15) + (size_t)a.TW * 3 * 8 + (size_t)a.TW * 3 * 4 +
                       (size_t)((a.TW + 3) & ~3) * 4 + (size_t)MAX_LAYERS * 4 + (size_t)C * C * 4;
    static bool attr_set = false;
    if (!attr_set) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_pointnet<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_pointnet<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_pointnet<D, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    static const bool want_trace = getenv("LAV_PILLAR_TRACE") != nullptr;
    static unsigned long long *d_trace = nullptr;
    static int trace_runs = 0;
    PillarArgs at = a;
    if (want_trace) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)ntiles * 8 * sizeof(unsigned long long)));
        LAV_HIP(hipMemsetAsync(d_trace, 0, (size_t)ntiles * 8 * sizeof(unsigned long long), st));
        at.trace = d_trace;
    }
    const int tok = timer_begin("pointnet_scatter", st);
    if (want_trace) {
        hipLaunchKernelGGL((k_tile_pointnet<D, true, true>), dim3(ntiles), dim3(256), lds, st, at, w.rec, w.tile_offset, w.tile_order, net->w1, net->b1, net->w2, net->b2, canvas);
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        if (++trace_runs == 20) dump_trace(d_trace, ntiles, st);
        return LAV_OK;
    }
    if (use_valu_impl())
        hipLaunchKernelGGL((k_tile_pointnet<D, false>), dim3(ntiles), dim3(256), lds, st, a, w.rec, w.tile_offset, w.tile_order, net->w1, net->b1, net->w2, net->b2, canvas);
    else
        hipLaunchKernelGGL((k_tile_pointnet<D, true>), dim3(ntiles), dim3(256), lds, st, a, w.rec, w.tile_offset, w.tile_order, net->w1, net->b1, net->w2, net->b2, canvas);
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
    LAV_REQUIRE(D + 1 <= REC_MAX, "lav_pillar_scatter: point width %d too large", D);
    hipStream_t st = static_cast<hipStream_t>(stream);

    Arena ar(workspace, workspace_bytes);
    Workspace w;
    carve(ar, w, batch, max_points, grid);
    if (!workspace || !ar.ok()) return fail(LAV_EWORKSPACE, "lav_pillar_scatter: workspace %zu < %zu bytes", workspace_bytes, ar.used);

    PillarArgs a;
    a.trace = nullptr;
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
    tile_geometry(a.nx, a.T, a.TW);
    {   // canvas row 0 collects key rows ny-1..nx, the last column tile collects key columns nx-1..ny (reference clamp)
        const long lay = (long)(a.nx - (a.ny - 1) + 1 > 1 ? a.nx - (a.ny - 1) + 1 : 1) * ((a.ny - a.nx + 1 > 0 ? a.ny - a.nx + 1 : 0) + 1);
        LAV_REQUIRE(lay <= MAX_LAYERS, "lav_pillar_scatter: grid %dx%d needs %ld clamp layers (max %d)", a.nx, a.ny, lay, MAX_LAYERS);
    }

    int rc;
    switch (D) {
        case 11: rc = launch_canvas<11>(a, w, net, canvas, st); break;
        case 4: rc = launch_canvas<4>(a, w, net, canvas, st); break;
        case 5: rc = launch_canvas<5>(a, w, net, canvas, st); break;
        case 8: rc = launch_canvas<8>(a, w, net, canvas, st); break;
        default: return fail(LAV_EINVAL, "lav_pillar_scatter: point width D=%d not instantiated (4,5,8,11)", D);
    }
    if (rc) return rc;

    if (unique_coords || inverse || counts) {
        const long ncells = (long)batch * a.KX * a.KY;
        const long total = (long)batch * max_points;
        LAV_HIP(hipMemsetAsync(w.cell_count, 0, ncells * sizeof(int), st));
        if (total > 0) {
            hipLaunchKernelGGL(k_cell_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, w.key, w.cell_count);
            LAV_LAUNCH_CHECK();
        }
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

// =========================================================================================================
// Training-side entry points: the torch_scatter ABI the reference's PointPillarNet uses in train mode
// (lav/models/point_pillar.py:55-68 decorate + scatter_mean, :33 scatter_max), with what autograd needs.
// =========================================================================================================
namespace {
__global__ __launch_bounds__(256) void k_keys(PillarArgs a, int *__restrict__ key) {
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / a.max_points);
    const int i = (int)(gid - (long)b * a.max_points);
    int k = -1;
    if (i < a.n[b]) {
        const float *pt = a.points + gid * a.D;
        k = cell_key(a, b, pt[0], pt[1]);
    }
    key[gid] = k;
}

__global__ __launch_bounds__(256) void k_dec_sums(PillarArgs a, const int *__restrict__ key, const int *__restrict__ cell_rank,
                                                  const int *__restrict__ kept_rank, unsigned long long *__restrict__ sums,
                                                  int *__restrict__ inverse, int *__restrict__ kept_src) {
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k < 0) return;
    const int p = cell_rank[k], j = kept_rank[gid];
    if (inverse) inverse[j] = p;
    if (kept_src) kept_src[j] = (int)gid;
    const float *pt = a.points + gid * a.D;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        atomicAdd(&sums[(long)p * 3 + d], (unsigned long long)__double2ll_rn((double)pt[d] * FIX_SCALE));
}

__global__ __launch_bounds__(256) void k_dec_write(PillarArgs a, const int *__restrict__ key, const int *__restrict__ cell_rank,
                                                   const int *__restrict__ cell_count, const int *__restrict__ kept_rank,
                                                   const unsigned long long *__restrict__ sums, float *__restrict__ out) {
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int k = key[gid];
    if (k < 0) return;
    const int b = (int)(gid / a.max_points);
    const int cellk = k - b * a.KX * a.KY;
    const int xi = cellk / a.KY, yi = cellk - xi * a.KY;
    const int p = cell_rank[k], n = cell_count[k];
    const float *pt = a.points + gid * a.D;
    float *o = out + (long)kept_rank[gid] * (a.D + 5);
    for (int d = 0; d < a.D; ++d) o[d] = pt[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float mean = (float)((double)(long long)sums[(long)p * 3 + d] / ((double)n * FIX_SCALE));
        o[a.D + d] = pt[d] - mean;
    }
    o[a.D + 3] = pt[0] - ((float)yi / a.ppm + a.min_x);  // sic: swapped, un-centred (point_pillar.py:57-58)
    o[a.D + 4] = pt[1] - ((float)xi / a.ppm + a.min_y);
}

__device__ __forceinline__ unsigned ord_f(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(256) void k_smax_init(unsigned *__restrict__ out_u, int *__restrict__ argmax, long pc, int n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < pc) { out_u[i] = 0u; argmax[i] = n; }
}
__global__ __launch_bounds__(256) void k_smax_max(const float *__restrict__ src, const int *__restrict__ index, long nc, int C,
                                                  unsigned *__restrict__ out_u) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nc) return;
    const long i = e / C;
    const int c = (int)(e - i * C);
    atomicMax(&out_u[(long)index[i] * C + c], ord_f(src[e]));
}
__global__ __launch_bounds__(256) void k_smax_arg(const float *__restrict__ src, const int *__restrict__ index, long nc, int C,
                                                  const unsigned *__restrict__ out_u, int *__restrict__ argmax) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nc) return;
    const long i = e / C;
    const int c = (int)(e - i * C);
    const long o = (long)index[i] * C + c;
    if (ord_f(src[e]) == out_u[o]) atomicMin(&argmax[o], (int)i);  // ties: lowest source row (deterministic)
}
__global__ __launch_bounds__(256) void k_smax_fin(unsigned *__restrict__ out_u, const int *__restrict__ argmax, long pc, int n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pc) return;
    const float v = argmax[i] < n ? unord_f(out_u[i]) : 0.f;  // empty segment: 0, argmax = n (torch_scatter's convention)
    out_u[i] = __float_as_uint(v);
}
__global__ __launch_bounds__(256) void k_smax_bwd(const float *__restrict__ grad_out, const int *__restrict__ argmax, long pc, int C,
                                                  int n, float *__restrict__ grad_src) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pc) return;
    const int a = argmax[i];
    if (a < n) grad_src[(long)a * C + (i % C)] = grad_out[i];  // every source element is the arg-max of at most one output
}
}  // namespace

extern "C" int lav_pillar_decorate(const float *points, const int *h_num_points, int batch, int max_points, int D,
                                   const lav_grid *grid, int *unique_coords, int *inverse, int *kept_src, float *decorated,
                                   int *counts, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(grid && h_num_points && counts && decorated, "lav_pillar_decorate: null argument");
    LAV_REQUIRE(batch >= 1 && batch <= MAX_BATCH, "lav_pillar_decorate: batch %d outside [1,%d]", batch, MAX_BATCH);
    LAV_REQUIRE(max_points >= 0 && (points || max_points == 0) && D >= 3 && D < REC_MAX, "lav_pillar_decorate: bad points");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Arena ar(workspace, workspace_bytes);
    Workspace w;
    carve(ar, w, batch, max_points, grid);
    if (!workspace || !ar.ok()) return fail(LAV_EWORKSPACE, "lav_pillar_decorate: workspace %zu < %zu bytes", workspace_bytes, ar.used);
    PillarArgs a;
    a.trace = nullptr;
    a.points = points; a.batch = batch; a.max_points = max_points; a.D = D;
    for (int b = 0; b < MAX_BATCH; ++b) a.n[b] = b < batch ? (h_num_points[b] < max_points ? h_num_points[b] : max_points) : 0;
    for (int b = 0; b < batch; ++b) LAV_REQUIRE(h_num_points[b] >= 0, "lav_pillar_decorate: negative num_points");
    a.min_x = grid->min_x; a.max_x = grid->max_x; a.min_y = grid->min_y; a.max_y = grid->max_y; a.ppm = grid->ppm;
    a.nx = grid->nx; a.ny = grid->ny; a.KX = grid->nx + 1; a.KY = grid->ny + 1;
    a.T = a.TW = 0;
    const long ncells = (long)batch * a.KX * a.KY;
    const long total = (long)batch * max_points;
    const unsigned gp = (unsigned)((total + 255) / 256);
    LAV_HIP(hipMemsetAsync(w.cell_count, 0, ncells * sizeof(int), st));
    LAV_HIP(hipMemsetAsync(w.totals, 0, 4 * sizeof(int), st));
    if (total > 0) {
        hipLaunchKernelGGL(k_keys, dim3(gp), dim3(256), 0, st, a, w.key);
        hipLaunchKernelGGL(k_cell_count, dim3(gp), dim3(256), 0, st, total, w.key, w.cell_count);
        LAV_LAUNCH_CHECK();
    }
    int rc = exclusive_scan(w.cell_count, ncells, 1, w.cell_rank, w.block_sums, w.totals + 0, st);
    if (rc) return rc;
    if (total > 0) {
        rc = exclusive_scan(w.key, total, 2, w.kept_rank, w.block_sums, w.totals + 1, st);
        if (rc) return rc;
        const long nsum = (ncells < total ? ncells : total) * 3;
        LAV_HIP(hipMemsetAsync(w.cell_sums, 0, nsum * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_dec_sums, dim3(gp), dim3(256), 0, st, a, w.key, w.cell_rank, w.kept_rank, w.cell_sums, inverse, kept_src);
        hipLaunchKernelGGL(k_dec_write, dim3(gp), dim3(256), 0, st, a, w.key, w.cell_rank, w.cell_count, w.kept_rank, w.cell_sums, decorated);
        LAV_LAUNCH_CHECK();
    }
    if (unique_coords) {
        hipLaunchKernelGGL(k_unique_coords, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, a, w.cell_count, w.cell_rank, ncells, unique_coords);
        LAV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_counts, dim3(1), dim3(1), 0, st, w.totals + 0, w.totals + 1, counts);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_scatter_max(const float *src, const int *index, int n, int channels, int num_segments, float *out, int *argmax,
                               void *stream) {
    LAV_REQUIRE(n >= 0 && channels > 0 && num_segments >= 0, "lav_scatter_max: bad sizes");
    LAV_REQUIRE(out && argmax && (n == 0 || (src && index)), "lav_scatter_max: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long pc = (long)num_segments * channels, nc = (long)n * channels;
    unsigned *out_u = reinterpret_cast<unsigned *>(out);
    if (pc == 0) return LAV_OK;
    const int tok = timer_begin("scatter_max", st);
    hipLaunchKernelGGL(k_smax_init, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, st, out_u, argmax, pc, n);
    if (nc > 0) {
        hipLaunchKernelGGL(k_smax_max, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, src, index, nc, channels, out_u);
        hipLaunchKernelGGL(k_smax_arg, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, src, index, nc, channels, out_u, argmax);
    }
    hipLaunchKernelGGL(k_smax_fin, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, st, out_u, argmax, pc, n);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_scatter_max_backward(const float *grad_out, const int *argmax, int n, int channels, int num_segments,
                                        float *grad_src, void *stream) {
    LAV_REQUIRE(n >= 0 && channels > 0 && num_segments >= 0, "lav_scatter_max_backward: bad sizes");
    LAV_REQUIRE(grad_src || n == 0, "lav_scatter_max_backward: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long pc = (long)num_segments * channels;
    if (n > 0) LAV_HIP(hipMemsetAsync(grad_src, 0, (size_t)n * channels * sizeof(float), st));
    if (pc == 0 || n == 0) return LAV_OK;
    LAV_REQUIRE(grad_out && argmax, "lav_scatter_max_backward: null argument");
    hipLaunchKernelGGL(k_smax_bwd, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, st, grad_out, argmax, pc, channels, n, grad_src);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

// PointPillars dynamic voxelisation -> decoration -> PointNet -> scatter-max -> dense BEV canvas.
//
// Replaces PointPillarNet.forward of the reference (lav/models/point_pillar.py:92-116) including the two
// torch_scatter calls (:33, :62) and coords.unique(dim=0) (:82).  gfx950 only.
//
// Canvas path = TWO launches (v2 agent: N ~ 196k points x 11 floats, 320x320 cells, C = 64):
//
//   k_bin   1 thread / point.  float32 cell id exactly as the reference; the UNIT the cell lands on = one canvas row
//           x 32 columns (one 128-byte line of every channel plane); arrival slot by one integer atomic on the unit's
//           counter (2 sub-counters on different cache lines); the point is written ONCE as a 48-byte record
//           {11 floats, packed cell} into the unit's fixed-capacity bucket, or appended to an overflow list when the
//           bucket is full (adversarial clouds only).  No scan, no second pass over the cloud.
//           Extra workgroups of k_bin rank the point counts that the ownership classes of k_rows held in the PREVIOUS
//           call and write the permutation class <- workgroup that pairs heavy with light classes on a CU (a hint: any
//           permutation gives the same canvas).
//   k_rows  persistent workgroups, two per CU (W = 512).  Ownership class w OWNS units w, w + W, w + 2 W, ...: six or
//           seven units spread evenly over the canvas, so that the dense cells around the ego vehicle are dealt out over
//           the whole chip without any scan, queue or inter-workgroup traffic - a workgroup reads which class it works on
//           (perm[blockIdx]), then 16 counters, and knows its work.  Its units are processed as one GROUP (up to 8 units
//           = 256 tile columns):
//        (0) units without points stream zeros straight from registers;
//        (a) per-cell xyz sums and counts in LDS (64-bit fixed point: order independent, so the canvas is
//            bit-identical for ANY arrival order / input permutation; exact to 2^-32 m); only the tile columns that
//            points touch are then zeroed;
//        (b) JOBS of 16 points, dealt round-robin to the 4 waves, run BOTH PointNet layers on the matrix cores
//            with every weight fragment and all activations in registers (v_mfma_f32_16x16x4_f32, exact fp32):
//              layer 1 (transposed)  D1[c][p]  = sum_k W1[k][c] * F[k][p]     A = weights, B = point features
//                 lane (p = l&15, g = l>>4) supplies feature 4s+g of point p at k-step s; the bias rides as
//                 feature K1 (= 1).  D1 leaves lane (p, g) holding channels 16*ct + 4*g + r  (r = 0..3).
//              layer 2               D2[p][c2] = sum_c H1[p][c] * W2[c][c2]   A = relu(D1) AS IT SITS, B = weights
//                 k-step (ct, r) covers k = 16*ct + 4*g + r over the four lane groups - a permutation of 0..63,
//                 which a sum does not care about - so no shuffles or LDS round trip between the layers;
//            a group of one (two) jobs is split four (two) ways over the waves by output channels;
//        (c) unsigned-integer max of the float bits (values >= 0 after ReLU) into an LDS tile [C][260];
//        (d) the units with points stream from the tile to the NCHW canvas with 16-byte stores (quads of columns no
//            point touched are written as zeros without an LDS read).
//           The canvas is written exactly once and never read or memset: algorithmic traffic 4*(N*D + C*ny*nx) bytes.
//           Weight fragments arrive pre-ordered (k_bin re-lays the 5.6 k floats on every call): 22 coalesced loads per
//           wave.  Barriers order LDS traffic only (s_waitcnt lgkmcnt + s_barrier), so prefetched records and canvas
//           stores stay in flight across them.
//
// Workspace contract: the first bytes of the workspace hold state that is ZERO AT REST (two sets of unit counters
// used alternately, two epoch words).  The caller zero-fills a workspace once (lav_pillar_workspace_init) and keeps it
// for one (batch, grid) geometry; every call leaves it clean for the next one without a memset launch.
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
#include <type_traits>
#include <vector>

#include "common.hpp"

// parity-critical float32 arithmetic: no fused multiply-add contraction anywhere in this file
// (HIP's __fadd_rn/__fmul_rn are plain operators that clang would otherwise fuse)
#pragma clang fp contract(off)

namespace {
using namespace lav;

constexpr int C = 64;            // PointNet width (config num_features [64,64])
constexpr int MAX_BATCH = 64;    // per-call limit on clouds (kernel-argument table)
constexpr int UW = 32;           // canvas columns per unit
constexpr int NSUB = 2;          // arrival counters per unit, on different cache lines
constexpr int GMAX = 8;          // units per group (consecutive units, whatever canvas rows they are on)
constexpr int GW = GMAX * UW;    // canvas columns per group (LDS tile width)
constexpr int TS = GW + 4;       // LDS tile row stride in floats (16-byte aligned rows)
constexpr int CAP_MIN = 16, CAP_MAX = 256;  // records per (unit, sub) bucket
constexpr int MAX_LAYERS = 64;   // regular + overflow layers a group may have (2-4 for square grids)
constexpr double FIX_SCALE = 4294967296.0;  // 2^32 fixed-point scale of the per-cell coordinate sums

// zero at rest (see the workspace contract above)
struct State {
    unsigned epoch_bin;   // written by k_bin, read by k_rows of the same call
    unsigned epoch_rows;  // written by k_rows, read by k_bin of the next call
    int n_ovf[2];         // overflow records appended by k_bin, per counter set
    int pad[60];
};
static_assert(sizeof(State) == 256, "State is 256 bytes");

struct PillarArgs {   // scalars first: they share the first cache line of the kernel-argument segment
    const float *points;
    int batch, max_points, D;
    float min_x, max_x, min_y, max_y, ppm;
    int nx, ny;  // nx = number of xi cells = canvas columns; ny = number of yi cells = canvas rows
    int KX, KY;  // key space (nx+1) x (ny+1)
    int UPR;     // units per canvas row
    int NUP;     // stride of one sub-counter array = number of units rounded up to 4 (16-byte loads)
    int cap;     // records per (unit, sub) bucket
    int nwg;     // workgroups of k_rows (classes of the unit ownership)
    int rstride; // dwords between records (rec_stride_host)
    unsigned long long *trace;  // debug (LAV_PILLAR_TRACE): [workgroups][16] wall-clock stamps of thread 0, else null
    int n[MAX_BATCH];
};

// What k_rows needs of the above: small enough to arrive with the first kernel-argument fetch.
struct RowsArgs {
    int batch, nx, ny, UPR, NUP, cap, rstride;
    float min_x, min_y, ppm;
    const int *perm;   // perm[workgroup] = ownership class it works on (written by k_bin from the previous call's loads)
    int *load;         // load[class] = points the class held in this call (the next call's hint)
    unsigned long long *trace;
    float *amax;       // [workgroups] or null: the largest finite value this workgroup's PointNet produced (lav_pillar_scatter_amax)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int rec_size(int D) { return D + 1 <= 8 ? 8 : 12; }  // dwords per record that carry data; the last one is the packed cell
// dwords from one record to the next.  Round 5: 64-byte records for the 11-float points (LAV_PILLAR_REC=48 restores 48): a 48-byte
// record straddles 64-byte sectors, every bucket write was a partial-sector write (k_bin wrote 1.9x its records' bytes, r04_pmc_pillar).
constexpr int REC_STRIDE_MAX = 16;
inline int rec_stride_host(int D) {
    static const bool wide = [] { const char *e = getenv("LAV_PILLAR_REC"); return !(e && atoi(e) == 48); }();
    return rec_size(D) == 12 && wide ? 16 : rec_size(D);
}

// PointNet weights in the order the matrix instructions of k_rows consume them, one float4 per lane and fragment row:
//   rows 0 .. KS1-1      layer-1 A operand of k-step s, the four 16-channel tiles:  W1[4s + lg][16 ct + lp]  (bias at k = K1)
//   rows KS1 .. KS1+15   layer-2 B operand of k-step (ct, r), the four output tiles:  W2[16 ct + 4 lg + r][16 ct2 + lp]
//   row  KS1+16          layer-2 bias b2[16 ct2 + lp]
// k_bin writes them into the workspace on every call (5.6 k floats), so that every wave of k_rows gets its 88 weight
// registers with 22 coalesced 16-byte loads instead of 88 scattered 4-byte ones.
constexpr int layer1_ksteps(int D) { return (D + 5 + 4) / 4; }
constexpr int packed_weight_floats(int D) { return (layer1_ksteps(D) + 17) * 64 * 4; }

// cell key of a point, or -1 (grid_locations, point_pillar.py:70-79)
__device__ __forceinline__ int cell_key(const PillarArgs &a, int b, float x, float y) {
    // NaN fails every comparison and is dropped, as in torch
    if (!(x >= a.min_x && x < a.max_x && y >= a.min_y && y < a.max_y)) return -1;
    const float fx = (x - a.min_x) * a.ppm;  // two roundings (contraction is off)
    const float fy = (y - a.min_y) * a.ppm;
    const int xi = (int)fx, yi = (int)fy;    // truncation; both are in [0, nx] x [0, ny]
    return (b * a.KX + xi) * a.KY + yi;
}

// unit that cell (b, xi, yi) lands on (scatter_points clamp, point_pillar.py:89)
__device__ __forceinline__ int unit_of(const PillarArgs &a, int b, int xi, int yi) {
    const int r = min(max(a.ny - 1 - xi, 0), a.ny - 1);
    const int col = min(yi, a.nx - 1);
    return (b * a.ny + r) * a.UPR + col / UW;
}

// ---------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_bin(PillarArgs a, State *__restrict__ st, int *__restrict__ counters,
                                             float *__restrict__ buckets, float *__restrict__ ovf, int *__restrict__ key_out,
                                             const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                                             const float *__restrict__ b2, float *__restrict__ wpack, const int *__restrict__ load,
                                             int *__restrict__ perm, float *__restrict__ canvas, long canvas_floats, int nzero) {
    constexpr int RS = rec_size(D);
    const int nbin = max(1, (int)(((long)a.batch * a.max_points + 255) / 256));   // workgroups that bin points
    const int nrank = (a.nwg + 63) / 64;                                           // workgroups that rank the ownership classes
    if ((int)blockIdx.x >= nbin + nrank) {
        // Round 5: the LAST `nzero` workgroups stream zeros over the whole canvas while the others bin (binning is a chain of
        // latencies - launch, points, one returning atomic, record stores - that leaves HBM idle: the 26 MB of zeros cost this kernel
        // ~1 us and take 60 % of the bytes out of k_rows, which then writes only the 16-byte quads that points touched).  The kernel
        // boundary orders these stores before k_rows' (same stream): write-after-write needs nothing else.
        const long z = (long)((int)blockIdx.x - nbin - nrank) * 256 + threadIdx.x, stride = (long)nzero * 256;
        if ((reinterpret_cast<uintptr_t>(canvas) & 15) == 0) {
            f32x4 *c4 = reinterpret_cast<f32x4 *>(canvas);
            const long n4 = canvas_floats >> 2;
            for (long i = z; i < n4; i += stride) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, c4 + i);
            for (long i = (n4 << 2) + z; i < canvas_floats; i += stride) canvas[i] = 0.f;
        } else {
            for (long i = z; i < canvas_floats; i += stride) canvas[i] = 0.f;
        }
        return;
    }
    if ((int)blockIdx.x >= nbin) {
        // Extra workgroups (past the ones that bin points): which ownership class each workgroup of k_rows takes.  The point
        // count of a class barely changes from one LiDAR frame to the next, so the loads k_rows recorded in the PREVIOUS call
        // are this call's hint: classes ranked by load, the heaviest handed to workgroups 0, 1, ... and the lightest to their
        // CU partners (workgroup w + W/2 lands on the CU of workgroup w), so that the two workgroups of a CU add up to about
        // the same.  Any permutation is valid - the hint moves time around, never results.  Each of these workgroups ranks
        // 64 classes, four lanes per class.
        __shared__ int s_key[2048];
        const int W = a.nwg;
        const int first = ((int)blockIdx.x - nbin) * 64;
        for (int i = threadIdx.x; i < W; i += 256) s_key[i] = (min(load[i], 0xfffff) << 11) | (2047 - i);   // distinct keys
        __syncthreads();
        const int i = first + (int)(threadIdx.x >> 2), q = threadIdx.x & 3;
        if (i < W) {
            const int ki = s_key[i];
            int r = 0;
            for (int j = q; j < W; j += 4) r += s_key[j] > ki;
            r += __shfl_xor(r, 1, 64);
            r += __shfl_xor(r, 2, 64);
            const int half = W / 2;
            const int slot = (W & 1) ? r : (r < half ? r : half + (W - 1 - r));
            if (q == 0) perm[slot] = i;
        }
        return;
    }
    // fetch both cache lines of the kernel-argument segment at once (taken in turn, each is a microsecond-scale miss)
    asm volatile("" ::"s"(a.max_points), "s"(key_out));
    const long total = (long)a.batch * a.max_points;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned e = st->epoch_rows;
    if (gid == 0) {
        st->epoch_bin = e;
        st->n_ovf[(e & 1u) ^ 1u] = 0;
    }
    {   // The counter set of the previous call was consumed by its k_rows: clean it here, so that k_rows can read "both sets
        // added up" without first waiting for the epoch word to learn which one is live.
        int *cz = counters + (size_t)((e & 1u) ^ 1u) * NSUB * a.NUP;
        for (long i = gid; i < (long)NSUB * a.NUP; i += (long)nbin * 256) cz[i] = 0;
    }
    {   // PointNet weights in fragment order (see packed_weight_floats)
        constexpr int K1 = D + 5, KS1 = layer1_ksteps(D);
        for (long i = gid; i < packed_weight_floats(D); i += (long)nbin * 256) {
            const int c4 = (int)i & 3, l = ((int)i >> 2) & 63, row = (int)i >> 8, lp = l & 15, lg = l >> 4;
            float v;
            if (row < KS1) {
                const int k = 4 * row + lg, c = 16 * c4 + lp;
                v = k < K1 ? w1[k * C + c] : (k == K1 ? b1[c] : 0.f);
            } else if (row < KS1 + 16) {
                const int ct = (row - KS1) >> 2, r = (row - KS1) & 3;
                v = w2[(16 * ct + 4 * lg + r) * C + 16 * c4 + lp];
            } else {
                v = b2[16 * c4 + lp];
            }
            wpack[i] = v;
        }
    }
    if (gid >= total) return;
    const int b = (int)(gid / a.max_points);
    const int i = (int)(gid - (long)b * a.max_points);
    int k = -1;
    const float *pt = a.points + gid * D;
    float v[RS];
    // the whole row is requested at once (one memory round trip before the arrival atomic, not two)
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = pt[d];
    if (i < a.n[b]) k = cell_key(a, b, v[0], v[1]);
    if (key_out) key_out[gid] = k;
    if (k < 0) return;
    const int cellk = k - b * a.KX * a.KY;
    const int xi = cellk / a.KY, yi = cellk - xi * a.KY;
    const int unit = unit_of(a, b, xi, yi);
    const int set = (int)(e & 1u);
    // the sub-counter only decorrelates concurrent arrivals
    const int sub = (threadIdx.x ^ (threadIdx.x >> 6) ^ blockIdx.x) & (NSUB - 1);
    const int slot = atomicAdd(&counters[((size_t)set * NSUB + sub) * a.NUP + unit], 1);
#pragma unroll
    for (int d = D; d < RS - 1; ++d) v[d] = 0.f;
    v[RS - 1] = __int_as_float((b << 24) | (xi << 12) | yi);
    float *o;
    if (slot < a.cap) {
        o = buckets + (((size_t)unit * NSUB + sub) * a.cap + slot) * a.rstride;
    } else {
        o = ovf + (size_t)atomicAdd(&st->n_ovf[set], 1) * a.rstride;
    }
#pragma unroll
    for (int q = 0; q < RS / 4; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (a.rstride > RS) reinterpret_cast<float4 *>(o)[RS / 4] = make_float4(0.f, 0.f, 0.f, 0.f);   // the whole 64-byte sector is written
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), which would serialise the record prefetches and the canvas stores of
// k_rows behind each barrier; the waves of k_rows exchange data through LDS only.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum inside every row of 16 lanes (DPP row shifts: no LDS round trip, unlike __shfl_up).
__device__ __forceinline__ int row_scan16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1, lanes shifted in read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    return v;
}
// Inclusive prefix sum over the 64 lanes of a wave.
__device__ __forceinline__ int wave_scan64(int v) {
    v = row_scan16(v);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (threadIdx.x & 63) >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// ---------------------------------------------------------------------------------------------------------
// One group of units of a workgroup, as every thread of the workgroup sees it (all values workgroup-uniform).
// Workgroup w of W owns units w, w + W, w + 2 W, ... (its "slots" 0, 1, 2, ...); a group is up to GMAX consecutive slots.
struct Group {
    int first, gn;  // slots [first, first + gn)
    int n;          // records in the buckets of the group
    bool has_ovf;   // some bucket of the group overflowed into the overflow list
    int slot;       // which of the wave's two prefix tables (gpre) describes its buckets
    unsigned mask;  // bit NSUB*k + s: bucket (unit slot first + k, sub s) holds records
};
constexpr int NBKT = GMAX * NSUB;  // buckets of a group

// ZB: the canvas arrives ZERO-FILLED (k_bin's zero workgroups): nothing is written but the quads of columns that points touched.
template <int D, bool USE_MFMA, bool VEC4, bool TRACE = false, bool ZB = false>
__global__ __launch_bounds__(256, 2) void k_rows(RowsArgs a, State *__restrict__ st, int *__restrict__ counters,
                                                 const float *__restrict__ buckets, const float *__restrict__ ovf,
                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                 const float *__restrict__ wpack, float *__restrict__ canvas) {
    constexpr int K1 = D + 5;              // decorated features
    constexpr int KS1 = layer1_ksteps(D);  // layer-1 k-steps of 4 incl. the bias feature (K1 = 16 -> 5)
    constexpr int RS = rec_size(D), RQ = RS / 4;
    __shared__ __attribute__((aligned(16))) float tile[C * TS];  // [C][TS]
    __shared__ unsigned long long sums[GW * 3];                  // [col][3]
    __shared__ int cnt[GW];
    __shared__ int nl[MAX_LAYERS];
    __shared__ int occ4[GW / 4];   // group number (+1) of the last group that put a point on this quad of columns
    __shared__ int newq[GW / 4];   // quads a layer's sweep touched for the first time in this group: their tile columns get zeroed
    __shared__ __attribute__((aligned(16))) int gpre[4][2][NBKT + 4];  // per wave, two slots: first record index of every bucket of a group
    __shared__ float s_wmax[4];
    float wmax = 0.f;   // largest finite PointNet output of this lane (round 6: the canvas' bound for the first BEV layer's fp16 scale)

    // fetch both cache lines of the kernel-argument segment at once (taken in turn, each is a microsecond-scale miss)
    asm volatile("" ::"s"(a.batch), "s"(canvas));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lp = lane & 15, lg = lane >> 4;
#define LAV_STAMP(i) do { if constexpr (TRACE) { if (tid == 0) a.trace[(long)blockIdx.x * 16 + (i)] = wall_clock64(); } } while (0)
    LAV_STAMP(0);
    long long cyc0 = 0;
    if constexpr (TRACE) cyc0 = clock64();
    const int nunits = a.batch * a.ny * a.UPR;
    const int W = (int)gridDim.x;
    const int w = a.perm[blockIdx.x];              // the ownership class this workgroup works on (k_bin's pairing, see there)
    const int nslots = (nunits - w + W - 1) / W;   // units of the class
    if (nslots <= 0) {
        if (tid == 0) {
            a.load[w] = 0;
            if (a.amax) a.amax[blockIdx.x] = 0.f;
        }
        return;
    }
    int load_sum = 0;
    // Two counter sets are used by alternate calls and k_bin has cleaned the one it did not fill: every counter is read
    // as the sum over both sets, which needs no knowledge of the epoch.
    const int *__restrict__ cn = counters;
    // bucket counts of the group that starts at slot `first`: lane l < NBKT holds bucket (unit slot first + l / NSUB, sub
    // l % NSUB).  Unconditional load from a clamped address (a load under an exec-mask branch is waited for inside the branch).
    auto fetch_counts = [&](int first) {
        const int k = first + lane / NSUB, sb = lane % NSUB;
        const int u = min(w + k * W, nunits - 1);
        const int c = cn[sb * a.NUP + u] + cn[(NSUB + sb) * a.NUP + u];
        return (lane < NBKT && k < nslots) ? c : 0;
    };
    const int c_first = fetch_counts(0);

    if (blockIdx.x == 0 && tid == 0) st->epoch_rows = st->epoch_bin + 1u;   // the next call fills the other counter set

    // weight fragments, once per workgroup (fragment order written by k_bin: 22 coalesced loads)
    float a1[4][KS1], w2f[4][4][4], b2v[4];
    if constexpr (USE_MFMA) {
        const float4 *wp = reinterpret_cast<const float4 *>(wpack) + lane;
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const float4 t = wp[s * 64];
            a1[0][s] = t.x; a1[1][s] = t.y; a1[2][s] = t.z; a1[3][s] = t.w;
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 t = wp[(KS1 + 4 * ct + r) * 64];
                w2f[ct][r][0] = t.x; w2f[ct][r][1] = t.y; w2f[ct][r][2] = t.z; w2f[ct][r][3] = t.w;
            }
        const float4 t = wp[(KS1 + 16) * 64];
        b2v[0] = t.x; b2v[1] = t.y; b2v[2] = t.z; b2v[3] = t.w;
    }
    // LDS state every group starts from: sums, counts, layer counts and markers zero.  The tile itself is NOT cleared:
    // only the quads of columns that points touch are zeroed (after the sums sweep) and read back.
    for (int i = tid; i < GW * 3; i += 256) sums[i] = 0ull;
    if (tid < GW) cnt[tid] = 0;
    if (tid < MAX_LAYERS) nl[tid] = 0;
    if (tid < GW / 4) { occ4[tid] = 0; newq[tid] = 0; }
    LAV_STAMP(1);

    // Every wave keeps its own copy of a group's bucket prefix in LDS (written and read by the same wave: the LDS
    // queue of a wave is in order, so no barrier is involved and groups without points need none either).
    auto describe = [&](int first, int c, int slot) {
        Group g;
        g.first = first;
        g.gn = min(GMAX, nslots - first);
        g.slot = slot;
        g.has_ovf = __ballot(c > a.cap) != 0ull;
        g.mask = (unsigned)__ballot(c > 0);
        const int cc = min(c, a.cap);
        const int inc = row_scan16(cc);
        static_assert(NBKT == 16, "one DPP row of 16 lanes scans the buckets of a group");
        if (lane < NBKT) gpre[wid][slot][lane] = inc - cc;   // exclusive
        if (lane == NBKT - 1) gpre[wid][slot][NBKT] = inc;   // total
        g.n = __builtin_amdgcn_readlane(inc, NBKT - 1);
        return g;
    };
    // address of record i (< g.n) of a group
    auto rec_ptr = [&](const Group &g, int i) {
        const int4 *pp = reinterpret_cast<const int4 *>(gpre[wid][g.slot]);
        const int4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
        const int q = (i >= p0.y) + (i >= p0.z) + (i >= p0.w) + (i >= p1.x) + (i >= p1.y) + (i >= p1.z) + (i >= p1.w) + (i >= p2.x) +
                      (i >= p2.y) + (i >= p2.z) + (i >= p2.w) + (i >= p3.x) + (i >= p3.y) + (i >= p3.z) + (i >= p3.w);
        const size_t unit = (size_t)w + (size_t)(g.first + q / NSUB) * W;
        return buckets + ((unit * NSUB + q % NSUB) * a.cap + (i - gpre[wid][g.slot][q])) * a.rstride;
    };
    auto load_rec = [&](const float *p, float4 (&r)[RQ]) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) r[q] = reinterpret_cast<const float4 *>(p)[q];
    };

    // The next job record of this wave: the next job of the current group, or - requested while the last job of a
    // group is on the matrix cores - the wave's first job of the following group.
    float4 nxt[RQ];
    // A group of one job is split four ways over the waves (each runs layer 1 and one 16-channel slice of layer 2:
    // 36 instead of 84 matrix instructions in the group's critical path), a group of two jobs two ways.
    auto split_shift = [&](const Group &gg) {
        const int ja = (gg.n + 15) >> 4;
        return gg.has_ovf ? 0 : (ja == 1 ? 2 : (ja == 2 ? 1 : 0));
    };
    auto request_first_job = [&](const Group &gg, int rot_) {
        if (gg.n > 0) load_rec(rec_ptr(gg, min(16 * (((wid - rot_) & 3) >> split_shift(gg)) + lp, gg.n - 1)), nxt);
    };

    int gi = 0;
    Group g = describe(0, c_first, 0);
    request_first_job(g, 0);
    int c_next = nslots > GMAX ? fetch_counts(GMAX) : 0;   // counters of the following group, one group ahead
    const long cstride = (long)a.ny * a.nx;
    const int xi_row0 = max(a.ny - 1, 0);                       // first key row that lands on canvas row 0
    const int nlay_over = max(0, a.ny - a.nx + 1) + 1;          // layers of a unit that holds the last canvas column
    const int nlay_max = min(MAX_LAYERS, max(a.nx - xi_row0 + 1, 1) * nlay_over);
    // walking this workgroup's units: one division here, carries afterwards
    const int step_rb = W / a.UPR, step_cu = W - step_rb * a.UPR, step_b = step_rb / a.ny, step_r = step_rb - step_b * a.ny;
    barrier_lds();   // LDS state set up

    while (true) {
        const int rot = gi & 3;
        const int gn = g.gn;   // units of the group; the tile column of (unit slot k, column c of the unit) is 32 k + c
        // what comes after this group (workgroup-uniform); its records are requested while this one computes
        const bool more = g.first + gn < nslots;
        if (gi == 0) LAV_STAMP(3);
        if (gi == 1) LAV_STAMP(11);
        if constexpr (TRACE) { if (tid == 0) a.trace[(long)blockIdx.x * 16 + 15] += (unsigned long long)g.n; }
        load_sum += g.n;
        bool prefetched = false;
        Group gnext;
        auto prefetch_next = [&]() {
            if (more && !prefetched) {
                gnext = describe(g.first + gn, c_next, g.slot ^ 1);
                request_first_job(gnext, (gi + 1) & 3);
            }
            prefetched = true;
        };
        // Streams units of the group to the canvas [B][C][ny][nx]: zeros from registers (FROM_LDS false) for the units
        // without points, the tile for the others.  A unit is one 128-byte line of every channel plane (less at the right
        // border).
        auto store_units = [&](auto FROM_LDS_) {
            constexpr bool FROM_LDS = decltype(FROM_LDS_)::value;
            if constexpr (ZB && !FROM_LDS) return;   // the zeros are already there
            const int un0 = w + g.first * W;
            const int rb0 = un0 / a.UPR;
            int cu = un0 - rb0 * a.UPR, b = rb0 / a.ny, r = rb0 - b * a.ny;
            const int j4 = tid & 7, ch0 = tid >> 3;
            int flags = 0;   // bit k: quad (unit k, j4) holds data in the tile
            if constexpr (FROM_LDS && VEC4) {
                int f[GMAX];
#pragma unroll
                for (int k = 0; k < GMAX; ++k) f[k] = occ4[k * (UW / 4) + j4];
#pragma unroll
                for (int k = 0; k < GMAX; ++k) flags |= (f[k] == gi + 1) << k;
            }
            for (int k = 0; k < gn; ++k) {
                const bool has = ((g.mask >> (NSUB * k)) & ((1u << NSUB) - 1)) != 0;
                if (has == FROM_LDS) {
                    const int c0 = cu * UW, wd = min(a.nx, c0 + UW) - c0;
                    float *dst = canvas + ((long)b * C * a.ny + r) * a.nx + c0;
                    if constexpr (VEC4) {
                        float4 val[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                        if constexpr (FROM_LDS) {
                            if ((flags >> k) & 1) {   // quads no point of this group touched hold garbage in LDS and zeros on the canvas
#pragma unroll
                                for (int h = 0; h < 2; ++h) val[h] = *reinterpret_cast<const float4 *>(tile + (ch0 + 32 * h) * TS + k * UW + 4 * j4);
                            }
                        }
                        if (4 * j4 < wd && (!ZB || ((flags >> k) & 1))) {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                __builtin_nontemporal_store(f32x4{val[h].x, val[h].y, val[h].z, val[h].w},
                                                            reinterpret_cast<f32x4 *>(dst + (ch0 + 32 * h) * cstride + 4 * j4));
                        }
                    } else {
                        for (int idx = tid; idx < C * wd; idx += 256) {
                            const int ch = idx / wd, j = idx - ch * wd;
                            float val = 0.f;
                            bool touched = false;
                            if constexpr (FROM_LDS) {
                                touched = occ4[(k * UW + j) >> 2] == gi + 1;
                                if (touched) val = tile[ch * TS + k * UW + j];
                            }
                            if (!ZB || touched) dst[ch * cstride + j] = val;
                        }
                    }
                }
                cu += step_cu;
                r += step_r + (cu >= a.UPR);
                b += step_b;
                if (cu >= a.UPR) cu -= a.UPR;
                if (r >= a.ny) { r -= a.ny; ++b; }
            }
        };
        using std::false_type;
        using std::true_type;

        if (g.n == 0 && !g.has_ovf) {
            prefetch_next();
            store_units(false_type{});   // no point lands here: zeros straight from registers
        } else {
            // Units of the group without points: a store only leaves the wave once the memory pipeline accepts it, and the
            // whole chip is streaming the canvas, so these zeros cost their share of the HBM time wherever they are put.  The
            // two workgroups of a CU put them at opposite ends - before the sums sweep (while the records are in flight) or
            // after the PointNet - so that one's stores run beside the other's latency-bound compute.
            // (workgroup b goes to XCD b % 8 and, with two workgroups per CU, shares its CU with workgroup b + gridDim/2)
            const bool zeros_first = blockIdx.x >= gridDim.x / 2;
            if (zeros_first) store_units(false_type{});
            // packed cell -> (layer, tile column, xi, yi, member of this group).  Layers order the pillars that the
            // reference's clamp (:89) sends to one canvas cell like its sorted unique rows: later layers replace earlier.
            auto classify = [&](int packed, int &layer, int &col, int &xi, int &yi) {
                xi = (packed >> 12) & 0xfff;
                yi = packed & 0xfff;
                const int r_ = min(max(a.ny - 1 - xi, 0), a.ny - 1);
                const int colc = min(yi, a.nx - 1), cu_ = colc / UW;
                // units of this workgroup are w, w + W, w + 2 W, ...: slot of the record's unit inside the group
                const int un = ((packed >> 24) * a.ny + r_) * a.UPR + cu_ - w;
                const int ks = un / (int)gridDim.x;
                const int k = ks - g.first;
                col = k * UW + (colc - cu_ * UW);
                const int over = yi >= a.nx ? yi - a.nx + 1 : 0;
                layer = (r_ > 0 ? 0 : xi - xi_row0) * (cu_ == a.UPR - 1 ? nlay_over : 1) + over;
                return un >= 0 && ks * (int)gridDim.x == un && (unsigned)k < (unsigned)gn;
            };
            const int n_ovf = g.has_ovf ? st->n_ovf[0] + st->n_ovf[1] : 0;   // (one of the two is zero, like the counters)
            const int JA = (g.n + 15) >> 4;
            const int JB = (n_ovf + 15) >> 4;
            const int J = JA + JB;
            const int sh = split_shift(g), VJ = J << sh;   // virtual jobs = (job, slice of the output channels)
            const int v0 = (wid - rot) & 3;                // this wave's first virtual job

            for (int L = 0; L < nlay_max; ++L) {
                if (L > 0) {
                    if (nl[L] == 0) continue;  // workgroup-uniform; nl[] is complete after layer 0's sums sweep
                    barrier_lds();
                    for (int j = tid; j < GW * 3; j += 256) sums[j] = 0ull;
                    if (tid < GW) cnt[tid] = 0;
                    barrier_lds();
                }
                // (a) per-cell coordinate sums and counts of this layer.  `part` = which of the four atomics of a point
                // this lane issues (0-2 = x, y, z sums, 3 = count and markers), or -1 = all of them.
                auto add_point = [&](float x, float y, float z, int packed, bool member_known, int part) {
                    int layer, col, xi, yi;
                    const bool member = classify(packed, layer, col, xi, yi);
                    if (!member_known && !member) return;
                    if ((part < 0 || part == 3) && L == 0 && layer != 0) atomicAdd(&nl[min(layer, MAX_LAYERS - 1)], 1);
                    if (layer == L) {
                        const float xyz[3] = {x, y, z};
#pragma unroll
                        for (int d = 0; d < 3; ++d)
                            if (part < 0 || part == d) {
                                const long long q = __double2ll_rn((double)xyz[d] * FIX_SCALE);
                                atomicAdd(&sums[col * 3 + d], (unsigned long long)q);
                            }
                        if (part < 0 || part == 3) {
                            atomicAdd(&cnt[col], 1);
                            if (occ4[col >> 2] != gi + 1) {   // (benign race: every racer writes the same values)
                                occ4[col >> 2] = gi + 1;
                                newq[col >> 2] = 1;
                            }
                        }
                    }
                };
                // the first 64 records are the waves' first jobs, already requested: the four lane groups hold one copy each
                // and share the point's four atomics
                if (L == 0 && (v0 & ((1 << sh) - 1)) == 0 && 16 * (v0 >> sh) + lp < g.n)
                    add_point(nxt[0].x, nxt[0].y, nxt[0].z, __float_as_int(nxt[RQ - 1].w), true, lg);
                for (int i = (L == 0 ? 64 : 0) + tid; i < g.n; i += 256) {
                    const float *p = rec_ptr(g, i);
                    const float4 s0 = reinterpret_cast<const float4 *>(p)[0];
                    const float4 s1 = reinterpret_cast<const float4 *>(p)[RQ - 1];
                    add_point(s0.x, s0.y, s0.z, __float_as_int(s1.w), true, -1);
                }
                for (int i = tid; i < n_ovf; i += 256) {
                    const float *p = ovf + (size_t)i * a.rstride;
                    const float4 s0 = reinterpret_cast<const float4 *>(p)[0];
                    const float4 s1 = reinterpret_cast<const float4 *>(p)[RQ - 1];
                    add_point(s0.x, s0.y, s0.z, __float_as_int(s1.w), false, -1);
                }
                if (gi == 0 && L == 0) LAV_STAMP(6);
                barrier_lds();
                if (gi == 0 && L == 0) LAV_STAMP(4);
                {   // zero the tile columns of the quads this sweep touched first (wave w takes quads w, w + 4, ...: one 16-byte
                    // store per channel), and - layers after the first - the cells a later pillar replaces
                    const unsigned long long fresh = __ballot(lane < GW / 16 && newq[wid + 4 * lane] != 0);
                    for (unsigned long long m = fresh; m; m &= m - 1) {
                        const int q = wid + 4 * (__ffsll((long long)m) - 1);
                        *reinterpret_cast<float4 *>(tile + lane * TS + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (lane < GW / 16) newq[wid + 4 * lane] = 0;
                    if (L > 0) {
                        barrier_lds();
                        for (int i = tid; i < C * GW; i += 256) {
                            const int ch = i / GW, j = i - ch * GW;
                            if (cnt[j] > 0 && occ4[j >> 2] == gi + 1) tile[ch * TS + j] = 0.f;
                        }
                    }
                    barrier_lds();
                }
                bool last_layer = true;
                for (int L2 = L + 1; L2 < nlay_max; ++L2) last_layer = last_layer && nl[L2] == 0;

                // decorated features of one record (decorate(), point_pillar.py:55-68)
                auto decorated = [&](const float (&v)[RS], int col, int xi, int yi, float (&f)[K1]) {
                    float mean[3];
                    const int n = cnt[col];
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        mean[d] = (float)((double)(long long)sums[col * 3 + d] / ((double)n * FIX_SCALE));
#pragma unroll
                    for (int d = 0; d < D; ++d) f[d] = v[d];
#pragma unroll
                    for (int d = 0; d < 3; ++d) f[D + d] = v[d] - mean[d];
                    // reference decorate(): x - (yi/ppm + min_x), y - (xi/ppm + min_y)  (sic: swapped, un-centred; :57-58)
                    f[D + 3] = v[0] - ((float)yi / a.ppm + a.min_x);
                    f[D + 4] = v[1] - ((float)xi / a.ppm + a.min_y);
                };

                if constexpr (!USE_MFMA) {
                    // cross-check path: one point per thread, plain fp32 FMAs (slow; selected by LAV_PILLAR_IMPL=valu)
                    if (last_layer) prefetch_next();
                    for (int i = tid; i < g.n + n_ovf; i += 256) {
                        float4 rq[RQ];
                        load_rec(i < g.n ? rec_ptr(g, i) : ovf + (size_t)(i - g.n) * a.rstride, rq);
                        float v[RS];
#pragma unroll
                        for (int q = 0; q < RQ; ++q) { v[4 * q] = rq[q].x; v[4 * q + 1] = rq[q].y; v[4 * q + 2] = rq[q].z; v[4 * q + 3] = rq[q].w; }
                        int layer, col, xi, yi;
                        const bool member = classify(__float_as_int(v[RS - 1]), layer, col, xi, yi);
                        if (!member || layer != L) continue;
                        float f[K1];
                        decorated(v, col, xi, yi, f);
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
                            wmax = fmaxf(wmax, o <= 3.4028235e38f ? o : 0.f);
                            atomicMax(reinterpret_cast<unsigned *>(&tile[c * TS + col]), __float_as_uint(o));
                        }
                    }
                } else {
                    // (b) jobs of 16 records; virtual job v runs on wave (v + rot) & 3
                    auto job_ptr = [&](int j) {
                        return j < JA ? rec_ptr(g, min(16 * j + lp, g.n - 1)) : ovf + (size_t)min(16 * (j - JA) + lp, n_ovf - 1) * a.rstride;
                    };
                    int v = v0;
                    float4 cur[RQ];
                    if (v < VJ) {
                        if ((v >> sh) < JA && L == 0) {
#pragma unroll
                            for (int q = 0; q < RQ; ++q) cur[q] = nxt[q];
                        } else {
                            load_rec(job_ptr(v >> sh), cur);
                        }
                    } else if (last_layer) {
                        prefetch_next();
                    }
                    for (; v < VJ; v += 4) {
                        const int j = v >> sh, part = v & ((1 << sh) - 1);
                        float v_[RS];
#pragma unroll
                        for (int q = 0; q < RQ; ++q) { v_[4 * q] = cur[q].x; v_[4 * q + 1] = cur[q].y; v_[4 * q + 2] = cur[q].z; v_[4 * q + 3] = cur[q].w; }
                        // The next record of this wave (next job of the group, or first job of the next group) is requested
                        // NOW: a wave is stuck issuing its 84 matrix instructions for their whole duration, so a request placed
                        // behind them leaves the L2 round trip exposed at the top of the next job (measured: 3.9 us per job).
                        if (v + 4 < VJ) load_rec(job_ptr((v + 4) >> sh), nxt);
                        else if (last_layer) prefetch_next();
                        const int i = j < JA ? 16 * j + lp : 16 * (j - JA) + lp;
                        bool live = i < (j < JA ? g.n : n_ovf);
                        int layer, col, xi, yi;
                        const bool member = classify(__float_as_int(v_[RS - 1]), layer, col, xi, yi);
                        live = live && member && layer == L;
                        if (__ballot(live) != 0ull) {
                            const int mycol = live ? col : -1;
                            float f[K1];
                            decorated(v_, live ? col : 0, xi, yi, f);
                            float fe[KS1];
#pragma unroll
                            for (int s = 0; s < KS1; ++s) {
                                float t[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int k = 4 * s + q;
                                    t[q] = k < K1 ? f[k < K1 ? k : 0] : (k == K1 ? 1.f : 0.f);
                                }
                                const float lo = (lg & 1) ? t[1] : t[0], hi = (lg & 1) ? t[3] : t[2];
                                fe[s] = live ? ((lg & 2) ? hi : lo) : 0.f;
                            }
                            // layer 1
                            f32x4 d1[4];
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) d1[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int s = 0; s < KS1; ++s)
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct) d1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ct][s], fe[s], d1[ct], 0, 0, 0);
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                                for (int q = 0; q < 4; ++q) d1[ct][q] = d1[ct][q] > 0.f ? d1[ct][q] : 0.f;
                            if (gi == 0 && v == v0) LAV_STAMP(7);
                            // the tile columns of the 4 points this lane gets results for come from the lanes that loaded them
                            int cols[4];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) cols[rr] = __shfl(mycol, 4 * lg + rr, 64);
                            // layer 2 for output channels 16*LO .. 16*(LO+CNT): this lane ends up with channel 16*ct2 + lp of
                            // points 4*lg + r; every accumulator is one chain in the same k order whatever the split, so a
                            // point's features do not depend on how many neighbours its group has.  (c) max into the tile.
                            auto layer2 = [&](auto LO_, auto CNT_) {
                                constexpr int LO = decltype(LO_)::value, CNT = decltype(CNT_)::value;
                                f32x4 d2[CNT];
#pragma unroll
                                for (int c = 0; c < CNT; ++c) d2[c] = f32x4{b2v[LO + c], b2v[LO + c], b2v[LO + c], b2v[LO + c]};
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                                        for (int c = 0; c < CNT; ++c)
                                            d2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[ct][rr], w2f[ct][rr][LO + c], d2[c], 0, 0, 0);
#pragma unroll
                                for (int c = 0; c < CNT; ++c) {
                                    unsigned *trow = reinterpret_cast<unsigned *>(tile + (16 * (LO + c) + lp) * TS);
#pragma unroll
                                    for (int rr = 0; rr < 4; ++rr) {
                                        const float o = d2[c][rr] > 0.f ? d2[c][rr] : 0.f;  // also maps -0 and NaN to +0
                                        if (cols[rr] >= 0) {
                                            wmax = fmaxf(wmax, o <= 3.4028235e38f ? o : 0.f);
                                            atomicMax(trow + cols[rr], __float_as_uint(o));
                                        }
                                    }
                                }
                            };
                            using std::integral_constant;
                            if (sh == 0) {
                                layer2(integral_constant<int, 0>{}, integral_constant<int, 4>{});
                            } else if (sh == 1) {
                                if (part == 0) layer2(integral_constant<int, 0>{}, integral_constant<int, 2>{});
                                else layer2(integral_constant<int, 2>{}, integral_constant<int, 2>{});
                            } else {
                                if (part == 0) layer2(integral_constant<int, 0>{}, integral_constant<int, 1>{});
                                else if (part == 1) layer2(integral_constant<int, 1>{}, integral_constant<int, 1>{});
                                else if (part == 2) layer2(integral_constant<int, 2>{}, integral_constant<int, 1>{});
                                else layer2(integral_constant<int, 3>{}, integral_constant<int, 1>{});
                            }
                            if (gi == 0 && v == v0) LAV_STAMP(8);
                        }
#pragma unroll
                        for (int q = 0; q < RQ; ++q) cur[q] = nxt[q];
                    }
                }
            }
            prefetch_next();   // (no-op when a wave already did it)
            barrier_lds();
            if (gi == 0) LAV_STAMP(9);
            // (d) stream the units that hold points out of the tile
            store_units(true_type{});
            if (!zeros_first) store_units(false_type{});
            for (int i = tid; i < GW * 3; i += 256) sums[i] = 0ull;
            if (tid < GW) cnt[tid] = 0;
            if (tid < MAX_LAYERS) nl[tid] = 0;
            barrier_lds();
        }

        if (gi == 0) LAV_STAMP(10);
        if (!more) break;
        ++gi;
        g = gnext;
        c_next = g.first + g.gn < nslots ? fetch_counts(g.first + g.gn) : 0;
    }
    if (tid == 0) a.load[w] = load_sum;   // next call's pairing hint
    if (a.amax) {   // (workgroup-uniform) a bound of what this workgroup wrote: replaced pillars of the clamp layers count too
        wmax = wave_finite_absmax(wmax);
        if (lane == 0) s_wmax[wid] = wmax;
        barrier_lds();
        if (tid == 0) a.amax[blockIdx.x] = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
    }
    LAV_STAMP(12);
    if constexpr (TRACE) {
        if (tid == 0) {
            a.trace[(long)blockIdx.x * 16 + 13] = (unsigned long long)(clock64() - cyc0);
            // HW_REG_XCC_ID (20) and HW_REG_HW_ID (4): where this workgroup ran
            a.trace[(long)blockIdx.x * 16 + 14] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
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


struct Workspace {
    State *state;      // zero at rest
    int *counters;     // [2 sets][NSUB][nunits rounded up to 4], zero at rest
    int *key;
    float *buckets;    // [nunits][NSUB][cap] records
    float *ovf;        // [batch * max_points] records (overflow list)
    float *wpack;      // PointNet weights in fragment order, rewritten by every call
    int *load, *perm;  // pairing hint of k_rows' workgroups: loads of the previous call, permutation of this one
    int *cell_count, *cell_rank, *kept_rank, *block_sums, *totals;
    unsigned long long *cell_sums;  // [min(ncells, points)][3] fixed-point coordinate sums (training entry lav_pillar_decorate)
    int cap, upr;
};

int bucket_capacity(int max_points, int units_per_cloud) {
    // room for 8x the mean load of a (unit, sub) bucket: the densest unit of a LiDAR-like cloud (the ego vehicle's
    // surroundings) holds about 20x the mean of a unit, spread over NSUB buckets
    const long want = 8l * max_points / ((long)units_per_cloud * NSUB) + 1;
    int cap = CAP_MIN;
    while (cap < want && cap < CAP_MAX) cap *= 2;
    return cap;
}

size_t carve(Arena &ar, Workspace &w, int batch, int max_points, const lav_grid *g, bool with_buckets) {
    const int upr = (g->nx + UW - 1) / UW;
    const size_t nunits = (size_t)batch * g->ny * upr;
    const size_t ncells = (size_t)batch * (g->nx + 1) * (g->ny + 1);
    const size_t total = (size_t)batch * max_points;
    const size_t nmax = ncells > total ? ncells : total;
    w.upr = upr;
    w.cap = bucket_capacity(max_points, g->ny * upr);
    w.state = ar.take<State>(1);
    w.counters = ar.take<int>(2 * NSUB * align_up(nunits, 4));
    w.wpack = ar.take<float>(packed_weight_floats(15));
    w.load = ar.take<int>(2048);
    w.perm = ar.take<int>(2048);
    w.key = ar.take<int>(total);
    w.cell_count = ar.take<int>(ncells);
    w.cell_rank = ar.take<int>(ncells + 1);
    w.kept_rank = ar.take<int>(total + 1);
    w.block_sums = ar.take<int>((nmax + SCAN_TILE - 1) / SCAN_TILE + 1);
    w.totals = ar.take<int>(4);
    w.cell_sums = ar.take<unsigned long long>((ncells < total ? ncells : total) * 3 + 3);
    if (with_buckets) {
        w.buckets = ar.take<float>(nunits * NSUB * w.cap * REC_STRIDE_MAX);
        w.ovf = ar.take<float>(total * REC_STRIDE_MAX + REC_STRIDE_MAX);
    } else {
        w.buckets = w.ovf = nullptr;
    }
    return align_up(ar.used, 256);
}

bool use_valu_impl() {
    const char *e = getenv("LAV_PILLAR_IMPL");
    return e && e[0] == 'v';
}

int persistent_workgroups() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        const char *e = getenv("LAV_PILLAR_WG_PER_CU");  // experiment knob
        const int per = e ? atoi(e) : 2;
        return cus * (per >= 1 && per <= 2 ? per : 2);
    }();
    return n;
}

// debug: per-workgroup phase times of one k_rows launch (100 MHz wall clock -> us)
void dump_trace(const unsigned long long *d_trace, int nwg, hipStream_t st) {
    std::vector<unsigned long long> h((size_t)nwg * 16);
    if (hipStreamSynchronize(st) != hipSuccess) return;
    if (hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 16]); tend = std::max(tend, h[i * 16 + 12]); }
    auto us = [&](unsigned long long t) { return t ? (double)(t - t0) / 100.0 : -1.0; };
    fprintf(stderr, "[pillar trace] %d workgroups, span %.1f us\n", nwg, us(tend));
    double acc[13] = {0}; int cntv[13] = {0};
    for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 13; ++k) if (h[i * 16 + k]) { acc[k] += us(h[i * 16 + k]); ++cntv[k]; }
    const char *nm[13] = {"start", "prologue issued", "partition", "g0 start", "g0 sums", "g0 empties out", "g0 sweep", "j0 layer1", "j0 layer2+max", "g0 jobs", "g0 out", "g1 start", "end"};
    for (int k = 0; k < 13; ++k) fprintf(stderr, "  mean %-15s %7.2f us (%d wgs)\n", nm[k], cntv[k] ? acc[k] / cntv[k] : 0.0, cntv[k]);
    {   // shader clock: s_memtime cycles over the wall-clock span of each workgroup
        double f = 0; int nf = 0;
        for (int i = 0; i < nwg; ++i) if (h[i * 16 + 12] > h[i * 16]) { f += (double)h[i * 16 + 13] / ((double)(h[i * 16 + 12] - h[i * 16]) / 100.0); ++nf; }
        fprintf(stderr, "  mean shader clock %.0f MHz\n", nf ? f / nf : 0.0);
    }
    {   // which workgroups shared a CU (XCC, SE, SH, CU of HW_ID)
        auto cu_of = [&](int i) { const unsigned long long v = h[(size_t)i * 16 + 14]; return (unsigned)(((v >> 32) & 0xf) << 16) | (unsigned)(v & 0xff00); };
        int paired = 0;
        for (int i = 0; i + nwg / 2 < nwg; ++i) paired += cu_of(i) == cu_of(i + nwg / 2);
        fprintf(stderr, "  workgroups i and i + %d on the same CU: %d of %d; cu of wg 0..7:", nwg / 2, paired, nwg / 2);
        for (int i = 0; i < std::min(nwg, 8); ++i) fprintf(stderr, " %06x", cu_of(i));
        fprintf(stderr, " | wg %d..:", nwg / 2);
        for (int i = nwg / 2; i < std::min(nwg, nwg / 2 + 8); ++i) fprintf(stderr, " %06x", cu_of(i));
        fprintf(stderr, "\n");
    }
    struct Row { double end; int i; };
    std::vector<Row> rows;
    for (int i = 0; i < nwg; ++i) rows.push_back({us(h[i * 16 + 12]), i});
    std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.end > y.end; });
    for (size_t k = 0; k < std::min<size_t>(rows.size(), 6); ++k) {
        const unsigned long long *r = &h[(size_t)rows[k].i * 16];
        fprintf(stderr, "  late wg %4d: hw %llx points %llu |", rows[k].i, r[14], r[15]);
        for (int q = 0; q < 13; ++q) fprintf(stderr, " %.1f", us(r[q]));
        fprintf(stderr, "\n");
    }
}

template <int D>
int launch_canvas(const PillarArgs &a, const Workspace &w, const lav_pointnet *net, float *canvas, bool want_keys, float *amax, hipStream_t st) {
    const long total = (long)a.batch * a.max_points;
    const int tok_prep = timer_begin("pillar_prep", st);
    // where the canvas' zeros come from: k_rows streams them itself (default) | LAV_PILLAR_ZERO=bin: extra workgroups of k_bin fill the
    // whole canvas while the others bin, k_rows writes only the quads that points touched.  Built in round 5,
    // bit-exact, measured SLOWER: config #2 stage 25.1-25.4 -> 27.9-28.1 us (k_bin 7.3 -> 9.2, k_rows 18.0 -> 18.7: taking 60 % of the
    // bytes out of k_rows does not shorten it - it is not store bound; profiles/r05_pillar_experiments.txt).  Kept as the A/B knob.
    static const bool zero_in_bin = [] { const char *e = getenv("LAV_PILLAR_ZERO"); return e && e[0] == 'b'; }();
    const long canvas_floats = (long)a.batch * C * a.ny * a.nx;
    const int nzero = zero_in_bin ? (int)std::min<long>(2 * persistent_workgroups(), std::max<long>(1, canvas_floats / 1024)) : 0;
    hipLaunchKernelGGL((k_bin<D>), dim3((unsigned)(std::max(1l, (total + 255) / 256) + (a.nwg + 63) / 64 + nzero)), dim3(256), 0, st, a, w.state, w.counters, w.buckets,
                       w.ovf, want_keys ? w.key : nullptr, net->w1, net->b1, net->w2, net->b2, w.wpack, w.load, w.perm, canvas, canvas_floats, nzero);
    timer_end(tok_prep, st);
    LAV_LAUNCH_CHECK();
    const int W = a.nwg;
    const bool vec4 = a.nx % 4 == 0 && reinterpret_cast<uintptr_t>(canvas) % 16 == 0;
    static const bool want_trace = getenv("LAV_PILLAR_TRACE") != nullptr;
    static unsigned long long *d_trace = nullptr;
    static int trace_runs = 0;
    RowsArgs at;
    at.batch = a.batch; at.nx = a.nx; at.ny = a.ny; at.UPR = a.UPR; at.NUP = a.NUP; at.cap = a.cap; at.rstride = a.rstride;
    at.min_x = a.min_x; at.min_y = a.min_y; at.ppm = a.ppm; at.trace = nullptr;
    at.perm = w.perm; at.load = w.load; at.amax = amax;

    if (want_trace && vec4) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)persistent_workgroups() * 16 * sizeof(unsigned long long)));
        LAV_HIP(hipMemsetAsync(d_trace, 0, (size_t)persistent_workgroups() * 16 * sizeof(unsigned long long), st));
        at.trace = d_trace;
    }
    const int tok = timer_begin("pointnet_scatter", st);
#define LAV_ROWS(MFMA, VEC, TR)                                                                                                 \
    do {                                                                                                                        \
        if (zero_in_bin)                                                                                                        \
            hipLaunchKernelGGL((k_rows<D, MFMA, VEC, TR, true>), dim3(W), dim3(256), 0, st, at, w.state, w.counters, w.buckets, w.ovf, net->w1, \
                               net->b1, net->w2, net->b2, w.wpack, canvas);                                                     \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_rows<D, MFMA, VEC, TR, false>), dim3(W), dim3(256), 0, st, at, w.state, w.counters, w.buckets, w.ovf, net->w1, \
                               net->b1, net->w2, net->b2, w.wpack, canvas);                                                     \
    } while (0)
    if constexpr (D == 11) {   // debug variants exist for the v2 agent's point width only
        if (want_trace && vec4) {
            LAV_ROWS(true, true, true);
            timer_end(tok, st);
            LAV_LAUNCH_CHECK();
            if (++trace_runs == 20) dump_trace(d_trace, W, st);
            return LAV_OK;
        }
        if (use_valu_impl()) {
            if (vec4) LAV_ROWS(false, true, false); else LAV_ROWS(false, false, false);
            timer_end(tok, st);
            LAV_LAUNCH_CHECK();
            return LAV_OK;
        }
    }
    {
        if (vec4) LAV_ROWS(true, true, false); else LAV_ROWS(true, false, false);
    }
#undef LAV_ROWS
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

int fill_args(PillarArgs &a, const float *points, const int *h_num_points, int batch, int max_points, int D, const lav_grid *grid,
              const Workspace &w, const char *who) {
    a.points = points;
    a.batch = batch;
    a.max_points = max_points;
    a.D = D;
    for (int b = 0; b < batch; ++b) {
        LAV_REQUIRE(h_num_points[b] >= 0, "%s: negative num_points", who);
        a.n[b] = h_num_points[b] < max_points ? h_num_points[b] : max_points;
    }
    for (int b = batch; b < MAX_BATCH; ++b) a.n[b] = 0;
    a.min_x = grid->min_x; a.max_x = grid->max_x; a.min_y = grid->min_y; a.max_y = grid->max_y; a.ppm = grid->ppm;
    a.nx = grid->nx; a.ny = grid->ny; a.KX = grid->nx + 1; a.KY = grid->ny + 1;
    a.UPR = w.upr;
    a.NUP = (int)align_up((size_t)batch * grid->ny * w.upr, 4);
    a.cap = w.cap;
    a.rstride = rec_stride_host(D);
    a.nwg = std::min(std::min(persistent_workgroups(), batch * grid->ny * w.upr), 2048);   // 2048: the pairing tables (load / perm, k_bin's s_key)
    a.trace = nullptr;
    return LAV_OK;
}

}  // namespace

extern "C" size_t lav_pillar_workspace_bytes(int batch, int max_points, const lav_grid *grid) {
    if (!grid || batch <= 0 || max_points < 0) return 0;
    Arena ar(nullptr, 0);
    Workspace w;
    return carve(ar, w, batch, max_points, grid, true);
}

extern "C" size_t lav_pillar_decorate_workspace_bytes(int batch, int max_points, const lav_grid *grid) {
    if (!grid || batch <= 0 || max_points < 0) return 0;
    Arena ar(nullptr, 0);
    Workspace w;
    return carve(ar, w, batch, max_points, grid, false);
}

extern "C" int lav_pillar_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(workspace || workspace_bytes == 0, "lav_pillar_workspace_init: null workspace");
    if (workspace_bytes) LAV_HIP(hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream)));
    return LAV_OK;
}

extern "C" int lav_pillar_amax_count(int batch, const lav_grid *grid) {
    if (!grid || batch <= 0) return 0;
    const int upr = (grid->nx + UW - 1) / UW;
    return std::min(std::min(persistent_workgroups(), batch * grid->ny * upr), 2048);   // = PillarArgs.nwg (fill_args)
}

extern "C" int lav_pillar_scatter(const float *points, const int *h_num_points, int batch, int max_points, int D,
                                  const lav_grid *grid, const lav_pointnet *net, float *canvas, int *unique_coords,
                                  int *inverse, int *counts, void *workspace, size_t workspace_bytes, void *stream) {
    return lav_pillar_scatter_amax(points, h_num_points, batch, max_points, D, grid, net, canvas, unique_coords, inverse, counts, nullptr,
                                   workspace, workspace_bytes, stream);
}

extern "C" int lav_pillar_scatter_amax(const float *points, const int *h_num_points, int batch, int max_points, int D,
                                       const lav_grid *grid, const lav_pointnet *net, float *canvas, int *unique_coords,
                                       int *inverse, int *counts, float *amax_parts, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(grid && net && canvas && h_num_points, "lav_pillar_scatter: null argument");
    LAV_REQUIRE(batch >= 1 && batch <= MAX_BATCH, "lav_pillar_scatter: batch %d outside [1,%d]", batch, MAX_BATCH);
    LAV_REQUIRE(max_points >= 0 && (points || max_points == 0), "lav_pillar_scatter: bad points");
    LAV_REQUIRE(net->channels == C, "lav_pillar_scatter: PointNet width %d unsupported (built for %d)", net->channels, C);
    LAV_REQUIRE(net->num_input == D + 5, "lav_pillar_scatter: num_input %d != D+5 (D=%d)", net->num_input, D);
    LAV_REQUIRE(grid->nx > 0 && grid->ny > 0, "lav_pillar_scatter: empty grid");
    LAV_REQUIRE(grid->nx < 4096 && grid->ny < 4096, "lav_pillar_scatter: grid %dx%d too large (12-bit cell coordinates)", grid->nx, grid->ny);
    LAV_REQUIRE((long)batch * (grid->nx + 1) * (grid->ny + 1) < (1l << 30) && (long)batch * max_points < (1l << 30),
                "lav_pillar_scatter: problem too large for 32-bit indices");
    hipStream_t st = static_cast<hipStream_t>(stream);

    Arena ar(workspace, workspace_bytes);
    Workspace w;
    carve(ar, w, batch, max_points, grid, true);
    if (!workspace || !ar.ok()) return fail(LAV_EWORKSPACE, "lav_pillar_scatter: workspace %zu < %zu bytes", workspace_bytes, ar.used);

    PillarArgs a;
    int rc = fill_args(a, points, h_num_points, batch, max_points, D, grid, w, "lav_pillar_scatter");
    if (rc) return rc;
    {   // canvas row 0 collects key rows ny-1..nx, the last column collects key columns nx-1..ny (reference clamp)
        const long lay = (long)(a.nx - (a.ny - 1) + 1 > 1 ? a.nx - (a.ny - 1) + 1 : 1) * ((a.ny - a.nx + 1 > 0 ? a.ny - a.nx + 1 : 0) + 1);
        LAV_REQUIRE(lay <= MAX_LAYERS, "lav_pillar_scatter: grid %dx%d needs %ld clamp layers (max %d)", a.nx, a.ny, lay, MAX_LAYERS);
    }
    const bool want_idx = unique_coords || inverse || counts;
    switch (D) {
        case 11: rc = launch_canvas<11>(a, w, net, canvas, want_idx, amax_parts, st); break;
        case 4: rc = launch_canvas<4>(a, w, net, canvas, want_idx, amax_parts, st); break;
        case 8: rc = launch_canvas<8>(a, w, net, canvas, want_idx, amax_parts, st); break;
        default: return fail(LAV_EINVAL, "lav_pillar_scatter: point width D=%d not instantiated (4,8,11)", D);
    }
    if (rc) return rc;

    if (want_idx) {
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
    LAV_REQUIRE(max_points >= 0 && (points || max_points == 0) && D >= 3 && D < 16, "lav_pillar_decorate: bad points");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Arena ar(workspace, workspace_bytes);
    Workspace w;
    carve(ar, w, batch, max_points, grid, false);
    if (!workspace || !ar.ok()) return fail(LAV_EWORKSPACE, "lav_pillar_decorate: workspace %zu < %zu bytes", workspace_bytes, ar.used);
    PillarArgs a;
    a.points = points; a.batch = batch; a.max_points = max_points; a.D = D;
    for (int b = 0; b < MAX_BATCH; ++b) a.n[b] = b < batch ? (h_num_points[b] < max_points ? h_num_points[b] : max_points) : 0;
    for (int b = 0; b < batch; ++b) LAV_REQUIRE(h_num_points[b] >= 0, "lav_pillar_decorate: negative num_points");
    a.min_x = grid->min_x; a.max_x = grid->max_x; a.min_y = grid->min_y; a.max_y = grid->max_y; a.ppm = grid->ppm;
    a.nx = grid->nx; a.ny = grid->ny; a.KX = grid->nx + 1; a.KY = grid->ny + 1;
    a.UPR = w.upr; a.NUP = 0; a.cap = w.cap; a.nwg = 0; a.trace = nullptr;
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

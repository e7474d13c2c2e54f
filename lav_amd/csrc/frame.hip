// Per-frame glue of LAVAgent.run_step that sits between the big kernels, one launch each instead of the ~60 tiny
// tensor ops the reference issues (a HIP-graph kernel node costs ~5 us of launch time on this stack, more than
// any of these ops computes):
//
//   lav_merge_ticks    cat([lidar, prev_lidar]) + preprocess() ego-box removal + prev_lidar = lidar
//                      (team_code_v2/lav_agent_fast.py:240-247, 450-452)
//   lav_stack_sweeps   history write + get_stacked_lidar(): move_lidar_points of sweeps t, t-5, t-10 into the current
//                      ego frame, feature columns, one-hot time channel (lav_agent_fast.py:363-383, 547-565)
//   lav_extract_peaks  sigmoid + 7x7 max-pool NMS + top-15 + gather of size / orientation at the peaks
//                      (team_code_v2/model_inference.py:95-121, 189-202)
//
// All three are HBM streaming / latency work: coalesced 16-byte loads, no LDS except for the NMS halo tile and the
// final top-k selection.  Compiled with -ffp-contract=off: the rotation below is two roundings per product-sum exactly
// as written, like the oracle.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {
using namespace lav;

// ------------------------------------------------------------------------------------------------ merge ticks
__global__ __launch_bounds__(256) void k_merge_ticks(const float4 *__restrict__ tick, float4 *__restrict__ prev, int rows,
                                                     float4 *__restrict__ cur) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const float4 a = tick[i], b = prev[i];
    auto mark = [](float4 p) {
        const bool ego = p.x > -2.4f && p.x < 0.f && p.y > -0.8f && p.y < 0.8f && p.z > -1.5f && p.z < -1.f;
        if (ego) p.x = __builtin_nanf("");
        return p;
    };
    cur[i] = mark(a);
    cur[rows + i] = mark(b);
    prev[i] = a;
}

// ------------------------------------------------------------------------------------------------ stack sweeps
template <int DIM, int NS>
__global__ __launch_bounds__(256) void k_stack_sweeps(const float *__restrict__ fused, float *__restrict__ ring,
                                                      const long *__restrict__ d_slot, const long *__restrict__ d_sweeps,
                                                      const float *__restrict__ d_R, const float *__restrict__ d_t, int rows,
                                                      float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= rows) return;
    const long slot = d_slot[0];
    const float *src = s == 0 ? fused + (long)i * DIM : ring + ((long)d_sweeps[s] * rows + i) * DIM;
    float v[DIM];
    if constexpr (DIM % 4 == 0) {
#pragma unroll
        for (int j = 0; j < DIM / 4; ++j) {
            const float4 q = reinterpret_cast<const float4 *>(src)[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < DIM; ++j) v[j] = src[j];
    }
    if (s == 0) {  // the newest sweep also enters the history ring
        float *dst = ring + ((long)slot * rows + i) * DIM;
        if constexpr (DIM % 4 == 0) {
#pragma unroll
            for (int j = 0; j < DIM / 4; ++j)
                reinterpret_cast<float4 *>(dst)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < DIM; ++j) dst[j] = v[j];
        }
    }
    const float *R = d_R + s * 9, *t = d_t + s * 3;
    float o[DIM + NS];
    // xyz @ R + t, products summed left to right
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = ((v[0] * R[c] + v[1] * R[3 + c]) + v[2] * R[6 + c]) + t[c];
#pragma unroll
    for (int j = 3; j < DIM; ++j) o[j] = v[j];
#pragma unroll
    for (int j = 0; j < NS; ++j) o[DIM + j] = j == s ? 1.f : 0.f;
    float *dst = out + ((long)s * rows + i) * (DIM + NS);
#pragma unroll
    for (int j = 0; j < DIM + NS; ++j) dst[j] = o[j];
}

// ------------------------------------------------------------------------------------------------ peaks
// Round 5 rewrite (56.6 -> measured in profiles/r05_*): 32 x 32 pixel tiles (a quarter of the workgroups, so a quarter of the
// same-address slot / ticket atomics), separable 7 + 7 tap max instead of 49 taps, the tile's survivors compacted before they are
// ranked against each other, and a last-arriver selection without barriers in its rounds: the two waves that own a class keep the
// class's candidates in registers and each extract their own best max_det (u64 max over the wave on v_permlane swaps + DPP, no DS
// instruction), one barrier, then the 2 x max_det keys of a class are ranked against each other.  Same rows as before: descending
// score, ties by ascending pixel index.
constexpr int PT = 32;              // tile side in pixels; 256 threads, four pixels each
constexpr int PEAK_MAX_KS = 15, PEAK_MAX_DET = 64, PEAK_MAX_CLS = 8;
constexpr int PEAK_PER_LANE = 16;   // candidates a lane of the selection keeps in registers: 2 waves x 64 x 16 = 2048 per class

__device__ __forceinline__ unsigned ordered(float f) {  // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// max of a 64-bit key over the wave, result in every lane (xor butterfly 32, 16, then rotations inside the rows of 16 lanes)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    auto take = [&](unsigned lo, unsigned hi) {
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    };
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        take(a[0] ^ a[1] ^ lo, b[0] ^ b[1] ^ hi);   // (x, x) swapped: one of the two results is the own value, the other the partner's
    }
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        take(a[0] ^ a[1] ^ lo, b[0] ^ b[1] ^ hi);
    }
#define LAV_ROT(ctrl) take((unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, ctrl, 0xf, 0xf, false), (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), ctrl, 0xf, 0xf, false))
    LAV_ROT(0x128); LAV_ROT(0x124); LAV_ROT(0x122); LAV_ROT(0x121);   // row_ror 8, 4, 2, 1
#undef LAV_ROT
    return v;
}

struct PeakArgs {
    const float *heat, *size, *ori;
    int ncls, H, W, ks, max_det, apply_sigmoid, size_c, ori_c, cand_stride;
    float *out;                       // [ncls][max_det][3 + size_c + ori_c]
    unsigned long long *cand;         // [ncls][cand_stride]
    int *count;                       // [ncls] + ticket at [ncls]
    unsigned long long *trace;        // debug (LAV_PEAKS_TRACE): [workgroup][8] wall-clock stamps, or null
};

__global__ __launch_bounds__(256) void k_extract_peaks(PeakArgs a) {
#define PK_STAMP(i) do { if (a.trace && threadIdx.x == 0) a.trace[((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
    PK_STAMP(0);
    constexpr int TMAX = PT + PEAK_MAX_KS - 1;
    __shared__ float s_tile[TMAX * TMAX];
    __shared__ float s_hmax[TMAX * PT];
    __shared__ unsigned long long s_list[PT * PT];
    __shared__ unsigned long long s_keep[PEAK_MAX_DET];
    __shared__ unsigned long long s_top[PEAK_MAX_CLS][2][PEAK_MAX_DET];
    __shared__ unsigned long long s_win[PEAK_MAX_CLS * PEAK_MAX_DET];
    __shared__ int s_flag, s_n, s_base;
    const int tid = threadIdx.x, cls = blockIdx.z;
    const int r = a.ks / 2, TW = PT + 2 * r, TH = PT + 2 * r;
    const int x0 = blockIdx.x * PT, y0 = blockIdx.y * PT;
    const float *hm = a.heat + (long)cls * a.H * a.W;
    if (tid == 0) s_n = 0;
    for (int j = tid; j < TW * TH; j += 256) {
        const int ty = j / TW, tx = j - ty * TW;
        const int y = y0 + ty - r, x = x0 + tx - r;
        float v = -__builtin_inff();  // max_pool2d pads with -inf
        if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
            v = hm[(long)y * a.W + x];
            if (a.apply_sigmoid) v = 1.f / (1.f + expf(-v));
        }
        s_tile[j] = v;
    }
    __syncthreads();
    PK_STAMP(1);
    // max over the ks x ks window = max over its rows of the row maxima (fmaxf ignores NaN in any order)
    for (int j = tid; j < TH * PT; j += 256) {
        const int ty = j / PT, lx = j - ty * PT;
        float m = s_tile[ty * TW + lx];
        for (int dx = 1; dx < a.ks; ++dx) m = fmaxf(m, s_tile[ty * TW + lx + dx]);
        s_hmax[j] = m;
    }
    __syncthreads();
    // possible_det = heat - (max > heat)*1e5: only non-suppressed pixels can reach the top-k while there are at least
    // max_det of them; NaN never compares greater, exactly like the reference's (max > heat)
#pragma unroll
    for (int i = 0; i < PT * PT / 256; ++i) {
        const int p = tid + 256 * i, lx = p % PT, ly = p / PT;
        const int x = x0 + lx, y = y0 + ly;
        if (x < a.W && y < a.H) {
            const float c = s_tile[(ly + r) * TW + lx + r];
            float m = c;
            for (int dy = 0; dy < a.ks; ++dy) m = fmaxf(m, s_hmax[(ly + dy) * PT + lx]);
            if (!(m > c)) s_list[atomicAdd(&s_n, 1)] = ((unsigned long long)ordered(c) << 32) | (0xffffffffu - (unsigned)(y * a.W + x));
        }
    }
    __syncthreads();
    PK_STAMP(2);
    // Of a tile's survivors only its own top max_det can be in the global top max_det (flat regions make EVERY pixel a survivor).
    const int n = s_n;
    for (int j = tid; j < n; j += 256) {
        const unsigned long long key = s_list[j];
        int rank = 0;
        for (int i = 0; i < n; ++i) rank += s_list[i] > key;
        if (rank < a.max_det) s_keep[rank] = key;   // keys are unique: ranks 0 .. min(n, max_det) - 1 are each taken once
    }
    __syncthreads();
    const int nk = min(n, a.max_det);
    // ONE global atomic per workgroup reserves the tile's slots
    if (tid == 0) s_base = nk ? atomicAdd(&a.count[cls], nk) : 0;
    __syncthreads();
    if (tid < nk)  // write-through (sc1) store: visible device-wide once it has left this wave, no L2 write-back fence needed
        __hip_atomic_store(&a.cand[(long)cls * a.cand_stride + s_base + tid], s_keep[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- hand-off to the last workgroup (MI355X guide G16, write-through form: sc1 payload stores, every wave drains
    // its stores, barrier, one relaxed agent-scope ticket; the last arriver takes ONE acquire before plain loads).
    PK_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int total = gridDim.x * gridDim.y * gridDim.z;
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&a.count[a.ncls], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == total - 1;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_flag = last;
    }
    __syncthreads();
    PK_STAMP(4);
    if (!s_flag) return;
    // ---- final selection by the last workgroup.  Waves 2p, 2p + 1 own the classes p, p + 2, ...; wave (2p + h) keeps the candidates
    // h*64 + lane + 128 j of the class in registers and extracts ITS best max_det in score order: no barrier, no LDS inside the rounds.
    const int ncol = 3 + a.size_c + a.ori_c;
    const int lane = tid & 63, wv = tid >> 6, pr = wv >> 1, hf = wv & 1;
    for (int c = pr; c < a.ncls; c += 2) {
        const int nc_ = min(__hip_atomic_load(&a.count[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.cand_stride);
        const unsigned long long *cand = a.cand + (long)c * a.cand_stride;
        const bool in_regs = nc_ <= 128 * PEAK_PER_LANE;
        unsigned long long mine[PEAK_PER_LANE];
#pragma unroll
        for (int j = 0; j < PEAK_PER_LANE; ++j) {
            const int i = hf * 64 + lane + 128 * j;
            mine[j] = in_regs && i < nc_ ? cand[i] : 0ull;
        }
        unsigned long long bound = ~0ull;  // keys are unique (they carry the pixel index): select strictly below the last pick
        for (int d = 0; d < a.max_det; ++d) {
            unsigned long long best = 0;
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < PEAK_PER_LANE; ++j)
                    if (mine[j] < bound && mine[j] > best) best = mine[j];
            } else {   // (more candidates than the registers hold: maps beyond 8 x 16 tiles; scan the list)
                for (int i = hf * 64 + lane; i < nc_; i += 128) {
                    const unsigned long long k = cand[i];
                    if (k < bound && k > best) best = k;
                }
            }
            best = wave_max_u64(best);
            if (lane == 0) s_top[c][hf][d] = best;
            bound = best ? best : 0ull;   // (0: this wave has run out of candidates - every later round yields 0 as well)
        }
    }
    __syncthreads();
    // the class's 2 x max_det keys ranked against each other: rank r < max_det is row r (0 = no candidate)
    for (int e = tid; e < a.ncls * 2 * a.max_det; e += 256) {
        const int c = e / (2 * a.max_det), i = e - c * 2 * a.max_det;
        const unsigned long long key = s_top[c][i / a.max_det][i % a.max_det];
        if (key) {
            int rank = 0;
            for (int j = 0; j < 2 * a.max_det; ++j) rank += s_top[c][j / a.max_det][j % a.max_det] > key;
            if (rank < a.max_det) s_win[c * PEAK_MAX_DET + rank] = key;
        }
    }
    {   // rows beyond the number of candidates
        for (int e = tid; e < a.ncls * a.max_det; e += 256) {
            const int c = e / a.max_det, d = e - c * a.max_det;
            int have = 0;
            for (int j = 0; j < 2 * a.max_det; ++j) have += s_top[c][j / a.max_det][j % a.max_det] != 0ull;
            if (d >= have) s_win[c * PEAK_MAX_DET + d] = 0ull;
        }
    }
    __syncthreads();
    PK_STAMP(5);
    for (int e = tid; e < a.ncls * a.max_det * ncol; e += 256) {
        const int col = e % ncol, d = (e / ncol) % a.max_det, c = e / (ncol * a.max_det);
        const unsigned long long best = s_win[c * PEAK_MAX_DET + d];
        float v;
        if (best == 0) {  // fewer candidates than max_det: a suppressed pixel's score, never above any threshold
            v = col == 0 ? -1e5f : 0.f;
        } else {
            const unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
            const int py = idx / a.W, px = idx - py * a.W;
            if (col == 0) v = unordered((unsigned)(best >> 32));
            else if (col == 1) v = (float)px;
            else if (col == 2) v = (float)py;
            else if (col < 3 + a.size_c) v = a.size[((long)(col - 3) * a.H + py) * a.W + px];
            else v = a.ori[((long)(col - 3 - a.size_c) * a.H + py) * a.W + px];
        }
        a.out[e] = v;
    }
    PK_STAMP(6);
    if (tid <= a.ncls) __hip_atomic_store(&a.count[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
}
}  // namespace

// ------------------------------------------------------------------------------------------------ pooled branch
namespace {
__global__ __launch_bounds__(256) void k_pool_affine(const float *__restrict__ x, int C, int H, int W, const float *__restrict__ scale,
                                                     const float *__restrict__ shift, int relu, float *__restrict__ y, int out_c_total,
                                                     int out_c_offset, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // one output element (n, c, oy, ox) of the pooled tensor
    if (e >= total) return;
    const int OW = W >> 1, OH = H >> 1;
    const int ox = (int)(e % OW), oy = (int)((e / OW) % OH), c = (int)((e / ((long)OW * OH)) % C);
    const long n = e / ((long)OW * OH * C);
    const float2 *p = reinterpret_cast<const float2 *>(x + ((n * C + c) * H + 2 * oy) * (long)W + 2 * ox);
    const float2 r0 = p[0], r1 = p[W >> 1];
    float v = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
    v = fmaf(v, scale[c], shift[c]);
    if (relu) v = v > 0.f ? v : 0.f;
    y[((n * out_c_total + out_c_offset + c) * OH + oy) * (long)OW + ox] = v;
}
}  // namespace

extern "C" int lav_pool_affine(const float *x, int batch, int channels, int h, int w, const float *scale, const float *shift, int relu,
                               float *y, int out_c_total, int out_c_offset, void *stream) {
    LAV_REQUIRE(batch >= 1 && channels >= 1 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "lav_pool_affine: even h, w expected");
    LAV_REQUIRE(x && scale && shift && y && out_c_offset >= 0 && out_c_offset + channels <= out_c_total, "lav_pool_affine: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long total = (long)batch * channels * (h / 2) * (w / 2);
    const int tok = timer_begin("pool_affine", st);
    hipLaunchKernelGGL(k_pool_affine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, channels, h, w, scale, shift, relu, y,
                       out_c_total, out_c_offset, total);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_merge_ticks(const float *tick, float *prev, int rows, int dim, float *cur, void *stream) {
    LAV_REQUIRE(rows >= 0 && dim == 4, "lav_merge_ticks: rows >= 0 and dim == 4 (x, y, z, intensity) expected");
    if (rows == 0) return LAV_OK;
    LAV_REQUIRE(tick && prev && cur, "lav_merge_ticks: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("merge_ticks", st);
    hipLaunchKernelGGL(k_merge_ticks, dim3((rows + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float4 *>(tick),
                       reinterpret_cast<float4 *>(prev), rows, reinterpret_cast<float4 *>(cur));
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_stack_sweeps(const float *fused, float *ring, const long *d_slot, const long *d_sweeps, const float *d_R,
                                const float *d_t, int num_sweeps, int rows, int dim, float *out, void *stream) {
    LAV_REQUIRE(rows >= 0 && num_sweeps == 3 && dim == 8, "lav_stack_sweeps: built for 3 sweeps of 8-float painted points");
    if (rows == 0) return LAV_OK;
    LAV_REQUIRE(fused && ring && d_slot && d_sweeps && d_R && d_t && out, "lav_stack_sweeps: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("stack_sweeps", st);
    hipLaunchKernelGGL((k_stack_sweeps<8, 3>), dim3((rows + 255) / 256, num_sweeps), dim3(256), 0, st, fused, ring, d_slot,
                       d_sweeps, d_R, d_t, rows, out);
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" size_t lav_extract_peaks_workspace_bytes(int ncls, int h, int w) {
    if (ncls <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)ncls * h * w * sizeof(unsigned long long) + align_up((size_t)(ncls + 1) * sizeof(int), 256);
}

extern "C" int lav_extract_peaks(const float *heat, int ncls, int h, int w, int ks, int max_det, int apply_sigmoid,
                                 const float *size, int size_c, const float *ori, int ori_c, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(ncls > 0 && ncls <= PEAK_MAX_CLS && h > 0 && w > 0 && (long)h * w < (1l << 31), "lav_extract_peaks: bad sizes (at most %d planes)", PEAK_MAX_CLS);
    LAV_REQUIRE(ks >= 1 && (ks & 1) && ks <= PEAK_MAX_KS, "lav_extract_peaks: odd kernel size <= %d expected", PEAK_MAX_KS);
    LAV_REQUIRE(max_det >= 1 && max_det <= PEAK_MAX_DET, "lav_extract_peaks: max_det in [1, %d]", PEAK_MAX_DET);
    LAV_REQUIRE(size_c >= 0 && ori_c >= 0 && 3 + size_c + ori_c <= 256, "lav_extract_peaks: too many gathered channels");
    LAV_REQUIRE(heat && out && (size || !size_c) && (ori || !ori_c), "lav_extract_peaks: null argument");
    const size_t need = lav_extract_peaks_workspace_bytes(ncls, h, w);
    if (!workspace || workspace_bytes < need) return fail(LAV_EWORKSPACE, "lav_extract_peaks: workspace %zu < %zu bytes", workspace_bytes, need);
    PeakArgs a;
    a.heat = heat; a.size = size; a.ori = ori;
    a.ncls = ncls; a.H = h; a.W = w; a.ks = ks; a.max_det = max_det; a.apply_sigmoid = apply_sigmoid; a.size_c = size_c; a.ori_c = ori_c;
    a.out = out;
    a.cand = static_cast<unsigned long long *>(workspace);
    // counters sit at the END of the caller's buffer (zero before the first launch, left zero by every launch)
    a.count = reinterpret_cast<int *>(static_cast<char *>(workspace) + workspace_bytes - align_up((size_t)(ncls + 1) * sizeof(int), 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("extract_peaks", st);
    const dim3 grid((w + PT - 1) / PT, (h + PT - 1) / PT, ncls);
    a.cand_stride = h * w;
    a.trace = nullptr;
    static const bool want_trace = getenv("LAV_PEAKS_TRACE") != nullptr;
    static unsigned long long *d_trace = nullptr;
    static int runs = 0;
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    if (want_trace && nwg <= 4096) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, 4096 * 8 * sizeof(unsigned long long)));
        LAV_HIP(hipMemsetAsync(d_trace, 0, nwg * 64, st));
        a.trace = d_trace;
    }
    hipLaunchKernelGGL(k_extract_peaks, grid, dim3(256), 0, st, a);
    if (a.trace && ++runs % 10 == 0) {
        std::vector<unsigned long long> h(nwg * 8);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < nwg; ++i) t0 = std::min(t0, h[i * 8]);
            double ph[4] = {0, 0, 0, 0}, last_start = 0, last_ticket = 0;
            for (size_t i = 0; i < nwg; ++i) {
                for (int k = 0; k < 4; ++k) ph[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]) / 100.0;
                last_start = std::max(last_start, (double)(h[i * 8] - t0) / 100.0);
                last_ticket = std::max(last_ticket, (double)(h[i * 8 + 4] - t0) / 100.0);
                if (h[i * 8 + 6]) fprintf(stderr, "[peaks trace] last workgroup: ticket at %.2f us, selection done %.2f, rows written %.2f\n",
                                          (double)(h[i * 8 + 4] - t0) / 100.0, (double)(h[i * 8 + 5] - t0) / 100.0, (double)(h[i * 8 + 6] - t0) / 100.0);
            }
            fprintf(stderr, "[peaks trace] %zu wgs: last start %.2f us, last ticket %.2f | mean us: tile load %.2f | nms %.2f | rank+emit %.2f | drain+ticket %.2f\n",
                    nwg, last_start, last_ticket, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg);
        }
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

// =========================================================================================================
// Detection decode on the device: which of the vehicle peaks become "other vehicles" of the motion forecast, their
// ego-frame position and heading, and how many there are - InferModel.det_inference's filters
// (team_code_v2/model_inference.py:95-121) followed by the others loop of InferModel.forward (:125-144).
// One wave; lane j owns peak row j of the vehicle class.  Comparisons in double on widened float32 values, like the
// Python floats of the reference.  Survivors keep their score order (prefix count over the ballot).
namespace {
__global__ __launch_bounds__(64) void k_det_decode(const float *__restrict__ rows, int cls, int max_det, double min_score, double ego_x,
                                                   double ego_y, double near_px, double far_px, double min_box, double cx, double cy,
                                                   double skip_px, double ppm, float *__restrict__ actors, int *__restrict__ n_out,
                                                   int ncls, float *host_rows, int *host_n, unsigned *host_seq) {
    const int j = threadIdx.x;
    bool ok = false;
    double X = 0, Y = 0, co = 1, si = 0;
    if (j < max_det) {
        const float *r = rows + ((long)cls * max_det + j) * 7;
        const double s = r[0], w = r[3], h = r[4];
        const long xi = (long)r[1], yi = (long)r[2];
        X = (double)xi; Y = (double)yi; co = r[5]; si = r[6];
        const double dist = sqrt((double)((xi - (long)ego_x) * (xi - (long)ego_x) + (yi - (long)ego_y) * (yi - (long)ego_y)));
        // every threshold is the host rule's Python float (float64): a score of exactly 0.2f passes 's > 0.2' on both sides
        ok = s > min_score && dist > near_px && dist < far_px && !(fmax(w, h) < min_box);
        ok = ok && sqrt((X - cx) * (X - cx) + (Y - cy) * (Y - cy)) > skip_px;
    }
    const unsigned long long m = __ballot(ok);
    const int pos = __popcll(m & ((1ull << j) - 1ull));
    if (j < max_det) {   // rows beyond the count read as an actor at the origin: finite inputs for the skipped crops
        actors[j * 2 + 0] = 0.f; actors[j * 2 + 1] = 0.f; actors[2 * max_det + j] = 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (ok) {
        actors[pos * 2 + 0] = (float)((X - cx) / ppm);
        actors[pos * 2 + 1] = (float)((Y - cy) / ppm);
        actors[2 * max_det + pos] = (float)atan2(si, co);
    }
    if (j == 0) *n_out = __popcll(m);
    if (host_rows) {
        // lav_det_decode_report: the peak rows and the count also go to host-visible (pinned, device-mapped) memory from inside this
        // launch, and a sequence word behind them - what the caller would otherwise fetch with two device->host copies and an event
        // between the heads and the others graph (three packets on the frame's critical chain).  System-scope release: the rows are
        // out of this wave's write path before the word that announces them.
        const int total = ncls * max_det * 7;
        for (int i = j; i < total; i += 64) __builtin_nontemporal_store(rows[i], host_rows + i);
        if (j == 0) __builtin_nontemporal_store((int)__popcll(m), host_n);
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence_system();
        __syncthreads();
        if (j == 0) __hip_atomic_store(host_seq, __hip_atomic_load(host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

extern "C" int lav_det_decode_report(const float *rows, int ncls, int max_det, int cls, double min_score, double ego_x, double ego_y,
                                     double near_px, double far_px, double min_box, double cx, double cy, double skip_px, double ppm,
                                     float *actors, int *n_out, float *host_rows, int *host_n, unsigned *host_seq, void *stream) {
    LAV_REQUIRE(rows && actors && n_out, "lav_det_decode: null argument");
    LAV_REQUIRE(ncls >= 1 && cls >= 0 && cls < ncls && max_det >= 1 && max_det <= 64, "lav_det_decode: bad sizes (max_det <= 64)");
    LAV_REQUIRE((host_rows != nullptr) == (host_n != nullptr) && (host_rows != nullptr) == (host_seq != nullptr),
                "lav_det_decode_report: host_rows, host_n and host_seq go together");
    hipLaunchKernelGGL(k_det_decode, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), rows, cls, max_det, min_score, ego_x, ego_y,
                       near_px, far_px, min_box, cx, cy, skip_px, ppm, actors, n_out, ncls, host_rows, host_n, host_seq);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_det_decode(const float *rows, int ncls, int max_det, int cls, double min_score, double ego_x, double ego_y,
                              double near_px, double far_px, double min_box, double cx, double cy, double skip_px, double ppm,
                              float *actors, int *n_out, void *stream) {
    return lav_det_decode_report(rows, ncls, max_det, cls, min_score, ego_x, ego_y, near_px, far_px, min_box, cx, cy, skip_px, ppm, actors, n_out,
                                 nullptr, nullptr, nullptr, stream);
}

// GRU waypoint decoders of the uniplanner (cast: 6 x GRU(512->64); plan: GRU(4->512) iterated 5x).
//
// Replaces UniPlanner.cast / plan / _plan of the reference (team_code_v2/models/uniplanner.py:255-308),
// which issues 6 (cast) + 30 (plan) cuDNN GRU calls of 20 steps each per frame.
//
// cast  - one workgroup per (sample, command).  The GRU input is the same embedding at every step, so
//         W_ih.embd + b_ih is computed once (wave-cooperative coalesced row dot products); W_hh (192x64) lives
//         in registers (one row per thread) and the 20-step recurrence, the Linear(64->2) and the cumulative
//         sum all run inside the workgroup.
// plan  - W_hh is 1536x512 fp32 = 3.1 MB: it does not fit one CU, so a step is spread over H/8 workgroups
//         (each owns 8 hidden units = 24 rows of W_hh, read coalesced from L2) and the time steps are
//         separate launches on the stream (first version; see DESIGN.md for the persistent variant).
//         Inputs are not autoregressive inside an iteration (uniplanner.py:264-270), so W_ih.u_t is formed on
//         the fly (4 FMAs).  Command branches never interact: with cmd >= 0 only that branch is evaluated.
//
// Gate order r, z, n as in torch.nn.GRU;  h' = (1-z)*n + z*h;  precise expf/tanhf (no fast-math).
#include <cstdlib>

#include "common.hpp"
#include "gru_seq.hpp"

namespace {
using namespace lav;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Cross-lane sums without LDS hardware (__shfl_xor compiles to ds_bpermute_b32, a DS instruction): gfx950's v_permlane32_swap /
// v_permlane16_swap exchange halves / rows of 16 lanes between two registers, DPP row rotations do the rest.
__device__ __forceinline__ float fold32(float x, float y) {   // lanes 0-31: x[l] + x[l+32]; lanes 32-63: y[l-32] + y[l]  (or the halves exchanged)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float x, float y) {   // even rows of 16 lanes: x's row pair summed; odd rows: y's
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_allsum(float x) {   // xor butterfly 8, 4, 2, 1 inside each row of 16 lanes (rotations of a period-2d value)
    x = dpp_add<0x128>(x);   // row_ror:8
    x = dpp_add<0x124>(x);   // row_ror:4
    x = dpp_add<0x122>(x);   // row_ror:2
    x = dpp_add<0x121>(x);   // row_ror:1
    return x;
}
// the xor butterfly 32, 16, 8, 4, 2, 1 of one value on those instructions: bit-identical to wave_sum (fp addition is commutative)
__device__ __forceinline__ float wave_sum_valu(float v) { return row_allsum(fold16(fold32(v, v), fold32(v, v))); }

// ------------------------------------------------------------------------------------------------- cast
// H = 64 fixed by the register layout (one W_hh row of 64 floats per thread, 3H = 192 threads)
constexpr int CAST_H = 64;

constexpr int CAST_THREADS = 768;  // 12 waves share the 192x512 input projection; waves 0-2 run the recurrence

// Extras of lav_embed_cast (each optional): the input is a feature map [B][embd_dim][hw] whose spatial mean is the embedding
// (AdaptiveAvgPool2d + Flatten of the embedder, uniplanner.py:36-40) and is written to embd_out; the command scores
// sigmoid(cmd_w . embd + cmd_b) (cast_cmd_pred, :50-53); the decoded waypoints rotated by the actor's heading and moved to
// its position (transform_points + translate, model_inference.py:164-165).
struct CastExtra {
    int hw;
    float *embd_out;
    const float *cmd_w, *cmd_b;
    float *cmds_out;
    const float *oris, *locs;
};

__global__ __launch_bounds__(CAST_THREADS) void k_gru_cast(const float *__restrict__ embd, int embd_dim, int num_cmds, int T,
                                                  const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                                  const float *__restrict__ b_ih, const float *__restrict__ b_hh,
                                                  const float *__restrict__ mlp_w, const float *__restrict__ mlp_b,
                                                  float *__restrict__ out, const int *__restrict__ n_valid, CastExtra x) {
    constexpr int H = CAST_H, G = 3 * H;
    __shared__ float gi[G];
    __shared__ float gh[G];
    __shared__ float h[H];
    __shared__ float s_e[1024];   // the embedding (embd_dim <= 1024)
    const int cmd = blockIdx.x, b = blockIdx.y;
    if (n_valid && b >= *n_valid) return;   // lav_batch_limit
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the embedding: given, or the spatial mean of the feature map (sum in pixel order, then one division - torch's mean)
    for (int c = tid; c < embd_dim; c += CAST_THREADS) {
        const float *src = embd + ((long)b * embd_dim + c) * x.hw;
        float sum = src[0];
        for (int i = 1; i < x.hw; ++i) sum += src[i];
        const float m = x.hw > 1 ? sum / (float)x.hw : sum;
        s_e[c] = m;
        if (x.embd_out && cmd == 0) x.embd_out[(long)b * embd_dim + c] = m;
    }
    __syncthreads();
    const float *e = s_e;
    const float *wih = w_ih + (long)cmd * G * embd_dim;
    // input projection, once: wave `wid` owns 16 rows, done 8 at a time with independent accumulators so the
    // 64 coalesced row loads and the 8 butterfly reductions of a batch are all in flight together
    for (int r0 = wid * 16; r0 < wid * 16 + 16; r0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k = lane; k < embd_dim; k += 64) {
            const float ev = e[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(wih[(long)(r0 + j) * embd_dim + k], ev, acc[j]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], d, 64);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gi[r0 + j] = acc[j] + b_ih[cmd * G + r0 + j];
        }
    }
    if (x.cmds_out && wid == 11) {   // this command's score: one more row on the last helper wave
        float acc = 0.f;
        for (int k = lane; k < embd_dim; k += 64) acc = fmaf(x.cmd_w[(long)cmd * embd_dim + k], e[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) x.cmds_out[(long)b * num_cmds + cmd] = sigmoidf_(acc + x.cmd_b[cmd]);
    }
    __syncthreads();  // gi[] complete and visible
    if (tid >= G) return;  // the nine helper waves are done (terminated waves drop out of later barriers)
    float w[H];
    {
        const float *wr = w_hh + ((long)cmd * G + tid) * H;
#pragma unroll
        for (int k = 0; k < H; ++k) w[k] = wr[k];
    }
    const float bh = b_hh[cmd * G + tid];
    if (tid < H) h[tid] = 0.f;
    const float m0 = tid < H ? mlp_w[(cmd * 2 + 0) * H + tid] : 0.f;
    const float m1 = tid < H ? mlp_w[(cmd * 2 + 1) * H + tid] : 0.f;
    float run0 = 0.f, run1 = 0.f;
    // (x, y) @ [[cos, sin], [-sin, cos]] + loc
    float rc = 1.f, rs = 0.f, lx = 0.f, ly = 0.f;
    if (x.oris) { const float o = x.oris[b]; rc = cosf(o); rs = sinf(o); }
    if (x.locs) { lx = x.locs[2 * b]; ly = x.locs[2 * b + 1]; }
    const bool xform = x.oris != nullptr || x.locs != nullptr;
    __syncthreads();
    float *o = out + (((long)b * num_cmds + cmd) * T) * 2;
    for (int t = 0; t < T; ++t) {
        float acc = bh;
#pragma unroll
        for (int k = 0; k < H; ++k) acc = fmaf(w[k], h[k], acc);
        gh[tid] = acc;
        __syncthreads();
        if (tid < H) {
            const float r = sigmoidf_(gi[tid] + gh[tid]);
            const float z = sigmoidf_(gi[H + tid] + gh[H + tid]);
            const float n = tanhf(gi[2 * H + tid] + r * gh[2 * H + tid]);
            const float hn = (1.f - z) * n + z * h[tid];
            h[tid] = hn;  // only this thread reads h[tid] in this phase; others read it after the barrier
            const float s0 = wave_sum(m0 * hn), s1 = wave_sum(m1 * hn);
            if (tid == 0) {
                run0 += s0 + mlp_b[cmd * 2 + 0];
                run1 += s1 + mlp_b[cmd * 2 + 1];
                o[t * 2 + 0] = xform ? (run0 * rc + run1 * -rs) + lx : run0;
                o[t * 2 + 1] = xform ? (run0 * rs + run1 * rc) + ly : run1;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------- plan
constexpr int PLAN_UNITS = 8;  // hidden units per workgroup  -> 24 W_hh rows, 6 per wave
constexpr int PLAN_RC = 6;     // state rows (sample x command) processed per register pass
constexpr int PLAN_MAXK = 8;   // H <= 64*PLAN_MAXK = 512

struct PlanArgs {
    const float *embd, *nxp, *cast_locs;
    const float *w_ih, *w_hh, *b_ih, *b_hh, *mlp_w, *mlp_b;
    float *out;   // [B][iters][NC][T][2]
    float *hseq;  // [T][R][H]
    int B, H, num_cmds, T, iters, cmd, NC, R;
    float ppm, crop;
};

// row r of the state = (sample b, evaluated command ci); ci maps to command c
__device__ __forceinline__ void row_decode(const PlanArgs &a, int r, int &b, int &ci, int &c) {
    b = r / a.NC;
    ci = r - b * a.NC;
    c = a.cmd >= 0 ? a.cmd : ci;
}

__global__ __launch_bounds__(256) void k_plan_step(PlanArgs a, int it, int t) {
    const int H = a.H;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = H / 64;
    __shared__ float gh_s[3 * PLAN_UNITS][PLAN_RC];
    const int j0 = blockIdx.x * PLAN_UNITS;
    // this wave's 6 rows of W_hh: local row lr = wid*6 + q  ->  gate g = lr / 8, unit j0 + lr % 8
    float w[6][PLAN_MAXK];
    float bh[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int lr = wid * 6 + q;
        const int row = (lr / PLAN_UNITS) * H + j0 + (lr % PLAN_UNITS);
        const float *wr = a.w_hh + (long)row * H;
#pragma unroll
        for (int i = 0; i < PLAN_MAXK; ++i) w[q][i] = i < nk ? wr[lane + 64 * i] : 0.f;
        bh[q] = a.b_hh[row];
    }
    const float *h_prev_base = t == 0 ? nullptr : a.hseq + (long)(t - 1) * a.R * H;
    float *h_out = a.hseq + (long)t * a.R * H;
    for (int r0 = 0; r0 < a.R; r0 += PLAN_RC) {
        const int nr = min(PLAN_RC, a.R - r0);
        float hv[PLAN_RC][PLAN_MAXK];
#pragma unroll
        for (int rr = 0; rr < PLAN_RC; ++rr) {
            int b, ci, c;
            row_decode(a, min(r0 + rr, a.R - 1), b, ci, c);
            const float *hp = t == 0 ? a.embd + (long)b * H : h_prev_base + (long)min(r0 + rr, a.R - 1) * H;
#pragma unroll
            for (int i = 0; i < PLAN_MAXK; ++i) hv[rr][i] = i < nk ? hp[lane + 64 * i] : 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < PLAN_RC; ++rr) {
            if (rr < nr) {  // workgroup-uniform: inference has a single live state row
                float acc[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[q] = 0.f;
#pragma unroll
                    for (int i = 0; i < PLAN_MAXK; ++i) acc[q] = fmaf(w[q][i], hv[rr][i], acc[q]);
                }
                // six independent butterfly reductions, interleaved so their shuffle latencies overlap
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
                    for (int q = 0; q < 6; ++q) acc[q] += __shfl_xor(acc[q], d, 64);
                if (lane == 0) {
                    {   // committed LDS stores (common.hpp: ds_write data hazards beside matrix-heavy neighbours)
                            float gv[6];
#pragma unroll
                            for (int q = 0; q < 6; ++q) gv[q] = acc[q] + bh[q];
#pragma unroll
                            for (int q = 0; q < 6; ++q) { gh_s[wid * 6 + q][rr] = gv[q]; lav::lds_store_fence(); }
                            lav::lds_commit();
#pragma unroll
                            for (int q = 0; q < 6; ++q) lav::lds_keep(gv[q]);
                        }
                }
            }
        }
        __syncthreads();
        if (tid < PLAN_UNITS * PLAN_RC) {
            const int u = tid % PLAN_UNITS, rr = tid / PLAN_UNITS;
            if (rr < nr) {
                const int r = r0 + rr, j = j0 + u;
                int b, ci, c;
                row_decode(a, r, b, ci, c);
                // u_t = [nxp*ppm/crop*2-1, previous iteration's waypoint]
                const float *prev = it == 0 ? a.cast_locs + (((long)b * a.num_cmds + c) * a.T + t) * 2
                                            : a.out + ((((long)b * a.iters + (it - 1)) * a.NC + ci) * a.T + t) * 2;
                float uu[4];
                uu[0] = a.nxp[b * 2 + 0] * a.ppm / a.crop * 2.f - 1.f;
                uu[1] = a.nxp[b * 2 + 1] * a.ppm / a.crop * 2.f - 1.f;
                uu[2] = prev[0];
                uu[3] = prev[1];
                float gi[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int row = g * H + j;
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(a.w_ih[row * 4 + k], uu[k], acc);
                    gi[g] = acc + a.b_ih[row];
                }
                const float hp = t == 0 ? a.embd[(long)b * H + j] : h_prev_base[(long)r * H + j];
                const float rg = sigmoidf_(gi[0] + gh_s[0 * PLAN_UNITS + u][rr]);
                const float zg = sigmoidf_(gi[1] + gh_s[1 * PLAN_UNITS + u][rr]);
                const float ng = tanhf(gi[2] + rg * gh_s[2 * PLAN_UNITS + u][rr]);
                h_out[(long)r * H + j] = (1.f - zg) * ng + zg * hp;
            }
        }
        __syncthreads();
    }
}

// Many state rows (the trainer's frozen teacher: batch x commands = 100+ rows): a step is lav::launch_gru_fwd_step (16 rows x 16
// units per workgroup, recurrent GEMM on MFMA) instead of k_plan_step, whose every workgroup walks ALL rows six at a time.
// This kernel lays out what that step reads: u[r][t] = (target point in crop units, previous iteration's waypoint) and, on the
// first iteration, the initial state h0[r] = embd[sample of r].
constexpr int PLAN_MFMA_MIN_ROWS = 16;
__global__ __launch_bounds__(256) void k_plan_inputs(PlanArgs a, int it, float *__restrict__ u, float *__restrict__ h0) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)a.R * a.T) {
        const int r = (int)(i / a.T), t = (int)(i - (long)r * a.T);
        int b, ci, c;
        row_decode(a, r, b, ci, c);
        const float *prev = it == 0 ? a.cast_locs + (((long)b * a.num_cmds + c) * a.T + t) * 2
                                    : a.out + ((((long)b * a.iters + (it - 1)) * a.NC + ci) * a.T + t) * 2;
        *reinterpret_cast<float4 *>(u + i * 4) =
            make_float4(a.nxp[b * 2 + 0] * a.ppm / a.crop * 2.f - 1.f, a.nxp[b * 2 + 1] * a.ppm / a.crop * 2.f - 1.f, prev[0], prev[1]);
    }
    if (it == 0) {
        for (long k = i; k < (long)a.R * a.H; k += (long)gridDim.x * 256) {
            const int r = (int)(k / a.H);
            h0[k] = a.embd[(long)(r / a.NC) * a.H + (k - (long)r * a.H)];
        }
    }
}

// After the T steps of iteration `it`: loc[r][t] = cumsum_t(mlp(h_t)) + prev[r][t].  One workgroup per state row:
// the T dot products run on separate waves, the 20-term cumulative sum on one thread.
constexpr int OUT_WAVES = 8;
__global__ __launch_bounds__(64 * OUT_WAVES) void k_plan_out(PlanArgs a, int it) {
    __shared__ float wp[64][2];  // T <= 64
    const int r = blockIdx.x, lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), H = a.H;
    int b, ci, c;
    row_decode(a, r, b, ci, c);
    for (int t = wid; t < a.T; t += OUT_WAVES) {
        const float *h = a.hseq + ((long)t * a.R + r) * H;
        float s0 = 0.f, s1 = 0.f;
        for (int k = lane; k < H; k += 64) {
            const float hv = h[k];
            s0 = fmaf(a.mlp_w[k], hv, s0);
            s1 = fmaf(a.mlp_w[H + k], hv, s1);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s0 += __shfl_xor(s0, d, 64);
            s1 += __shfl_xor(s1, d, 64);
        }
        if (lane == 0) {
            wp[t][0] = s0 + a.mlp_b[0];
            wp[t][1] = s1 + a.mlp_b[1];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float run0 = 0.f, run1 = 0.f;
        float *o = a.out + ((((long)b * a.iters + it) * a.NC + ci) * a.T) * 2;
        for (int t = 0; t < a.T; ++t) {
            const float *prev = it == 0 ? a.cast_locs + (((long)b * a.num_cmds + c) * a.T + t) * 2
                                        : a.out + ((((long)b * a.iters + (it - 1)) * a.NC + ci) * a.T + t) * 2;
            run0 += wp[t][0];
            run1 += wp[t][1];
            o[t * 2 + 0] = run0 + prev[0];
            o[t * 2 + 1] = run1 + prev[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------- plan, persistent
// All iters*T dependent steps in ONE launch.  H/8 workgroups stay resident, each keeping its 24 rows of W_hh in
// registers; after every step the workgroups exchange the new hidden state through 8-byte {epoch, value} granules
// written with relaxed agent-scope (write-through) atomic stores and polled with relaxed agent-scope atomic loads:
// the data is the flag, so no fences and no separate counters (MI355X guide, guideline 16 form R2).  Two granule
// buffers alternate; a workgroup can only be one step ahead of the slowest one, so a buffer is never overwritten while
// still being read.  Every workgroup re-derives the waypoints (Linear(512->2) + cumsum) itself because it needs them
// as GRU inputs in the next iteration; workgroup 0 also writes them out.  Spins are bounded: a workgroup that times out
// (its peers are not co-resident, e.g. the chip is oversubscribed by other streams) stops waiting, sets the status word
// of the workspace (lav_gru_plan_status) and returns; k_plan_poison, enqueued right behind the persistent kernel, then
// overwrites the WHOLE output with NaN if the status word is set (one agent, after every workgroup has left: status and output
// always agree), so that a plan that was not computed can never be mistaken for one: the reference agent's own rule for NaN
// waypoints (no steering, no throttle, lav_agent_fast.py:325-328) applies, and lav_gru_plan_steps recomputes it without any
// co-residency requirement.
constexpr long long PLAN_SPIN_LIMIT = 1ll << 22;
constexpr size_t PLAN_CTRL_BYTES = 512;   // head of a plan workspace: sticky counters + status / diagnosis words

// sticky[0] / sticky[1]: aborted / all persistent launches on this workspace since the caller zero-filled it (outside
// the range the per-launch memset clears) - a benchmark or a drive reads them once at its end (lav_gru_plan_diag words 10, 11).
__global__ __launch_bounds__(256) void k_plan_reset(uint4 *__restrict__ p, unsigned n16) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void k_plan_poison(const int *__restrict__ status, int *__restrict__ sticky, float *__restrict__ out, long n_out) {
    const bool aborted = *status != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(sticky + 1, 1);
        if (aborted) atomicAdd(sticky, 1);
    }
    if (!aborted) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (long)gridDim.x * 256) out[i] = __uint_as_float(0x7fc00000u);
}

// Diagnosis words behind the status word (ints; lav_gru_plan_diag copies all 16): [0] status, [1] workgroups that entered the
// kernel, [2] 1 + workgroup / [3] wave / [4] epoch of the FIRST wave that gave up, [5] its spin count, [6] the granule index
// it was waiting for, [7] the tag it last saw there, [8] microseconds between its kernel entry and the abort, [9] workgroups
// that ran to completion.  The words are cleared with the granule tags at every launch.
// POLL = 0: every wave sweeps all R*H granules itself (rounds 1-3).  POLL = 1 (H a multiple of 256): a wave polls only its quarter
// of the state and the four quarters meet in LDS - a quarter of the polling traffic per workgroup (the chip's other streams pay
// for every poll: MI355X guide, "polling-cost"), two loads instead of eight per lane and poll round.
// -DLAV_PLAN_LDS_SYNC=1: every LDS access of the persistent kernel one at a time, instruction + wait in one asm block - the build that
// was immune to matrix + LDS heavy neighbours on its CUs in round 4 (+60 us per plan).  Round 5 found the real cause of those wrong
// results - packed fp32 instructions with an op_sel bit, not LDS (common.hpp) - removed them from the library and moved the
// frame to k_plan_wave, which has no LDS at all; this kernel and the switch stay as the known victim for tools/coresidency.py.
#ifndef LAV_PLAN_LDS_SYNC
#define LAV_PLAN_LDS_SYNC 0
#endif
// -DLAV_PLAN_VARIANT=bits: experiments on this kernel as the known victim (tools/plan_variants.sh, profiles/r05_coresidency.md):
// 1 no s_sleep between poll rounds, 2 the wave sums on v_permlane swaps + DPP instead of ds_bpermute_b32 (no DS instruction but the
// kernel's own LDS reads and writes), 4 abort flag read once per step instead of three times.
#ifndef LAV_PLAN_VARIANT
#define LAV_PLAN_VARIANT 0
#endif
#if LAV_PLAN_LDS_SYNC
#define LDSR(x) lav::lds_read_sync(&(x))
#define LDSW(x, v) lav::lds_write_sync(&(x), (v))
#else
#define LDSR(x) (x)
#define LDSW(x, v) ((x) = (v))
#endif
template <int POLL, int RC>   // RC: state rows held per register pass (1 for the frame's single commanded branch, else RC)
__global__ __launch_bounds__(256) void k_plan_persistent(PlanArgs a, unsigned long long *__restrict__ gran, int *__restrict__ status,
                                                         long long spin_limit) {
    const int H = a.H, T = a.T, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = H / 64;
    const unsigned long long t_entry = wall_clock64();
    if (tid == 0) atomicAdd(status + 1, 1);
    __shared__ float gh_s[3 * PLAN_UNITS][RC];
    __shared__ float loc_s[2][RC][64][2];
    __shared__ float run_s[RC][2];
    __shared__ int abort_s;
    __shared__ float h_s[POLL ? RC : 1][POLL ? 64 * PLAN_MAXK : 1];
    const int j0 = blockIdx.x * PLAN_UNITS;
    float w[6][PLAN_MAXK], bh[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int lr = wid * 6 + q;
        const int row = (lr / PLAN_UNITS) * H + j0 + (lr % PLAN_UNITS);
        const float *wr = a.w_hh + (long)row * H;
#pragma unroll
        for (int i = 0; i < PLAN_MAXK; ++i) w[q][i] = i < nk ? wr[lane + 64 * i] : 0.f;
        bh[q] = a.b_hh[row];
    }
    float m0[PLAN_MAXK], m1[PLAN_MAXK];
#pragma unroll
    for (int i = 0; i < PLAN_MAXK; ++i) {
        m0[i] = i < nk ? a.mlp_w[lane + 64 * i] : 0.f;
        m1[i] = i < nk ? a.mlp_w[H + lane + 64 * i] : 0.f;
    }
    for (int e = tid; e < R * T; e += 256) {  // iteration 0 refines the cast waypoints
        const int r = e / T, t = e - r * T;
        int b, ci, c;
        row_decode(a, r, b, ci, c);
        const float *src = a.cast_locs + (((long)b * a.num_cmds + c) * T + t) * 2;
        float l0 = src[0], l1 = src[1];
        LDSW(loc_s[0][r][t][0], l0);
        LDSW(loc_s[0][r][t][1], l1);
        lav::lds_commit();
        lav::lds_keep(l0); lav::lds_keep(l1);
    }
    if (tid == 0) abort_s = 0;
    // this thread's gate job (tid < 8*R): unit u of state row rr
    const int gu = tid % PLAN_UNITS, grr = tid / PLAN_UNITS;
    const bool gate_thread = tid < PLAN_UNITS * R;
    int gb = 0, gci = 0, gc = 0;
    if (gate_thread) row_decode(a, grr, gb, gci, gc);
    float wih[3][4], bih[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int row = g * H + j0 + gu;
#pragma unroll
        for (int k = 0; k < 4; ++k) wih[g][k] = a.w_ih[row * 4 + k];
        bih[g] = a.b_ih[row];
    }
    const float u0 = gate_thread ? a.nxp[gb * 2 + 0] * a.ppm / a.crop * 2.f - 1.f : 0.f;
    const float u1 = gate_thread ? a.nxp[gb * 2 + 1] * a.ppm / a.crop * 2.f - 1.f : 0.f;
    float hself = 0.f;
    __syncthreads();

    int cur = 0;
    for (int it = 0; it < a.iters; ++it) {
        if (tid < R * 2) LDSW(run_s[tid >> 1][tid & 1], 0.f);
        for (int t = 0; t <= T; ++t) {  // t == T only gathers h_{T-1} to finish the iteration's waypoints
            const unsigned epoch = (unsigned)(it * T + t);  // the state published by the previous step carries this tag
            float hv[RC][PLAN_MAXK];
            if (t == 0) {
#pragma unroll
                for (int rr = 0; rr < RC; ++rr) {
                    int b, ci, c;
                    row_decode(a, min(rr, R - 1), b, ci, c);
#pragma unroll
                    for (int i = 0; i < PLAN_MAXK; ++i) hv[rr][i] = i < nk ? a.embd[(long)b * H + lane + 64 * i] : 0.f;
                }
            } else {
                const unsigned long long *g = gran + (long)((epoch - 1) & 1) * R * H;
                long long spins = 0;
                bool ok;
                constexpr int NP = POLL ? PLAN_MAXK / 4 : PLAN_MAXK;   // granules per lane and row in one poll round
                const int np = POLL ? nk / 4 : nk;
                const int base = POLL ? wid * (H / 4) : 0;              // first unit of this wave's share
                float pv[RC][NP];
                do {
                    ok = true;
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr) {
                        if (rr < R) {
#pragma unroll
                            for (int i = 0; i < NP; ++i) {
                                if (i < np) {
                                    const unsigned long long x = __hip_atomic_load(g + (long)rr * H + base + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = ok && (unsigned)(x >> 32) == epoch;
                                    pv[rr][i] = __uint_as_float((unsigned)x);
                                }
                            }
                        }
                    }
                    const unsigned long long bad = __ballot(!ok);
                    ok = bad == 0;
                    if (!ok) {
                        if (++spins > spin_limit || *(volatile int *)&abort_s) {
                            abort_s = 1;
                            if (lane == (int)__builtin_ctzll(bad)) {
                                atomicExch(status, 1);
                                if (atomicCAS(status + 2, 0, (int)blockIdx.x + 1) == 0) {   // the first wave of the grid to give up
                                    int bi = 0;
                                    unsigned bt = 0;
                                    for (int rr = R - 1; rr >= 0; --rr)
                                        for (int i = np - 1; i >= 0; --i) {
                                            const unsigned long long x = __hip_atomic_load(g + (long)rr * H + base + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            if ((unsigned)(x >> 32) != epoch) { bi = rr * H + base + lane + 64 * i; bt = (unsigned)(x >> 32); }
                                        }
                                    status[3] = wid; status[4] = (int)epoch; status[5] = (int)min(spins, 0x7fffffffll);
                                    status[6] = bi; status[7] = (int)bt;
                                    status[8] = (int)((wall_clock64() - t_entry) / 100);   // 100 MHz constant clock
                                }
                            }
                            break;
                        }
                        if (!(LAV_PLAN_VARIANT & 1)) __builtin_amdgcn_s_sleep(2);
                    }
                } while (!ok);
                if constexpr (POLL) {   // the four quarters meet in LDS (also the rendezvous of an abort: every wave leaves together)
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr)
                        if (rr < R) {
#pragma unroll
                            for (int i = 0; i < NP; ++i)
                                if (i < np) LDSW(h_s[rr][base + lane + 64 * i], pv[rr][i]);
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr)
#pragma unroll
                        for (int i = 0; i < NP; ++i) lav::lds_keep(pv[rr][i]);   // (common.hpp: pinned until the stores have left)
                    if (*(volatile int *)&abort_s) return;
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr)
                        if (rr < R) {
#pragma unroll
                            for (int i = 0; i < PLAN_MAXK; ++i) hv[rr][i] = i < nk ? LDSR(h_s[rr][lane + 64 * i]) : 0.f;
                        }
                } else {
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr)
#pragma unroll
                        for (int i = 0; i < NP; ++i) hv[rr][i] = pv[rr][i];
                }
                if (*(volatile int *)&abort_s) return;   // give up: never hang the device (k_plan_poison, next on the stream, voids the output)
                // waypoint t-1 of this iteration from h_{t-1} (every workgroup needs it as next iteration's input)
                if (wid == 0) {
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr) {
                        if (rr < R) {
                            float s0 = 0.f, s1 = 0.f;
#pragma unroll
                            for (int i = 0; i < PLAN_MAXK; ++i) {
                                s0 = fmaf(m0[i], hv[rr][i], s0);
                                s1 = fmaf(m1[i], hv[rr][i], s1);
                            }
                            if (LAV_PLAN_VARIANT & 2) {
                                s0 = wave_sum_valu(s0);
                                s1 = wave_sum_valu(s1);
                            } else {
#pragma unroll
                                for (int d = 32; d >= 1; d >>= 1) {
                                    s0 += __shfl_xor(s0, d, 64);
                                    s1 += __shfl_xor(s1, d, 64);
                                }
                            }
                            if (lane == 0) {
                                float r0 = LDSR(run_s[rr][0]) + (s0 + a.mlp_b[0]), r1 = LDSR(run_s[rr][1]) + (s1 + a.mlp_b[1]);
                                float o0 = r0 + LDSR(loc_s[cur][rr][t - 1][0]), o1 = r1 + LDSR(loc_s[cur][rr][t - 1][1]);
                                LDSW(run_s[rr][0], r0);
                                LDSW(run_s[rr][1], r1);
                                LDSW(loc_s[cur ^ 1][rr][t - 1][0], o0);
                                LDSW(loc_s[cur ^ 1][rr][t - 1][1], o1);
                                lav::lds_commit();   // (common.hpp)
                                lav::lds_keep(r0); lav::lds_keep(r1); lav::lds_keep(o0); lav::lds_keep(o1);
                                if (blockIdx.x == 0) {
                                    int b, ci, c;
                                    row_decode(a, rr, b, ci, c);
                                    float *o = a.out + ((((long)b * a.iters + it) * a.NC + ci) * T + (t - 1)) * 2;
                                    o[0] = o0;
                                    o[1] = o1;
                                }
                            }
                        }
                    }
                }
            }
            if (t == T) break;
#pragma unroll
            for (int rr = 0; rr < RC; ++rr) {
                if (rr < R) {
                    float acc[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        acc[q] = 0.f;
#pragma unroll
                        for (int i = 0; i < PLAN_MAXK; ++i) acc[q] = fmaf(w[q][i], hv[rr][i], acc[q]);
                    }
                    if (LAV_PLAN_VARIANT & 2) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) acc[q] = wave_sum_valu(acc[q]);
                    } else {
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
                            for (int q = 0; q < 6; ++q) acc[q] += __shfl_xor(acc[q], d, 64);
                    }
                    if (lane == 0) {
                        {   // committed LDS stores (common.hpp: ds_write data hazards beside matrix-heavy neighbours)
                            float gv[6];
#pragma unroll
                            for (int q = 0; q < 6; ++q) gv[q] = acc[q] + bh[q];
#pragma unroll
                            for (int q = 0; q < 6; ++q) { LDSW(gh_s[wid * 6 + q][rr], gv[q]); lav::lds_store_fence(); }
                            lav::lds_commit();
#pragma unroll
                            for (int q = 0; q < 6; ++q) lav::lds_keep(gv[q]);
                        }
                    }
                }
            }
            __syncthreads();
            if (gate_thread) {
                if (t == 0) hself = a.embd[(long)gb * H + j0 + gu];
                const float uu[4] = {u0, u1, LDSR(loc_s[cur][grr][t][0]), LDSR(loc_s[cur][grr][t][1])};
                float gi[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(wih[g][k], uu[k], acc);
                    gi[g] = acc + bih[g];
                }
                const float rg = sigmoidf_(gi[0] + LDSR(gh_s[0 * PLAN_UNITS + gu][grr]));
                const float zg = sigmoidf_(gi[1] + LDSR(gh_s[1 * PLAN_UNITS + gu][grr]));
                const float ng = tanhf(gi[2] + rg * LDSR(gh_s[2 * PLAN_UNITS + gu][grr]));
                hself = (1.f - zg) * ng + zg * hself;
                const unsigned long long x = ((unsigned long long)(epoch + 1) << 32) | __float_as_uint(hself);
                __hip_atomic_store(gran + (long)(epoch & 1) * R * H + (long)grr * H + j0 + gu, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid == 0) atomicAdd(status + 9, 1);
}
#undef LDSR
#undef LDSW

// ------------------------------------------------------------------------------------------------- plan, persistent, LDS-free (round 5)
// The default since round 5.  One WAVE per workgroup, H/8 workgroups: the wave holds all three gates of its 8 hidden units (24 rows of
// W_hh + the two rows of Linear(H->2): 208 registers per lane), so that a unit's r / z / n pre-activations meet inside the wave and
// nothing of the recurrence ever goes through LDS: no DS instruction (the cross-lane sums are v_permlane32_swap / v_permlane16_swap /
// DPP adds, not ds_bpermute), no barrier, no LDS allocation - the victim surface of DESIGN 4.4c (LDS-dependent results going wrong beside
// matrix + LDS heavy neighbours) does not exist in this kernel, and tests/test_capi_host.py disassembles the library to keep it so.
// The waypoints of the previous iteration live in registers (lane t holds waypoint t; v_readlane feeds the gate inputs).
// Same arithmetic as k_plan_step / k_plan_persistent: per lane an 8-term fmaf chain over k = lane + 64 i, then the xor butterfly
// 32, 16, 8, 4, 2, 1 - the swaps + adds below ARE that butterfly (fp addition is commutative, both partners of a pair end up with the
// same bits), 26 sums folded 32 -> 16 -> 8 registers on the way down instead of 26 x 6 exchanges: bit-identical results.
// Polling traffic per workgroup = the quarter-poll scheme's (every granule is read once per round by one wave).
// p[q][j]: this lane's partial sum of slot (quarter q, j): j < 6 -> unit 2q + j / 3, gate j % 3; (0, 6) and (0, 7): the two waypoint rows.
// Returns in z[j] the complete sum of slot (Q, j), where Q is the quarter of this lane's row of 16 lanes (kernel entry calibrates Q).
__device__ __forceinline__ void plan_fold(const float (&p)[4][6], float s0, float s1, float (&z)[8]) {
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 6; ++j) { a0[j] = fold32(p[0][j], p[2][j]); a1[j] = fold32(p[1][j], p[3][j]); }
    a0[6] = fold32(s0, 0.f);
    a0[7] = fold32(s1, 0.f);
#pragma unroll
    for (int j = 0; j < 6; ++j) z[j] = row_allsum(fold16(a0[j], a1[j]));
    z[6] = row_allsum(fold16(a0[6], 0.f));
    z[7] = row_allsum(fold16(a0[7], 0.f));
}

template <int RC, int NKT>   // NKT: H / 64 known at compile time (8 for the uniplanner's H = 512: every load unconditional), 0: any H
__global__ __launch_bounds__(64) void k_plan_wave(PlanArgs a, unsigned long long *__restrict__ gran, int *__restrict__ status, long long spin_limit) {
    const int H = NKT ? 64 * NKT : a.H, T = a.T, R = a.R;
    const int lane = threadIdx.x;
    const int nk = NKT ? NKT : H / 64;
    const unsigned long long t_entry = wall_clock64();
    if (lane == 0) atomicAdd(status + 1, 1);
    const int j0 = blockIdx.x * PLAN_UNITS;
    // which quarter's sums this lane's row ends up with, and where quarter 0's (the waypoint sums) can be read: folded once over
    // known values instead of trusting a reading of the swap instructions' lane polarity
    int Q, wp_lane;
    {
        float p[4][6], z[8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 6; ++j) p[q][j] = (float)q;
        plan_fold(p, 0.f, 0.f, z);
        Q = (int)(z[0] * (1.f / 64.f));
        wp_lane = (int)__builtin_ctzll(__ballot(Q == 0));
    }
    const int my_u = 2 * Q + (lane & 1);                        // the unit whose gates this lane evaluates (16 lanes per quarter: 8 copies each)
    const bool odd = (lane & 1) != 0, publisher = (lane & 15) < 2;
    float w[4][6][PLAN_MAXK], m0[PLAN_MAXK], m1[PLAN_MAXK];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float *wr = a.w_hh + (long)((j % 3) * H + j0 + 2 * q + j / 3) * H;
#pragma unroll
            for (int i = 0; i < PLAN_MAXK; ++i) w[q][j][i] = i < nk ? wr[lane + 64 * i] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < PLAN_MAXK; ++i) {
        m0[i] = i < nk ? a.mlp_w[lane + 64 * i] : 0.f;
        m1[i] = i < nk ? a.mlp_w[H + lane + 64 * i] : 0.f;
    }
    float wih[3][4], bih[3], bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int row = g * H + j0 + my_u;
#pragma unroll
        for (int k = 0; k < 4; ++k) wih[g][k] = a.w_ih[row * 4 + k];
        bih[g] = a.b_ih[row];
        bh[g] = a.b_hh[row];
    }
    const float mb0 = a.mlp_b[0], mb1 = a.mlp_b[1];
    // per state row: target point in crop units (uniform), the previous iteration's waypoints (lane t = waypoint t), this unit's state
    float u0[RC], u1[RC], locp0[RC], locp1[RC], locn0[RC], locn1[RC], hself[RC], hinit[RC];
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
        int b, ci, c;
        row_decode(a, min(rr, R - 1), b, ci, c);
        u0[rr] = a.nxp[b * 2 + 0] * a.ppm / a.crop * 2.f - 1.f;
        u1[rr] = a.nxp[b * 2 + 1] * a.ppm / a.crop * 2.f - 1.f;
        const float *src = a.cast_locs + (((long)b * a.num_cmds + c) * T + min(lane, T - 1)) * 2;   // iteration 0 refines the cast waypoints
        locp0[rr] = src[0];
        locp1[rr] = src[1];
        locn0[rr] = locn1[rr] = 0.f;
        hinit[rr] = a.embd[(long)b * H + j0 + my_u];
        hself[rr] = 0.f;
    }

    for (int it = 0; it < a.iters; ++it) {
        float run0[RC], run1[RC];
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) run0[rr] = run1[rr] = 0.f;
        for (int t = 0; t <= T; ++t) {   // t == T only gathers h_{T-1} to finish the iteration's waypoints
            const unsigned epoch = (unsigned)(it * T + t);   // the state published by the previous step carries this tag
            float hv[RC][PLAN_MAXK];
            if (t == 0) {
#pragma unroll
                for (int rr = 0; rr < RC; ++rr) {
                    int b, ci, c;
                    row_decode(a, min(rr, R - 1), b, ci, c);
#pragma unroll
                    for (int i = 0; i < PLAN_MAXK; ++i) hv[rr][i] = i < nk ? a.embd[(long)b * H + lane + 64 * i] : 0.f;
                }
            } else {
                const unsigned long long *g = gran + (long)((epoch - 1) & 1) * R * H;
                long long spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int rr = 0; rr < RC; ++rr) {
                        if (rr < R) {
#pragma unroll
                            for (int i = 0; i < PLAN_MAXK; ++i) {
                                if (i < nk) {
                                    const unsigned long long x = __hip_atomic_load(g + (long)rr * H + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = ok && (unsigned)(x >> 32) == epoch;
                                    hv[rr][i] = __uint_as_float((unsigned)x);
                                } else {
                                    hv[rr][i] = 0.f;
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < PLAN_MAXK; ++i) hv[rr][i] = 0.f;
                        }
                    }
                    const unsigned long long bad = __ballot(!ok);
                    if (bad == 0) break;
                    ++spins;
                    // give up: never hang the device (k_plan_poison, next on the stream, voids the output).  A wave also leaves when another
                    // one has given up (status word, looked at every 64th round), so an aborted launch ends within microseconds.
                    const bool peer_gone = (spins & 63) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (spins > spin_limit || peer_gone) {
                        if (!peer_gone && lane == (int)__builtin_ctzll(bad)) {
                            atomicExch(status, 1);
                            if (atomicCAS(status + 2, 0, (int)blockIdx.x + 1) == 0) {   // the first wave of the grid to give up
                                int bi = 0;
                                unsigned bt = 0;
                                for (int rr = R - 1; rr >= 0; --rr)
                                    for (int i = nk - 1; i >= 0; --i) {
                                        const unsigned long long x = __hip_atomic_load(g + (long)rr * H + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if ((unsigned)(x >> 32) != epoch) { bi = rr * H + lane + 64 * i; bt = (unsigned)(x >> 32); }
                                    }
                                status[3] = 0; status[4] = (int)epoch; status[5] = (int)min(spins, 0x7fffffffll);
                                status[6] = bi; status[7] = (int)bt;
                                status[8] = (int)((wall_clock64() - t_entry) / 100);   // 100 MHz constant clock
                            }
                        }
                        return;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
#pragma unroll
            for (int rr = 0; rr < RC; ++rr) {
                if (rr < R) {   // uniform
                    float p[4][6], s0 = 0.f, s1 = 0.f, z[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            float acc = 0.f;
#pragma unroll
                            for (int i = 0; i < PLAN_MAXK; ++i) acc = fmaf(w[q][j][i], hv[rr][i], acc);
                            p[q][j] = acc;
                        }
#pragma unroll
                    for (int i = 0; i < PLAN_MAXK; ++i) {
                        s0 = fmaf(m0[i], hv[rr][i], s0);
                        s1 = fmaf(m1[i], hv[rr][i], s1);
                    }
                    plan_fold(p, s0, s1, z);
                    if (t > 0) {   // waypoint t-1 of this iteration from h_{t-1} (every workgroup needs it as next iteration's input)
                        const float w0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z[6]), wp_lane));
                        const float w1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z[7]), wp_lane));
                        run0[rr] += w0 + mb0;
                        run1[rr] += w1 + mb1;
                        const float o0 = run0[rr] + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(locp0[rr]), t - 1));
                        const float o1 = run1[rr] + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(locp1[rr]), t - 1));
                        locn0[rr] = lane == t - 1 ? o0 : locn0[rr];
                        locn1[rr] = lane == t - 1 ? o1 : locn1[rr];
                        if (blockIdx.x == 0 && lane == 0) {
                            int b, ci, c;
                            row_decode(a, rr, b, ci, c);
                            float *o = a.out + ((((long)b * a.iters + it) * a.NC + ci) * T + (t - 1)) * 2;
                            o[0] = o0;
                            o[1] = o1;
                        }
                    }
                    if (t < T) {
                        if (t == 0) hself[rr] = hinit[rr];
                        const float uu[4] = {u0[rr], u1[rr], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(locp0[rr]), t)),
                                             __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(locp1[rr]), t))};
                        float gi[3], gh[3];
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc = fmaf(wih[g][k], uu[k], acc);
                            gi[g] = acc + bih[g];
                            gh[g] = (odd ? z[3 + g] : z[g]) + bh[g];
                        }
                        const float rg = sigmoidf_(gi[0] + gh[0]);
                        const float zg = sigmoidf_(gi[1] + gh[1]);
                        const float ng = tanhf(gi[2] + rg * gh[2]);
                        hself[rr] = (1.f - zg) * ng + zg * hself[rr];
                        if (publisher) {
                            const unsigned long long x = ((unsigned long long)(epoch + 1) << 32) | __float_as_uint(hself[rr]);
                            __hip_atomic_store(gran + (long)(epoch & 1) * R * H + (long)rr * H + j0 + my_u, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) { locp0[rr] = locn0[rr]; locp1[rr] = locn1[rr]; }
    }
    if (lane == 0) atomicAdd(status + 9, 1);
}
}  // namespace

extern "C" size_t lav_gru_cast_workspace_bytes(int, int, int, int, int) { return 0; }

extern "C" int lav_gru_cast(const float *embd, int B, int embd_dim, int H, int num_cmds, int T, const float *w_ih,
                            const float *w_hh, const float *b_ih, const float *b_hh, const float *mlp_w,
                            const float *mlp_b, float *out, void *, size_t, void *stream) {
    LAV_REQUIRE(B >= 0 && embd_dim > 0 && num_cmds > 0 && T > 0, "lav_gru_cast: bad sizes");
    LAV_REQUIRE(H == CAST_H, "lav_gru_cast: hidden size %d not instantiated (%d)", H, CAST_H);
    LAV_REQUIRE(B <= 65535, "lav_gru_cast: B too large");
    if (B == 0) return LAV_OK;
    LAV_REQUIRE(embd && w_ih && w_hh && b_ih && b_hh && mlp_w && mlp_b && out, "lav_gru_cast: null argument");
    const int tok = timer_begin("gru_cast", static_cast<hipStream_t>(stream));
    LAV_REQUIRE(embd_dim <= 1024, "lav_gru_cast: embedding of %d > 1024 values", embd_dim);
    hipLaunchKernelGGL(k_gru_cast, dim3(num_cmds, B), dim3(CAST_THREADS), 0, static_cast<hipStream_t>(stream), embd, embd_dim,
                       num_cmds, T, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, out, lav::batch_limit(), CastExtra{1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
    timer_end(tok, static_cast<hipStream_t>(stream));
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" int lav_embed_cast(const float *feat, int B, int embd_dim, int hw, float *embd_out, int H, int num_cmds, int T,
                              const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *mlp_w,
                              const float *mlp_b, const float *cmd_w, const float *cmd_b, float *cmds_out, const float *oris,
                              const float *locs, float *out, void *stream) {
    LAV_REQUIRE(B >= 0 && embd_dim > 0 && embd_dim <= 1024 && hw >= 1 && num_cmds > 0 && T > 0, "lav_embed_cast: bad sizes");
    LAV_REQUIRE(H == CAST_H, "lav_embed_cast: hidden size %d not instantiated (%d)", H, CAST_H);
    LAV_REQUIRE(B <= 65535, "lav_embed_cast: B too large");
    LAV_REQUIRE((cmd_w == nullptr) == (cmds_out == nullptr) && (cmd_w == nullptr) == (cmd_b == nullptr), "lav_embed_cast: cmd_w, cmd_b and cmds_out come together");
    if (B == 0) return LAV_OK;
    LAV_REQUIRE(feat && w_ih && w_hh && b_ih && b_hh && mlp_w && mlp_b && out, "lav_embed_cast: null argument");
    const int tok = timer_begin("gru_cast", static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(k_gru_cast, dim3(num_cmds, B), dim3(CAST_THREADS), 0, static_cast<hipStream_t>(stream), feat, embd_dim,
                       num_cmds, T, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, out, lav::batch_limit(), CastExtra{hw, embd_out, cmd_w, cmd_b, cmds_out, oris, locs});
    timer_end(tok, static_cast<hipStream_t>(stream));
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

extern "C" size_t lav_gru_plan_workspace_bytes(int B, int H, int num_cmds, int T) {
    // step-per-launch path: h sequence [T][R][H] floats; persistent path: 2 granule buffers [R][H] u64 + status word
    // (+ many-row path: initial state [R][H] and inputs [R][T][4])
    // Layout: [0, 256) sticky counters (launches / aborted launches since the caller zero-filled the workspace), [256, 512) the
    // status + diagnosis words of the last persistent launch, from PLAN_CTRL_BYTES on the granules resp. the step path's buffers.
    const size_t seq = ((size_t)T * B * num_cmds * H + (size_t)B * num_cmds * H + (size_t)B * num_cmds * T * 4) * sizeof(float) + 512;
    const size_t gran = lav::align_up(2 * (size_t)PLAN_RC * H * sizeof(unsigned long long), 256);
    return lav::align_up(seq > gran ? seq : gran, 256) + PLAN_CTRL_BYTES;
}

namespace {
int plan_launch(bool allow_persistent, const float *embd, const float *nxp, const float *cast_locs, int B, int H, int num_cmds,
                int T, int iters, int cmd, float pixels_per_meter, float crop_size, const float *w_ih,
                const float *w_hh, const float *b_ih, const float *b_hh, const float *mlp_w,
                const float *mlp_b, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    LAV_REQUIRE(B >= 0 && num_cmds > 0 && T > 0 && T <= 64 && iters > 0, "lav_gru_plan: bad sizes");
    LAV_REQUIRE(H % 64 == 0 && H <= 64 * PLAN_MAXK && H % PLAN_UNITS == 0, "lav_gru_plan: hidden size %d unsupported", H);
    LAV_REQUIRE(cmd >= -1 && cmd < num_cmds, "lav_gru_plan: cmd %d out of range", cmd);
    if (B == 0) return LAV_OK;
    LAV_REQUIRE(embd && nxp && cast_locs && w_ih && w_hh && b_ih && b_hh && mlp_w && mlp_b && out, "lav_gru_plan: null argument");
    PlanArgs a;
    a.embd = embd; a.nxp = nxp; a.cast_locs = cast_locs;
    a.w_ih = w_ih; a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.mlp_w = mlp_w; a.mlp_b = mlp_b;
    a.out = out; a.hseq = reinterpret_cast<float *>(static_cast<char *>(workspace) + PLAN_CTRL_BYTES);
    a.B = B; a.H = H; a.num_cmds = num_cmds; a.T = T; a.iters = iters; a.cmd = cmd;
    a.NC = cmd >= 0 ? 1 : num_cmds;
    a.R = B * a.NC;
    a.ppm = pixels_per_meter; a.crop = crop_size;
    const size_t need = (size_t)T * a.R * H * sizeof(float) + PLAN_CTRL_BYTES;
    if (!workspace || workspace_bytes < need) return lav::fail(LAV_EWORKSPACE, "lav_gru_plan: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tok = timer_begin("gru_plan", st);
    const char *impl = getenv("LAV_PLAN_IMPL");
    if (allow_persistent && a.R <= PLAN_RC && !(impl && impl[0] == 's')) {
        // persistent kernel: its H/8 workgroups (64 of 256 CUs) must become co-resident while other streams' kernels hold the
        // chip - not guaranteed by HIP, hence the bounded spins, the status / diagnosis words and the NaN poisoning
        const size_t gbytes = 2 * (size_t)a.R * H * sizeof(unsigned long long);
        LAV_REQUIRE(workspace_bytes >= lav::align_up(gbytes, 256) + PLAN_CTRL_BYTES, "lav_gru_plan: workspace too small for the persistent kernel");
        unsigned long long *gran = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + PLAN_CTRL_BYTES);
        int *sticky = static_cast<int *>(workspace);
        int *status = sticky + 64;
        // status words and tags start at 0.  A kernel, not hipMemsetAsync: a memset node captured into a HIP graph replays with a
        // garbage fill value on ROCm 7.2 (first replay fine, later ones wrote a pointer-like pattern: every graph-mode frame of
        // round 3 read a non-zero status word and poisoned a plan that had in fact completed - tools/plan_timeout_probe.py)
        const unsigned reset_words = (unsigned)((256 + lav::align_up(gbytes, 256)) / 4);
        hipLaunchKernelGGL(k_plan_reset, dim3((reset_words / 4 + 255) / 256), dim3(256), 0, st, reinterpret_cast<uint4 *>(status), reset_words / 4);
        const char *lim = getenv("LAV_PLAN_SPIN_LIMIT");   // test knob: 1 forces the time-out path
        const long long spin_limit = lim && atoll(lim) > 0 ? atoll(lim) : PLAN_SPIN_LIMIT;
        // quarter (default where H allows it) | all.  History (profiles/r04_plan_stress.txt): round 4's golden drive showed finite but
        // WRONG plans on random ticks once the 7x7 crop stems ran on the tap-pair split kernel.  tools/plan_stress.py reproduced it on
        // every launch, for BOTH polling variants, whenever this kernel's waves shared CUs with workgroups that issue matrix
        // instructions and LDS traffic (the stem kernel, or the synthetic neighbour of tools/probes/lds_hog.hip); the hidden state
        // exchanged between the workgroups was always right (instrumented), the damage was inside a workgroup: (1) a ds_write2_b64
        // whose data registers hipcc overwrote with the next instructions (common.hpp: lds_store_fence - with that alone a neighbour
        // that only issues matrix instructions is harmless), (2) something that committing the LDS stores (lds_commit / lds_keep) did
        // NOT cure - round 5: packed fp32 instructions with an op_sel bit on their second source, which hipcc's SLP vectoriser had put into
        // this kernel's reductions, go wrong in lanes 48-63 beside such neighbours (common.hpp; the library is built without them and
        // tests/test_capi_host.py scans the ISA).  Round 4 kept the aggressors away with LDS claims (now an opt-in, LAV_LDS_EXCLUSIVE=1).
        // tests/test_gpu_paint_gru.py compares every implementation with the step path, and bench.py re-computes the plans of frames
        // after its timed ones on the step path (`plan_vs_step_path_max_abs`).
        // Default: k_plan_wave (round 5, no LDS at all).  LAV_PLAN_IMPL=lds: rounds 2-4's four-wave kernel (quarter poll, LAV_PLAN_POLL=all:
        // every wave polls everything) - kept as the known victim of tools/coresidency.py, not used by the frame.
        static const char *poll_env = getenv("LAV_PLAN_POLL");
        const bool quarter = H % 256 == 0 && !(poll_env && poll_env[0] == 'a');
        if (impl && impl[0] == 'l') {
#define LAV_PLAN_CASE(P_, RC_) hipLaunchKernelGGL((k_plan_persistent<P_, RC_>), dim3(H / PLAN_UNITS), dim3(256), 0, st, a, gran, status, spin_limit)
            if (a.R == 1) { if (quarter) LAV_PLAN_CASE(1, 1); else LAV_PLAN_CASE(0, 1); }
            else { if (quarter) LAV_PLAN_CASE(1, PLAN_RC); else LAV_PLAN_CASE(0, PLAN_RC); }
#undef LAV_PLAN_CASE
        } else {
#define LAV_PLAN_CASE(RC_, NK_) hipLaunchKernelGGL((k_plan_wave<RC_, NK_>), dim3(H / PLAN_UNITS), dim3(64), 0, st, a, gran, status, spin_limit)
            if (a.R == 1) { if (H == 512) LAV_PLAN_CASE(1, 8); else LAV_PLAN_CASE(1, 0); }
            else { if (H == 512) LAV_PLAN_CASE(PLAN_RC, 8); else LAV_PLAN_CASE(PLAN_RC, 0); }
#undef LAV_PLAN_CASE
        }
        const long n_out = (long)a.B * a.iters * a.NC * T * 2;
        hipLaunchKernelGGL(k_plan_poison, dim3((unsigned)std::min<long>((n_out + 255) / 256, 64)), dim3(256), 0, st, status, sticky, a.out, n_out);
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    }
    const size_t seq_floats = lav::align_up((size_t)T * a.R * H, 64);
    const bool many_rows = a.R >= PLAN_MFMA_MIN_ROWS && workspace_bytes >= (seq_floats + (size_t)a.R * H + 64 + (size_t)a.R * T * 4) * sizeof(float) + PLAN_CTRL_BYTES &&
                           !(impl && impl[0] == 'v');   // LAV_PLAN_IMPL=v: the VALU step kernel at any size (A/B knob)
    float *h0 = a.hseq + seq_floats, *u = h0 + lav::align_up((size_t)a.R * H, 64);
    for (int it = 0; it < iters; ++it) {
        if (many_rows) {
            hipLaunchKernelGGL(k_plan_inputs, dim3((unsigned)(((long)a.R * T + 255) / 256)), dim3(256), 0, st, a, it, u, h0);
            for (int t = 0; t < T; ++t) {
                lav::GruFwdArgs f{};
                f.h_prev = t == 0 ? h0 : a.hseq + (long)(t - 1) * a.R * H; f.h_prev_stride = H;
                f.u = u + (long)t * 4; f.u_stride = (long)T * 4; f.I = 4; f.w_ih = w_ih; f.b_ih = b_ih;
                f.w_hh = w_hh; f.b_hh = b_hh;
                f.h_out = a.hseq + (long)t * a.R * H; f.h_out_stride = H;
                f.R = a.R; f.H = H;
                lav::launch_gru_fwd_step(f, st);
            }
        } else {
            for (int t = 0; t < T; ++t) hipLaunchKernelGGL(k_plan_step, dim3(H / PLAN_UNITS), dim3(256), 0, st, a, it, t);
        }
        hipLaunchKernelGGL(k_plan_out, dim3(a.R), dim3(64 * OUT_WAVES), 0, st, a, it);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}
}  // namespace

extern "C" int lav_gru_plan(const float *embd, const float *nxp, const float *cast_locs, int B, int H, int num_cmds,
                            int T, int iters, int cmd, float pixels_per_meter, float crop_size, const float *w_ih,
                            const float *w_hh, const float *b_ih, const float *b_hh, const float *mlp_w,
                            const float *mlp_b, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    return plan_launch(true, embd, nxp, cast_locs, B, H, num_cmds, T, iters, cmd, pixels_per_meter, crop_size, w_ih, w_hh, b_ih, b_hh,
                       mlp_w, mlp_b, out, workspace, workspace_bytes, stream);
}

extern "C" int lav_gru_plan_steps(const float *embd, const float *nxp, const float *cast_locs, int B, int H, int num_cmds,
                                  int T, int iters, int cmd, float pixels_per_meter, float crop_size, const float *w_ih,
                                  const float *w_hh, const float *b_ih, const float *b_hh, const float *mlp_w,
                                  const float *mlp_b, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    return plan_launch(false, embd, nxp, cast_locs, B, H, num_cmds, T, iters, cmd, pixels_per_meter, crop_size, w_ih, w_hh, b_ih, b_hh,
                       mlp_w, mlp_b, out, workspace, workspace_bytes, stream);
}

extern "C" int lav_gru_plan_status(const void *workspace, size_t workspace_bytes, int B, int H, int num_cmds, int cmd,
                                   int *h_status, void *stream) {
    LAV_REQUIRE(workspace && h_status, "lav_gru_plan_status: null argument");
    LAV_REQUIRE(B >= 1 && H > 0 && num_cmds > 0 && cmd >= -1 && cmd < num_cmds, "lav_gru_plan_status: bad sizes");
    const int R = B * (cmd >= 0 ? 1 : num_cmds);
    *h_status = 0;
    if (R > PLAN_RC) return LAV_OK;   // the step-per-launch path has no spin loops
    LAV_REQUIRE(workspace_bytes >= PLAN_CTRL_BYTES, "lav_gru_plan_status: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LAV_HIP(hipMemcpyAsync(h_status, static_cast<const char *>(workspace) + 256, sizeof(int), hipMemcpyDeviceToHost, st));
    LAV_HIP(hipStreamSynchronize(st));
    return LAV_OK;
}

extern "C" int lav_gru_plan_diag(const void *workspace, size_t workspace_bytes, int B, int H, int num_cmds, int cmd,
                                 int *h_words16, void *stream) {
    LAV_REQUIRE(workspace && h_words16, "lav_gru_plan_diag: null argument");
    LAV_REQUIRE(B >= 1 && H > 0 && num_cmds > 0 && cmd >= -1 && cmd < num_cmds, "lav_gru_plan_diag: bad sizes");
    for (int i = 0; i < 16; ++i) h_words16[i] = 0;
    LAV_REQUIRE(workspace_bytes >= PLAN_CTRL_BYTES, "lav_gru_plan_diag: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LAV_HIP(hipMemcpyAsync(h_words16, static_cast<const char *>(workspace) + 256, 10 * sizeof(int), hipMemcpyDeviceToHost, st));
    LAV_HIP(hipMemcpyAsync(h_words16 + 10, workspace, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LAV_HIP(hipStreamSynchronize(st));
    return LAV_OK;
}

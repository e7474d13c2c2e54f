// Device side of the split-operand convolution kernels (conv_split.hpp has the description and the host side): operand splitting,
// the warp-specialised workgroup body split_body<>.  Included - inside the including file's namespace - by conv.hip (bf16x6 kernels)
// and by conv_f16.hip (the LAV_CONV_F16X3 kernels, a translation unit of their own since round 6: compile time).  Expects
// MAX_CLASSES, MAX_TAPS and f32x16 from the including file.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SPLIT_NT = 8;          // most staged positions per activation-loader thread (128 threads): plane <= 1024
constexpr int SPLIT_NT_TP = 12;      // tap-pair mode (8-channel chunks: half the registers per position): plane <= 1536
constexpr int SPLIT_LOADERS = 128;

struct SplitArgs {
    const float *x;
    const unsigned char *w;          // split packed weights (bytes)
    const float *bias, *scale, *shift, *res;
    const int *n_valid;
    float *y, *partial;
    long long *trace;                // debug (LAV_SPLIT_TRACE): [workgroup][8] cycle counts
    int in_c_total, in_c_offset, cin, H, W;
    int cout, out_c_total, out_c_offset, OH, OW;
    int nchunks, nblk_total;         // cin_pad / 16, cout_pad / 32
    int QH, QW, in_s, out_s, nclasses, ksplit;
    int tw, th, tiles_x;             // 2-D tile (tw = 0: linearised pixels)
    int Wst, Wsub, ROWS, plane;      // staged patch: ROWS x Wst entries per (piece, k half), Wst = in_s * Wsub
    int tap_group, taps_per_class;
    int wring;                       // weight ring slots in LDS (one tap of the tile each): the DMA runs this many steps ahead
    int relu_pre, relu_post, sigmoid;
    float pad_value;
    int cls_ntaps[MAX_CLASSES], cls_in_oy[MAX_CLASSES], cls_in_ox[MAX_CLASSES], cls_out_oy[MAX_CLASSES], cls_out_ox[MAX_CLASSES];
    long cls_woff[MAX_CLASSES];      // byte offset of the class's weights
    int toff[MAX_TAPS];              // class c, tap t -> entry offset dy*Wst + (dx % in_s)*Wsub + dx / in_s
    // LAV_CONV_F16X3: maxima of |x| over the finite inputs in parts (k_absmax_parts, or the producers' epilogues: amax_out of the
    // launches that wrote x), the packed weights' scale (device)
    const float *f16_parts, *f16_wscale;
    int f16_nparts;
    // round 6: where this launch leaves the largest finite |y| of each workgroup (one float per workgroup, linear index; rows beyond
    // lav_batch_limit leave 0) - the next layer's f16_parts, so that no layer needs a launch to measure its input.  Null: not wanted.
    // Split-K launches leave it to k_conv_reduce.
    float *amax_out;
    int sync_off;                    // byte offset of two LDS words behind the tile buffers (the workgroup's maximum and its arrival count)
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// barrier that adds the cycles spent in it to `acc` (trace builds of the loop only)
#define SPLIT_TIMED(barrier_call, acc) do { if (a.trace) { const long long t_ = clock64(); barrier_call; acc += clock64() - t_; } else { barrier_call; } } while (0)

// x -> three bf16 pieces (round half up on the dropped bits), returned in the HIGH halves of p0..p2.  Where the round-up would
// carry into the Inf / NaN exponent (|x| within half a bf16 ulp of FLT_MAX) the first piece is truncated instead: the pieces still
// sum to x exactly.  Non-finite x: the first piece keeps Inf / NaN and the rest become NaN (Inf - Inf), so the output is NaN where
// the fp32 kernels (and the reference) propagate Inf - documented in lav_amd.h.  fp32 subnormals are flushed by the hardware.
__device__ __forceinline__ void split3(float x, unsigned &p0, unsigned &p1, unsigned &p2) {
    const unsigned u = __float_as_uint(x), r = u + 0x8000u;
    p0 = ((r & 0x7f800000u) == 0x7f800000u ? u : r) & 0xffff0000u;
    const float r1 = x - __uint_as_float(p0);          // exact
    p1 = (__float_as_uint(r1) + 0x8000u) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p1);         // exact
    p2 = __float_as_uint(r2) + 0x8000u;                // low half is dropped by the pack
}
// Two values at once on the conversion unit (round 4): v_cvt_pk_bf16_f32 rounds both to bf16 (nearest even) and packs them, so a
// pair costs 13 instructions instead of ~25 - the activation loaders' conversion of a chunk sat in the critical path of its
// barrier interval (in-kernel trace of the BEV layers: 2.4 k cycles per chunk).  q0..q2 = the three pieces of (x0, x1), x0 in the
// low half.  Exactness as split3: x - bf16(x) and the second remainder are exact, the third piece has at most 8 significant bits
// left.  The first piece of |x| > the largest finite bf16 is that bound (the remainder carries the rest): no finite input
// overflows; Inf / NaN end in NaN outputs as documented in lav_amd.h.
typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));
typedef float split_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &q0, unsigned &q1, unsigned &q2) {
    constexpr float M = 3.3895313892515355e38f;   // 0x7f7f0000
    const float c0 = __builtin_amdgcn_fmed3f(x0, -M, M), c1 = __builtin_amdgcn_fmed3f(x1, -M, M);
    q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(split_f32x2{c0, c1}, split_bf16x2));
    const float r0 = x0 - __uint_as_float(q0 << 16), r1 = x1 - __uint_as_float(q0 & 0xffff0000u);
    q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(split_f32x2{r0, r1}, split_bf16x2));
    const float s0 = r0 - __uint_as_float(q1 << 16), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
    q2 = __builtin_bit_cast(unsigned, __builtin_convertvector(split_f32x2{s0, s1}, split_bf16x2));
}
// Round 5, LAV_CONV_F16X3: two values -> two fp16 pieces each, u = h0 + h1 + O(2^-22 |u|) (round to nearest; |u| <= 32768 by the caller's
// power-of-two scale, so nothing overflows; what is below fp16's subnormal quantum 2^-24 - 2^-39 of the tensor's largest value - is lost).
// With a . b ~ a0 b0 + a0 b1 + a1 b0 that is THREE v_mfma_f32_32x32x16_f16 per 16 k-steps instead of six bf16 ones, at 22 instead of
// 24 bits per operand: the error of the dot product stays at the level of its fp32 accumulation (tests/test_gpu_conv.py).
typedef _Float16 split_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h_pair(float u0, float u1, unsigned &q0, unsigned &q1) {
    const split_f16x2 h0 = __builtin_convertvector(split_f32x2{u0, u1}, split_f16x2);
    const split_f32x2 f0 = __builtin_convertvector(h0, split_f32x2);
    const split_f16x2 h1 = __builtin_convertvector(split_f32x2{u0 - f0[0], u1 - f0[1]}, split_f16x2);
    q0 = __builtin_bit_cast(unsigned, h0);
    q1 = __builtin_bit_cast(unsigned, h1);
}
// {hi half of odd, hi half of even} -> one dword of two bf16 (even in the low half)
__device__ __forceinline__ unsigned pack_hi(unsigned even, unsigned odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }

// scheduling hint: K-th of NR groups "some MFMAs, then one LDS read"
template <int K, int NM, int NR>
struct SplitInterleave {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, (NM * (K + 1)) / NR - (NM * K) / NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        SplitInterleave<K + 1, NM, NR>::run();
    }
};
template <int NM, int NR>
struct SplitInterleave<NR, NM, NR> {
    static __device__ __forceinline__ void run() {}
};

template <int MP, int MC, int NPC = 3>
struct SplitOps {
    u32x4 a[MC][NPC], b[MP][NPC];
};

// NT: staged positions per activation-loader thread (plane <= 192 * NT); every loader thread issues all 16 * NT loads
// of a chunk unconditionally (clamped addresses, values selected afterwards): loads behind branches make the compiler
// drain the queue at each join, one memory round trip per position
// G: taps per barrier.  A barrier costs ~300 cycles of skew between the eight waves whatever the work between two of
// them; a 2x2 wave tile has 24 matrix instructions (~1000 cycles) per tap, a 1x1 tile only 6.
// TP ("tap pairs", round 4): the 16 k-steps of a matrix instruction are 8 channels of TWO taps (k half h = tap 2p + h) instead
// of 16 channels of one.  A staged position then costs 48 bytes per buffer instead of 96, which is what lets a 7x7 stride-2 stem
// (every output pixel drags ~4 input positions along) hold a 128-pixel tile double buffered: its tiles were 64 pixels x 64
// couts before - LDS-read bound, no faster than the fp32 kernel.  Same packed weights: lane (h, cout) of a fragment fetches
// its 16 bytes from tap 2p + h, channel half (chunk & 1) of the ordinary layout; the odd tap out (49 = 24 pairs + 1) is
// zeroed on its way into LDS.  A chunk is 8 channels, a step one tap pair.
// The work of one workgroup on one tile: pixel tile `by`, cout tile `bx`, class `cls`, image `n`, chunks [chunk_lo, chunk_hi) of the K
// loop; part < 0: the tile's whole K range, epilogue applied and written to y; part >= 0: raw partial sums into slab `part` of
// a.partial.  SK (stream-K, k_conv_split_sk): the function is called for one segment after the other - every role ends on one more
// LDS barrier, so that the loaders enter the next segment while the compute waves write this one out.
template <int MP, int MC, int WPX, int NT, int G, bool TP, bool SK, bool F16 = false>
__device__ __forceinline__ void split_body(const SplitArgs &a, unsigned char *smem_raw, const int bx, const int by, const int cls, const int n, const int batch,
                                           const int chunk_lo, const int chunk_hi, const int part, const long wg) {
    constexpr int WCO = 4 / WPX, NBLK = WCO * MC, PIXW = WPX * MP * 32;
    constexpr int CH = TP ? 8 : 16;                            // channels per chunk
    constexpr int NPC = F16 ? 2 : 3, NPROD = F16 ? 3 : 6;      // pieces per operand, partial products per k-block
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.n_valid && n >= *a.n_valid) {          // workgroup-uniform
        if (a.amax_out && part < 0 && tid == 0) a.amax_out[wg] = 0.f;
        return;
    }
    unsigned *s_sync = reinterpret_cast<unsigned *>(smem_raw + a.sync_off);
    if (a.amax_out && tid == 0) { s_sync[0] = 0u; s_sync[1] = 0u; }   // (wave 0 is a compute wave: ordered before their use by the loop's barriers)
    // F16: scale of the activations = the power of two that puts the tensor's largest finite |x| into [16384, 32768); every wave
    // reduces the absmax launch's per-workgroup maxima itself (a few hundred floats from L2: no LDS, no barrier)
    float inv_sx = 1.f, out_sx = 1.f, out_sw = 1.f;
    if (F16 && (wid < 4 || wid >= 6)) {   // (the weight waves need no scale)
        float m = parts_absmax(a.f16_parts, a.f16_nparts, lane);
        m = fmaxf(m, finite_abs(a.pad_value));   // (out-of-image positions hold pad_value: it must fit the scale too)
        int e = 0;
        (void)frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1): m / 2^(e - 15) in [16384, 32768)
        // (exponent floored at -100: a tensor whose largest value is below 2^-100 - or subnormal - would give a subnormal scale and an
        //  infinite reciprocal; such values flush to zero in fp16 either way.  The two scales are applied one after the other in the
        //  epilogue: their product alone can overflow for a huge activation maximum times a huge weight maximum.)
        const float sx = ldexpf(1.f, m > 0.f ? max(e, -100) - 15 : 0);
        inv_sx = 1.f / sx;                         // (a power of two: exact)
        out_sx = sx; out_sw = *a.f16_wscale;
    }
    const int ntaps_real = a.cls_ntaps[cls];
    const int ntaps = TP ? (ntaps_real + 1) >> 1 : ntaps_real;   // steps per chunk (TP: tap pairs)
    int qy0, qx0, q0 = 0;
    if (a.tw) {
        const int tx = by % a.tiles_x, ty = by / a.tiles_x;
        qy0 = ty * a.th; qx0 = tx * a.tw;
    } else {
        q0 = by * PIXW; qy0 = q0 / a.QW; qx0 = 0;
    }
    const int iy_base = qy0 * a.in_s + a.cls_in_oy[cls], in_ox = qx0 * a.in_s + a.cls_in_ox[cls];
    const int plane = a.plane;
    const int ibuf_bytes = (TP ? NPC : 2 * NPC) * plane * 16;   // [pieces][2 k halves][plane] x 16 B (TP: no k-half dimension)
    constexpr int WSLOT = NBLK * NPC * 1024;                  // one tap's weights of the tile: [cout block][piece][lane] x 16 B
    unsigned char *s_in = smem_raw, *s_w = smem_raw + 2 * ibuf_bytes;
    const int nchunk = chunk_hi - chunk_lo, nsteps = nchunk * ntaps;   // a step = one tap of one 16-channel chunk
    constexpr int WFM = 4 / G;                                 // groups of G taps the weight waves keep in flight in registers
    const int ngroups = (nsteps + G - 1) / G;                 // a group = the G taps between two barriers
    const int ngroups_pad = (ngroups + WFM - 1) / WFM * WFM;  // barriers every role executes (the weight waves' loop is unrolled by WFM)
    const int *toff = a.toff + cls * a.taps_per_class;

    if (wid == 4 || wid == 5) {
        // ------------------------------------------------------------------------------ weight waves (2 x 64 threads)
        // One tap of the tile = NBLK * 3 pieces of 1 KB ([cout block][bf16 piece][lane] x 16 B), dealt alternately to the
        // two waves; a piece travels HBM/L2 -> registers (global_load_dwordx4) -> LDS (ds_write_b128).  The asynchronous
        // LDS DMA (global_load_lds) saturates at ~25 GB/s per CU on this part; a 128-cout tile needs 12 KB per ~0.35 us of
        // matrix work.  WF steps of requests are in flight in registers (the compiler counts vmcnt for them), so the LDS
        // ring is just two slots: step i+2 is written while the compute waves fetch step i+1 and multiply step i.
        constexpr int NPW = (NBLK * NPC + 1) / 2;
        const int lw = wid - 4;
        const unsigned char *wcls = a.w + a.cls_woff[cls];
        // Everything below is unconditional straight-line code per step (a dead piece of an odd piece count repeats the
        // wave's first piece, requests past the last step re-read the last one, deposits past it land in a slot nobody
        // reads any more): with branches around the loads the compiler drains vmcnt to 0 at every deposit, i.e. pays the
        // full memory latency every step instead of once.
        unsigned rel[NPW];   // source offset of this wave's pieces relative to (tap, chunk)
        int doff[NPW];       // LDS offset of the piece inside a slot
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            int pc = lw + 2 * k;
            if (pc >= NBLK * NPC) pc = lw;
            const int b = pc / NPC, pl = pc - b * NPC;
            const int blk = min(bx * NBLK + b, a.nblk_total - 1);
            rel[k] = (unsigned)((blk * ntaps_real * a.nchunks * NPC + pl) * 1024) + lane * 16;
            if constexpr (TP) rel[k] = (unsigned)((blk * ntaps_real * a.nchunks * NPC + pl) * 1024) + l31 * 16 + half * (unsigned)(a.nchunks * NPC * 1024);
            doff[k] = pc * 1024 + lane * 16;
        }
        const unsigned tap_hop = TP ? half * (unsigned)(a.nchunks * NPC * 1024) : 0u;   // TP: the second tap of a pair sits one tap further
        u32x4 wreg[WFM][G][NPW];
        int r_t = 0, r_chunk = chunk_lo;   // next tap to request
        auto request = [&](u32x4 (&dst)[NPW]) {
            const int cch = min(r_chunk, chunk_hi - 1);
            const unsigned char *base = TP ? wcls + ((long)(2 * r_t) * a.nchunks + (cch >> 1)) * (NPC * 1024) + (cch & 1) * 512
                                           : wcls + ((long)r_t * a.nchunks + cch) * (NPC * 1024);
            // TP, odd tap count: the last pair's second tap does not exist - its lanes re-read the first (zeroed at the deposit)
            const unsigned back = TP && (ntaps_real & 1) && r_t == ntaps - 1 ? tap_hop : 0u;
#pragma unroll
            for (int k = 0; k < NPW; ++k) dst[k] = *reinterpret_cast<const u32x4 *>(base + (rel[k] - back));
            const bool wrap = r_t + 1 == ntaps;
            r_t = wrap ? 0 : r_t + 1;
            r_chunk += wrap ? 1 : 0;
        };
        // LDS ring of 3 groups: while group k is multiplied (its taps, and the first tap of group k+1, are fetched during
        // it), group k+2 is written
        int w_grp = 0, d_t = 0;   // d_t: step (tap / tap pair) within its chunk of the next deposit
        auto deposit = [&](const u32x4 (&src)[G][NPW]) {
            unsigned char *dst = s_w + w_grp * (G * WSLOT);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool dead = TP && (ntaps_real & 1) && d_t == ntaps - 1 && half;   // the tap that does not exist
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    u32x4 v = src[g][k];
                    if constexpr (TP) v = dead ? u32x4{0u, 0u, 0u, 0u} : v;
                    *reinterpret_cast<u32x4 *>(dst + g * WSLOT + doff[k]) = v;
                }
                d_t = d_t + 1 == ntaps ? 0 : d_t + 1;
            }
            w_grp = w_grp == 2 ? 0 : w_grp + 1;
        };
        long long waited = 0;
        // prologue: groups 0 and 1 into the ring, groups 2 .. WFM+1 requested (set f holds group 2 + f)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int g = 0; g < G; ++g) request(wreg[0][g]);
            deposit(wreg[0]);
        }
#pragma unroll
        for (int f = 0; f < WFM; ++f)
#pragma unroll
            for (int g = 0; g < G; ++g) request(wreg[f][g]);
        lds_barrier();
        // group k: deposit group k+2 (set k % WFM), then request group k+2+WFM into the same set
        for (int k = 0; k < ngroups_pad; k += WFM) {
#pragma unroll
            for (int f = 0; f < WFM; ++f) {
                deposit(wreg[f]);
#pragma unroll
                for (int g = 0; g < G; ++g) request(wreg[f][g]);
                SPLIT_TIMED(lds_barrier(), waited);
            }
        }
        if (a.trace && tid == 256) a.trace[wg * 8 + 7] = waited;
        if constexpr (SK) lds_barrier();
        return;
    }
    if (wid >= 6) {
        // ------------------------------------------------------------------------------ activation waves (2 x 64 threads)
        const int lt = tid - 384;
        int goff[NT];   // byte offset of the position inside a channel plane, < 0: outside the image / the staged patch
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pos = lt + SPLIT_LOADERS * i;
            const int rr = pos / a.Wst, xl = pos - rr * a.Wst;
            const int par = xl / a.Wsub, xq = xl - par * a.Wsub;
            const int iy = iy_base + rr, ix = in_ox + xq * a.in_s + par;
            const bool ok = pos < plane && rr < a.ROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            goff[i] = ok ? (iy * a.W + ix) * 4 : -1;
        }
        const long cplane = (long)a.H * a.W;
        const char *xin = reinterpret_cast<const char *>(a.x + ((long)n * a.in_c_total + a.in_c_offset) * cplane);
        float v[NT][CH];
        auto issue_loads_to = [&](float (&v)[NT][CH], int sc) __attribute__((always_inline)) {
            const int ci0 = (chunk_lo + sc) * CH;
            const int nc = min(CH, a.cin - ci0);   // wave-uniform (TP: the host requires cin % 16 == 0, so nc = 8)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const char *pc = xin + ((long)ci0 + min(c, nc - 1)) * cplane * 4;   // scalar base, 32-bit lane offset
#pragma unroll
                for (int i = 0; i < NT; ++i) v[i][c] = *reinterpret_cast<const float *>(pc + (unsigned)max(goff[i], 0));
            }
        };
        auto convert_store_from = [&](const float (&v)[NT][CH], int sc) __attribute__((always_inline)) {
            const int ci0 = (chunk_lo + sc) * CH;
            const int nc = min(CH, a.cin - ci0);
            unsigned char *dst = s_in + (sc & 1) * ibuf_bytes;
            constexpr int KH = CH / 8;   // k halves of a chunk
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int pos = lt + SPLIT_LOADERS * i;
                const bool ok = goff[i] >= 0;
                u32x4 q[NPC][KH];
#pragma unroll
                for (int h = 0; h < KH; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c0 = 8 * h + 2 * e;
                        const float x0 = c0 < nc ? (ok ? v[i][c0] : a.pad_value) : 0.f;
                        const float x1 = c0 + 1 < nc ? (ok ? v[i][c0 + 1] : a.pad_value) : 0.f;
                        if constexpr (F16) {
                            unsigned p0, p1;
                            split2h_pair(x0 * inv_sx, x1 * inv_sx, p0, p1);
                            q[0][h][e] = p0; q[1][h][e] = p1;
                        } else {
                            unsigned p0, p1, p2;
                            split3_pair(x0, x1, p0, p1, p2);
                            q[0][h][e] = p0; q[1][h][e] = p1; q[NPC - 1][h][e] = p2;
                        }
                    }
                if (pos < plane) {
#pragma unroll
                    for (int pl = 0; pl < NPC; ++pl)
#pragma unroll
                        for (int h = 0; h < KH; ++h) *reinterpret_cast<u32x4 *>(dst + ((pl * KH + h) * plane + pos) * 16) = q[pl][h];
                }
            }
        };
        // Chunk c+2 is converted into the buffer of chunk c during the LAST tap of chunk c (the compute waves fetched that
        // tap's operands a step earlier) from registers whose loads were issued a whole chunk before; the same registers
        // then take the loads of chunk c+3.
        // During group k the compute waves fetch taps kG+1 .. (k+1)G, so the chunks below the one of tap kG+1 are free, and the
        // barrier that closes the group promises the chunks up to tap (k+2)G.  Host: 2G <= taps + 1, i.e. those 2G taps touch
        // at most two chunks - the two LDS buffers.  A chunk is converted as soon as its buffer is free, from registers whose
        // loads were issued when the previous chunk was converted (about a chunk of matrix work earlier).
        auto issue_loads = [&](int sc) __attribute__((always_inline)) { issue_loads_to(v, sc); };
        auto convert_store = [&](int sc) __attribute__((always_inline)) { convert_store_from(v, sc); };
        long long waited = 0, conv = 0;
        if constexpr (NT <= 2) {   // small patches: the first two chunks travel together (one memory latency before the first tap)
            float v2[NT][CH];
            if (nchunk > 0) issue_loads_to(v, 0);
            if (nchunk > 1) issue_loads_to(v2, 1);
            if (nchunk > 0) convert_store_from(v, 0);
            if (nchunk > 2) issue_loads_to(v, 2);
            if (nchunk > 1) convert_store_from(v2, 1);
        } else {
            if (nchunk > 0) { issue_loads(0); convert_store(0); }
            if (nchunk > 1) { issue_loads(1); convert_store(1); }
            if (nchunk > 2) issue_loads(2);
        }
        int conv_next = 2;
        lds_barrier();
        int m_c = 0, m_t = 1;   // chunk / tap of micro-step kG+1
        if (m_t >= ntaps) { m_t -= ntaps; ++m_c; }
        for (int k = 0; k < ngroups; ++k) {
            if (conv_next < nchunk && conv_next <= m_c + 1) {
                SPLIT_TIMED(convert_store(conv_next), conv);
                if (conv_next + 1 < nchunk) issue_loads(conv_next + 1);
                ++conv_next;
            }
            SPLIT_TIMED(lds_barrier(), waited);
            m_t += G;
            while (m_t >= ntaps) { m_t -= ntaps; ++m_c; }
        }
        for (int k = ngroups; k < ngroups_pad; ++k) lds_barrier();
        if (a.trace && tid == 384) { a.trace[wg * 8 + 5] = conv; a.trace[wg * 8 + 6] = waited; }
        if constexpr (SK) lds_barrier();
        return;
    }
    // -------------------------------------------------------------------------------------- compute waves
    const int wp = wid % WPX, wc = wid / WPX;
    const int Q = a.QH * a.QW;
    int pqy[MP], pqx[MP], base[MP];
    bool pvalid[MP];
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) {
        const int local = (wp * MP + mp) * 32 + l31;
        if (a.tw) {
            const int ty = local / a.tw, tx = local - ty * a.tw;
            pqy[mp] = qy0 + ty; pqx[mp] = qx0 + tx;
            pvalid[mp] = pqy[mp] < a.QH && pqx[mp] < a.QW;
            pqy[mp] = min(pqy[mp], a.QH - 1); pqx[mp] = min(pqx[mp], a.QW - 1);
        } else {
            const int q = q0 + local;
            pvalid[mp] = q < Q;
            const int qc = min(q, Q - 1);
            pqy[mp] = qc / a.QW; pqx[mp] = qc - pqy[mp] * a.QW;
        }
        base[mp] = ((pqy[mp] - qy0) * a.in_s * a.Wst + (pqx[mp] - qx0) + (TP ? 0 : half * plane)) * 16;
    }
    f32x16 acc[MC][MP];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int mp = 0; mp < MP; ++mp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mc][mp][r] = 0.f;

    const int pstride = (TP ? 1 : 2) * plane * 16;   // bytes between the pieces of the input buffer
    auto load_ops = [&](SplitOps<MP, MC, NPC> &o, const unsigned char *bin, const unsigned char *bw, int to) {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
            for (int pl = 0; pl < NPC; ++pl) o.a[mc][pl] = *reinterpret_cast<const u32x4 *>(bw + (mc * NPC + pl) * 1024);
#pragma unroll
        for (int mp = 0; mp < MP; ++mp)
#pragma unroll
            for (int pl = 0; pl < NPC; ++pl) o.b[mp][pl] = *reinterpret_cast<const u32x4 *>(bin + pl * pstride + base[mp] + to * 16);
    };
    auto mma = [&](const SplitOps<MP, MC, NPC> &o) {
        // smallest terms first; consecutive instructions go to different accumulators
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};   // fp16 pieces: a1 b0, a0 b1, a0 b0
#pragma unroll
        for (int k = 0; k < NPROD; ++k)
#pragma unroll
            for (int mc = 0; mc < MC; ++mc)
#pragma unroll
                for (int mp = 0; mp < MP; ++mp) {
                    if constexpr (F16)
                        acc[mc][mp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, o.a[mc][HA[k]]), __builtin_bit_cast(f16x8, o.b[mp][HB[k]]),
                                                                             acc[mc][mp], 0, 0, 0);
                    else
                        acc[mc][mp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.a[mc][PA[k]]), __builtin_bit_cast(bf16x8, o.b[mp][PB[k]]),
                                                                              acc[mc][mp], 0, 0, 0);
                }
    };
    // tap offsets: lane t of one VGPR holds tap t's offset, fetched with v_readlane (a scalar load per tap would share
    // lgkmcnt with the LDS reads and drain the operand pipeline at every tap)
    const int toff_lane = toff[min(lane, a.taps_per_class - 1)];
    auto tap_off = [&](int t) {
        if constexpr (TP) {   // k half h multiplies tap 2t + h (the tap that does not exist has zero weights: any staged entry will do)
            const int t0 = __builtin_amdgcn_readlane(toff_lane, 2 * t), t1 = __builtin_amdgcn_readlane(toff_lane, min(2 * t + 1, ntaps_real - 1));
            return half ? t1 : t0;
        } else {
            return __builtin_amdgcn_readlane(toff_lane, t);
        }
    };
    long long waited = 0;
    if (a.trace && tid == 0) a.trace[wg * 8 + 0] = clock64();
    lds_barrier();
    if (a.trace && tid == 0) a.trace[wg * 8 + 1] = clock64();
    {
        // operands of the NEXT tap: tap f_t of chunk parity f_par, weight slot f_slot of the 3G-slot ring
        int f_t = 0, f_par = 0, f_slot = 0;
        const unsigned char *bw_lane = s_w + (wc * MC * NPC * 64 + lane) * 16;
        auto fetch = [&](SplitOps<MP, MC, NPC> &o) {
            load_ops(o, s_in + f_par * ibuf_bytes, bw_lane + f_slot * WSLOT, tap_off(f_t));
            if (++f_t == ntaps) { f_t = 0; f_par ^= 1; }
            if (++f_slot == 3 * G) f_slot = 0;
        };
        SplitOps<MP, MC, NPC> o0, o1;
        if (nsteps > 0) fetch(o0);
        // U taps per iteration (static register sets, a barrier after every G-th).  The operand fetch is unconditional - past
        // the last tap it reads stale LDS that nobody uses - so that fetch and matrix instructions share one basic block, and
        // the LDS reads of the next tap are spread between this tap's matrix instructions (one read per NM / NR of them):
        // issued as one burst they leave the matrix pipe idle.
        constexpr int U = G > 2 ? G : 2;
        int i = 0, bars = 0;
        for (; i + U <= nsteps; i += U) {
#pragma unroll
            for (int j = 0; j < U; j += 2) {
                fetch(o1);
                mma(o0);
                SplitInterleave<0, NPROD * MP * MC, NPC * (MP + MC)>::run();
                if ((j + 1) % G == 0) { SPLIT_TIMED(lds_barrier(), waited); ++bars; }
                fetch(o0);
                mma(o1);
                SplitInterleave<0, NPROD * MP * MC, NPC * (MP + MC)>::run();
                if ((j + 2) % G == 0) { SPLIT_TIMED(lds_barrier(), waited); ++bars; }
            }
        }
        // tail: fewer than U taps left (the sets keep alternating from o0)
#pragma unroll
        for (int j = 0; j < U - 1; ++j) {
            if (i + j < nsteps) {
                if (j % 2 == 0) { fetch(o1); mma(o0); } else { fetch(o0); mma(o1); }
                if ((i + j + 1) % G == 0) { lds_barrier(); ++bars; }
            }
        }
        for (; bars < ngroups_pad; ++bars) lds_barrier();
    }
    if constexpr (SK) lds_barrier();   // the LDS of this segment is free: the loaders go on to the next one
    if (a.trace && tid == 0) { a.trace[wg * 8 + 2] = clock64(); a.trace[wg * 8 + 4] = waited; }

    // -------------------------------------------------------------------------------------- epilogue (as k_conv)
    const int cb = (bx * NBLK + wc * MC) * 32;
    const int out_oy = a.cls_out_oy[cls], out_ox = a.cls_out_ox[cls];
    if (part >= 0) {
        const long plane_o = (long)a.OH * a.OW;
        float *pbase = a.partial + ((long)part * batch + n) * a.cout * plane_o;
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                const int oy = pqy[mp] * a.out_s + out_oy, ox = pqx[mp] * a.out_s + out_ox;
                const bool pix_ok = pvalid[mp] && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb + mc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    // (F16: the accumulators are in units of activation scale x weight scale; the slabs hold true values)
                    if (pix_ok && co < a.cout) pbase[co * plane_o + (long)oy * a.OW + ox] = F16 ? acc[mc][mp][r] * out_sx * out_sw : acc[mc][mp][r];
                }
            }
        return;
    }
    // epilogue vectors: always loaded (from a valid address) and selected afterwards - no loads behind branches
    const bool has_bias = a.bias != nullptr, has_aff = a.scale != nullptr, has_res = a.res != nullptr;
    float wmax = 0.f;
    const float *bias_p = has_bias ? a.bias : a.x, *scale_p = has_aff ? a.scale : a.x, *shift_p = has_aff ? a.shift : a.x;
    const float *res_p = has_res ? a.res : a.y;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
        float bv[16], sv[16], tv[16];
        int cov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb + mc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            cov[r] = co;
            const int cc = has_bias || has_aff ? min(co, a.cout - 1) : 0;
            bv[r] = bias_p[cc]; sv[r] = scale_p[cc]; tv[r] = shift_p[cc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bv[r] = has_bias ? bv[r] : 0.f; sv[r] = has_aff ? sv[r] : 1.f; tv[r] = has_aff ? tv[r] : 0.f;
        }
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) {
            const int oy = pqy[mp] * a.out_s + out_oy, ox = pqx[mp] * a.out_s + out_ox;
            const bool pix_ok = pvalid[mp] && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
            const long pix = (long)oy * a.OW + ox;
            const long cbase = ((long)n * a.out_c_total + a.out_c_offset) * a.OH * a.OW;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long idx = cbase + (long)min(cov[r], a.cout - 1) * a.OH * a.OW + (pix_ok ? pix : 0);
                rv[r] = res_p[has_res ? idx : 0];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v;
                if constexpr (F16) v = fmaf(acc[mc][mp][r] * out_sx, out_sw, bv[r]);   // the accumulators are in units of (activation scale x weight scale)
                else v = acc[mc][mp][r] + bv[r];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                v = fmaf(v, sv[r], tv[r]);
                v += has_res ? rv[r] : 0.f;
                if (a.relu_post) v = v > 0.f ? v : 0.f;
                if (a.sigmoid && cov[r] >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
                if (pix_ok && cov[r] < a.cout) {
                    a.y[cbase + (long)cov[r] * a.OH * a.OW + pix] = v;
                    wmax = fmaxf(wmax, finite_abs(v));
                }
            }
        }
    }
    if (a.amax_out) {
        // the workgroup's largest finite |y|: ds_max per compute wave, the fourth arrival (LDS executes a wave's operations in order,
        // so every ds_max precedes its wave's ds_add) writes the workgroup's slot - no barrier, no global atomic, nothing to zero
        wmax = wave_finite_absmax(wmax);
        if (lane == 0) {
            (void)__hip_atomic_fetch_max(&s_sync[0], __float_as_uint(wmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned arrived = __hip_atomic_fetch_add(&s_sync[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (arrived == 3u) a.amax_out[wg] = __uint_as_float(__hip_atomic_load(&s_sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
    }
}

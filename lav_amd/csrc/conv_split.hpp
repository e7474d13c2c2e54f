// fp32 convolution on the bf16 matrix cores by operand splitting - included by conv.hip (inside its namespace).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 matrix rate).  Every fp32 value is
// the exact sum of three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significant bits; x1 = bf16(x - x0), ...), so
//     a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a2b0 + a1b1) + O(2^-24 |ab|)
// is six v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulate) per 16 k-steps instead of eight fp32 MFMAs of 64
// cycles: 192 instead of 512 matrix-pipe cycles, with the error of an fp32 dot product (tools/probes/split_mfma_probe.hip
// on MI355X, K = 1152, 2^12 dynamic range: max |err| / sum|ab| 3.1e-7 against 6.3e-7 for the fp32 MFMA chain).  fp32
// range is kept (bf16 has the fp32 exponent); fp32 SUBNORMAL inputs are flushed by the bf16 matrix pipe.
//
// Kernel structure (one workgroup = 8 waves):
//   waves 0-3  compute: MP x MC tiles of 32 pixels x 32 couts each (wave grid WPX x 4/WPX), operands from LDS with one
//              ds_read_b128 per (plane, tile row/column) and tap, software pipelined over taps
//   waves 4-7  loaders.  Weights: pre-split packed weights [cout block][tap][16-channel chunk][plane][lane][8 bf16] move
//              HBM/L2 -> LDS by asynchronous DMA (global_load_lds, 1 KB per instruction, pieces dealt round-robin to the
//              four waves), one or two tap groups ahead (ring of 2-3 buffers).  Activations: NCHW fp32 -> registers
//              (coalesced along the row, all loads of a chunk in flight, issued a whole chunk before they are needed) ->
//              three bf16 pieces -> LDS as [plane][k half][position][8 channels] (16 bytes per entry = one lane's B
//              operand), double buffered per 16-channel chunk.  Stride-s layers are staged column-parity split so
//              that the 32 lanes of a tap read consecutive entries.
//   barriers   one per (chunk, tap group); they order LDS only: the loaders wait with a COUNTED s_waitcnt vmcnt for the
//              DMA the next stage needs, their register loads stay in flight across barriers.
// The pixel tile is a 2-D block of the output grid (th rows x tw columns, tw a multiple of 32) - a 3x3 layer stages
// (th+2)(tw+2) positions for th*tw pixels instead of three full-width rows per 128 - or PIXW consecutive linearised
// pixels for maps narrower than 32.  Epilogue, split-K partials and transposed-convolution classes: as k_conv.
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
    // LAV_CONV_F16X3: per-workgroup maxima of |x| over the finite inputs (k_absmax_parts), the packed weights' scale (device)
    const float *f16_parts, *f16_wscale;
    int f16_nparts;
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
    static_assert(!(F16 && TP), "the fp16 two-piece mode is built for the ordinary 16-channel chunks");
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.n_valid && n >= *a.n_valid) return;   // workgroup-uniform
    // F16: scale of the activations = the power of two that puts the tensor's largest finite |x| into [16384, 32768); every wave
    // reduces the absmax launch's per-workgroup maxima itself (a few hundred floats from L2: no LDS, no barrier)
    float inv_sx = 1.f, out_scale = 1.f;
    if constexpr (F16) {
        float m = 0.f;
        for (int i = lane; i < a.f16_nparts; i += 64) m = fmaxf(m, a.f16_parts[i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        int e = 0;
        (void)frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1): m / 2^(e - 15) in [16384, 32768)
        const float sx = ldexpf(1.f, m > 0.f ? e - 15 : 0);
        inv_sx = 1.f / sx;                         // (a power of two: exact)
        out_scale = sx * *a.f16_wscale;
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
    const int ibuf_bytes = (TP ? 3 : 2 * NPC) * plane * 16;   // [pieces][2 k halves][plane] x 16 B (TP: no k-half dimension)
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
            if constexpr (TP) rel[k] = (unsigned)((blk * ntaps_real * a.nchunks * 3 + pl) * 1024) + l31 * 16 + half * (unsigned)(a.nchunks * 3072);
            doff[k] = pc * 1024 + lane * 16;
        }
        const unsigned tap_hop = TP ? half * (unsigned)(a.nchunks * 3072) : 0u;   // TP: the second tap of a pair sits one tap further
        u32x4 wreg[WFM][G][NPW];
        int r_t = 0, r_chunk = chunk_lo;   // next tap to request
        auto request = [&](u32x4 (&dst)[NPW]) {
            const int cch = min(r_chunk, chunk_hi - 1);
            const unsigned char *base = TP ? wcls + ((long)(2 * r_t) * a.nchunks + (cch >> 1)) * 3072 + (cch & 1) * 512
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
                    if (pix_ok && co < a.cout) pbase[co * plane_o + (long)oy * a.OW + ox] = acc[mc][mp][r];
                }
            }
        return;
    }
    // epilogue vectors: always loaded (from a valid address) and selected afterwards - no loads behind branches
    const bool has_bias = a.bias != nullptr, has_aff = a.scale != nullptr, has_res = a.res != nullptr;
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
                if constexpr (F16) v = fmaf(acc[mc][mp][r], out_scale, bv[r]);   // the accumulators are in units of (activation scale x weight scale)
                else v = acc[mc][mp][r] + bv[r];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                v = fmaf(v, sv[r], tv[r]);
                v += has_res ? rv[r] : 0.f;
                if (a.relu_post) v = v > 0.f ? v : 0.f;
                if (a.sigmoid && cov[r] >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
                if (pix_ok && cov[r] < a.cout) a.y[cbase + (long)cov[r] * a.OH * a.OW + pix] = v;
            }
        }
    }
}

template <int MP, int MC, int WPX, int NT, int G, bool TP = false>
__global__ __launch_bounds__(512) void k_conv_split(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ks = blockIdx.z % a.ksplit;
    const int cls = (blockIdx.z / a.ksplit) % a.nclasses, n = blockIdx.z / (a.ksplit * a.nclasses);
    const int nchunks_k = TP ? 2 * a.nchunks : a.nchunks;
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    split_body<MP, MC, WPX, NT, G, TP, false>(a, smem_raw, blockIdx.x, blockIdx.y, cls, n, gridDim.z / (a.ksplit * a.nclasses), ks * nchunks_k / a.ksplit,
                                              (ks + 1) * nchunks_k / a.ksplit, a.ksplit > 1 ? ks : -1, wg);
}

// LAV_CONV_F16X3 (round 5): the same kernel on two fp16 pieces per operand and three products (split_body<..., F16 = true>), for
// stride-1 single-class layers on the 2x2/w2 G = 2 tile - the head convolution.  k_absmax_parts runs first: workgroup g writes the
// largest FINITE |x| of its share of the layer's input channels to parts[g] (no atomics, nothing to zero); Inf / NaN inputs do not
// enter the scale and propagate through the data path as they are.
constexpr int F16_PARTS = 512;
__global__ __launch_bounds__(256) void k_absmax_parts(const float *__restrict__ x, int batch, int in_c_total, int in_c_offset, int cin, long plane,
                                                      float *__restrict__ parts) {
    __shared__ float s_m[4];
    const long per_img = (long)cin * plane, total = (long)batch * per_img;
    float m = 0.f;
    if ((plane & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long total4 = total >> 2, per4 = per_img >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const long n = i / per4, r = i - n * per4;
            const float4 v = *reinterpret_cast<const float4 *>(x + ((long)n * in_c_total + in_c_offset) * plane + 4 * r);
            const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
            m = fmaxf(m, a0 <= 3.4028235e38f ? a0 : 0.f); m = fmaxf(m, a1 <= 3.4028235e38f ? a1 : 0.f);
            m = fmaxf(m, a2 <= 3.4028235e38f ? a2 : 0.f); m = fmaxf(m, a3 <= 3.4028235e38f ? a3 : 0.f);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const long n = i / per_img, r = i - n * per_img;
            const float a0 = fabsf(x[((long)n * in_c_total + in_c_offset) * plane + r]);
            m = fmaxf(m, a0 <= 3.4028235e38f ? a0 : 0.f);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

template <int MP, int MC, int WPX, int NT, int G>
__global__ __launch_bounds__(512) void k_conv_split_f16(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    split_body<MP, MC, WPX, NT, G, false, false, true>(a, smem_raw, blockIdx.x, blockIdx.y, 0, blockIdx.z, gridDim.z, 0, a.nchunks, -1, 0);
}

// Stream-K launch of a single-image, single-class layer (round 5).  The head convolution's 400 tiles ran as two rounds of a 256-CU
// chip with the second round 56 % full (0.78 of the tile time wasted).  Here W persistent workgroups (one per CU) share
// the layer's U = tiles x chunks units of K work evenly: workgroup i takes units [i U / W, (i + 1) U / W) of the linear order (tile,
// chunk), i.e. the tail of one tile, whole tiles, and the head of another.  With U / W >= chunks per tile a tile is cut at most once:
// its head part (chunks [0, c)) goes to slab 0 of a.partial, its tail part to slab 1, and k_conv_sk_fixup adds the two in that order
// and applies the epilogue - fixed cuts, fixed order: bit-reproducible.  Whole tiles take the ordinary epilogue.
template <int MP, int MC, int WPX, int NT, int G>
__global__ __launch_bounds__(512) void k_conv_split_sk(SplitArgs a, int nbx, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nck = a.nchunks;
    const long U = (long)ntiles * nck;
    const int W = gridDim.x, npx = ntiles / nbx;
    // The tiles are ordered cout tile first (tile = bx * npx + by), and the workgroups of one XCD (workgroup b is dispatched to XCD b % 8)
    // take neighbouring ranges of that order: an XCD's L2 then streams the weights of ONE cout tile (2.6 MB of the head convolution's
    // 5.3 MB; both do not fit its 4 MB - the first version, workgroup b on range b, was slower than whole tiles).
    int j = blockIdx.x;
    if (8 % nbx == 0 && W % 8 == 0) {
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        j = (x % nbx) * (W / nbx) + (x / nbx) * (W / 8) + slot;
    }
    long u = (long)j * U / W;
    const long u1 = (long)(j + 1) * U / W;
    while (u < u1) {
        const int tile = (int)(u / nck), c_lo = (int)(u - (long)tile * nck);
        const int c_hi = (int)(u1 - u < (long)(nck - c_lo) ? c_lo + (u1 - u) : nck);
        const int part = c_lo == 0 && c_hi == nck ? -1 : (c_lo == 0 ? 0 : 1);
        split_body<MP, MC, WPX, NT, G, false, true>(a, smem_raw, tile / npx, tile % npx, 0, 0, 1, c_lo, c_hi, part, 0);
        u += c_hi - c_lo;
    }
}

struct SkFixupArgs {
    const float *partial, *bias, *scale, *shift, *res;
    const int *n_valid;
    float *y;
    int cout, out_c_total, out_c_offset, OH, OW;
    int tw, th, tiles_x, couts_per_tile, nbx, nck, ntiles, W;
    int relu_pre, relu_post, sigmoid;
};
// y = epilogue(slab 0 + slab 1) for the outputs of the tiles that k_conv_split_sk cut (the same arithmetic as k_conv_reduce)
__global__ __launch_bounds__(256) void k_conv_sk_fixup(SkFixupArgs a) {
    const long plane_o = (long)a.OH * a.OW, total = (long)a.cout * plane_o;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    if (a.n_valid && *a.n_valid < 1) return;
    const int co = (int)(e / plane_o);
    const int pix = (int)(e - (long)co * plane_o), oy = pix / a.OW, ox = pix - oy * a.OW;
    const int tile = (co / a.couts_per_tile) * (a.ntiles / a.nbx) + (oy / a.th) * a.tiles_x + ox / a.tw;
    const long U = (long)a.ntiles * a.nck, lo = (long)tile * a.nck;
    const long i0 = lo * a.W / U + 1, b = i0 * U / a.W;   // the first workgroup boundary past the tile's first unit
    if (!(i0 < a.W && b > lo && b < lo + a.nck)) return;  // the tile was not cut
    float v = a.partial[e] + a.partial[total + e];
    if (a.bias) v += a.bias[co];
    if (a.relu_pre) v = v > 0.f ? v : 0.f;
    if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
    const long idx = ((long)a.out_c_offset + co) * plane_o + pix;
    if (a.res) v += a.res[idx];
    if (a.relu_post) v = v > 0.f ? v : 0.f;
    if (a.sigmoid && co >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
    a.y[idx] = v;
}

// ------------------------------------------------------------------------------------------------ host side
// bytes of the split packed weights of one plan
inline size_t split_weight_bytes(const Plan &p) {
    size_t taps = 0;
    for (auto &t : p.taps) taps += t.size();
    return taps * (size_t)(p.cout_pad / 32) * (p.cin_pad / 16) * 3 * 1024;
}

// LAV_CONV_F16X3 packing: [cout block][tap][chunk][piece 2][lane = khalf*32 + cout%32][8 channels] fp16 of w / s_w, then the scale s_w
// (a power of two that puts the largest |w| into [16384, 32768)) as one float at a 16-byte aligned offset
inline size_t split_weight_bytes_f16(const Plan &p) {
    size_t taps = 0;
    for (auto &t : p.taps) taps += t.size();
    return taps * (size_t)(p.cout_pad / 32) * (p.cin_pad / 16) * 2 * 1024;
}
inline bool f16x3_layer(const lav_conv &c, const Plan &p) {   // the layers the mode is built for: stride-1 single-class 3x3-like convolutions with whole chunks
    return !c.transposed && p.nclasses == 1 && c.stride == 1 && c.cin % 16 == 0 && c.cin >= 64 && c.cout >= 128;
}
inline void split_pack_weights_f16(const lav_conv &c, const Plan &p, const float *h_weight, unsigned char *out) {
    const int nblk = p.cout_pad / 32, nchunks = p.cin_pad / 16;
    const size_t nw = (size_t)c.cout * c.cin * c.kh * c.kw;
    float m = 0.f;
    for (size_t i = 0; i < nw; ++i) { const float v = fabsf(h_weight[i]); if (v <= 3.4028235e38f && v > m) m = v; }
    int e = 0;
    (void)frexpf(m, &e);
    const float sw = ldexpf(1.f, m > 0.f ? e - 15 : 0), inv = 1.f / sw;
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    const auto &t = p.taps[0];
    parallel_for(nblk, [&](int blk) {
        for (size_t ti = 0; ti < t.size(); ++ti)
            for (int ch = 0; ch < nchunks; ++ch)
                for (int lane = 0; lane < 64; ++lane)
                    for (int el = 0; el < 8; ++el) {
                        const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + el;
                        float w = 0.f;
                        if (co < c.cout && ci < c.cin) w = h_weight[(((size_t)co * c.cin + ci) * c.kh + t[ti].ky) * c.kw + t[ti].kx] * inv;
                        const _Float16 h0 = (_Float16)w, h1 = (_Float16)(w - (float)h0);
                        const size_t frag = ((((size_t)blk * t.size() + ti) * nchunks + ch) * 2) * 512;
                        o[frag + lane * 8 + el] = h0;
                        o[frag + 512 + lane * 8 + el] = h1;
                    }
    });
    memcpy(out + split_weight_bytes_f16(p), &sw, sizeof(float));
}

inline unsigned short bf16_round(float x, float &rest) {
    unsigned u;
    memcpy(&u, &x, 4);
    const unsigned r = u + 0x8000u;
    u = ((r & 0x7f800000u) == 0x7f800000u ? u : r) & 0xffff0000u;   // as split3: no round-up into the Inf exponent
    float b;
    memcpy(&b, &u, 4);
    rest = x - b;
    return (unsigned short)(u >> 16);
}

// [class][cout block][tap][chunk][piece][lane = khalf*32 + cout%32][8 channels] bf16
inline void split_pack_weights(const lav_conv &c, const Plan &p, const float *h_weight, unsigned char *out) {
    const int nblk = p.cout_pad / 32, nchunks = p.cin_pad / 16;
    unsigned short *o = reinterpret_cast<unsigned short *>(out);
    size_t cls_off = 0;   // in u16
    for (int cls = 0; cls < p.nclasses; ++cls) {
        const auto &t = p.taps[cls];
        parallel_for(nblk, [&, cls_off](int blk) {
            for (size_t ti = 0; ti < t.size(); ++ti)
                for (int ch = 0; ch < nchunks; ++ch)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + e;
                            float w = 0.f;
                            if (co < c.cout && ci < c.cin)
                                w = c.transposed ? h_weight[(((size_t)ci * c.cout + co) * c.kh + t[ti].ky) * c.kw + t[ti].kx]
                                                 : h_weight[(((size_t)co * c.cin + ci) * c.kh + t[ti].ky) * c.kw + t[ti].kx];
                            float r1, r2, r3;
                            const unsigned short p0 = bf16_round(w, r1), p1 = bf16_round(r1, r2), p2 = bf16_round(r2, r3);
                            const size_t frag = cls_off + ((((size_t)blk * t.size() + ti) * nchunks + ch) * 3) * 512;
                            o[frag + 0 * 512 + lane * 8 + e] = p0;
                            o[frag + 1 * 512 + lane * 8 + e] = p1;
                            o[frag + 2 * 512 + lane * 8 + e] = p2;
                        }
        });
        cls_off += t.size() * (size_t)nblk * nchunks * 3 * 512;
    }
}

struct SplitPlan {
    bool ok;
    int MP, MC, WPX, tw, th, tiles_x, tiles, Wst, Wsub, ROWS, plane, tap_group, ksplit, wring;
    size_t lds;
    double cost;
    int tp;   // tap-pair mode (k_conv_split<..., TP = true>)
    int sk_w; // > 0: stream-K launch with this many persistent workgroups (k_conv_split_sk)
    int f16;  // LAV_CONV_F16X3 applies: two fp16 pieces per operand, three products (k_conv_split_f16)
};

// Tile shape, tile geometry, tap group and split-K factor of the split kernel, by estimated time (us).
// LAV_SPLIT_FORCE="MP,MC,WPX,tw,tg,ks" pins a configuration (probes); fields that are 0 / -1 stay free.
inline SplitPlan choose_split(const lav_conv &c, const Plan &p) {
    SplitPlan best{};
    best.ok = false; best.cost = 1e30;
    int f_mp = 0, f_mc = 0, f_wpx = 0, f_tw = -1, f_tg = 0, f_ks = 0, f_wring = 0, f_tp = -1;
    if (const char *e = getenv("LAV_SPLIT_FORCE")) sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &f_mp, &f_mc, &f_wpx, &f_tw, &f_tg, &f_ks, &f_wring, &f_tp);
    // tap pairs (see the kernel): deep many-tap stems only - LAV_SPLIT_TP=0 switches the mode off
    static const bool tp_on = [] { const char *e = getenv("LAV_SPLIT_TP"); return !e || atoi(e) != 0; }();
    const bool tp_can = tp_on && !c.transposed && p.nclasses == 1 && p.taps_per_class >= 25 && p.taps_per_class <= 63 && c.cin % 16 == 0;
    static const double c_fixed = [] { const char *e = getenv("LAV_SPLIT_C_FIXED"); return e ? atof(e) : 11.0; }();
    static const double c_mma = [] { const char *e = getenv("LAV_SPLIT_C_MMA"); return e ? atof(e) : 0.105; }();
    static const double c_stage = [] { const char *e = getenv("LAV_SPLIT_C_STAGE"); return e ? atof(e) : 0.18; }();
    static const double c_chunk = [] { const char *e = getenv("LAV_SPLIT_C_CHUNK"); return e ? atof(e) : 1.2; }();
    const long ncu = c.target_cus >= 16 && c.target_cus <= 256 ? c.target_cus : 256;
    const int nchunks = p.cin_pad / 16;
    int min_taps = p.taps_per_class;   // transposed convolutions: the output-parity classes have different tap counts
    for (auto &t : p.taps) min_taps = std::min<int>(min_taps, (int)t.size());
    // An output-parity class without taps (kernel < stride, e.g. the adjoint of a 1x1 stride-2 convolution) has no weights:
    // the loaders' unconditional prologue loads would read past the packed buffer.  Such layers stay on the fp32 kernels.
    if (min_taps < 1) return best;
    const size_t LDS_MAX = 160 * 1024;
    const int shapes[6][3] = {{2, 2, 2}, {2, 2, 4}, {1, 2, 4}, {1, 2, 2}, {1, 1, 4}, {1, 1, 2}};   // MP, MC, WPX (ties: first wins)
    for (auto &sh : shapes) {
        const int MP = sh[0], MC = sh[1], WPX = sh[2], WCO = 4 / WPX, NBLK = WCO * MC, PIXW = WPX * MP * 32;
        if ((f_mp && MP != f_mp) || (f_mc && MC != f_mc) || (f_wpx && WPX != f_wpx)) continue;
        if (NBLK * 32 > p.cout_pad && NBLK > 1 && !(f_mc && f_wpx)) continue;   // more cout blocks than the layer has
        for (int tp = 0; tp <= (tp_can ? 1 : 0); ++tp)
        for (int tw : {0, 16, 32, 64, 128, 256}) {
            if (f_tp >= 0 && tp != f_tp) continue;
            // tap-pair kernels are built for two tile shapes; 16-wide tiles (a wave's 32 pixels = two tile rows) exist for them only
            if (tp && !((MP == 1 && MC == 2 && WPX == 4) || (MP == 2 && MC == 2 && WPX == 4))) continue;
            if (tw == 16 && !tp) continue;
            if (tw > PIXW || (f_tw >= 0 && tw != f_tw)) continue;
            if (tw && tw / 2 >= p.QW && tw > 32) continue;        // half of the tile would hang over the image
            if (tw == 0 && p.QW >= 64 && f_tw < 0) continue;      // wide maps: full-width rows of a linearised tile are too much staging
            const int th = tw ? PIXW / tw : 0;
            int cols, rows;
            long tiles;
            int tiles_x = 1;
            if (tw) {
                cols = tw; rows = th;
                tiles_x = (p.QW + tw - 1) / tw;
                tiles = (long)tiles_x * ((p.QH + th - 1) / th);
            } else {
                cols = p.QW; rows = (int)std::min<long>((PIXW - 1 + p.QW - 1) / p.QW + 1, p.QH);
                tiles = ((long)p.QH * p.QW + PIXW - 1) / PIXW;
            }
            const int Wreal = (cols - 1) * p.in_s + p.max_dx + 1;
            const int Wsub = (Wreal + p.in_s - 1) / p.in_s, Wst = Wsub * p.in_s;
            const int ROWS = (rows - 1) * p.in_s + p.max_dy + 1;
            const int plane = (ROWS * Wst + 63) / 64 * 64;
            if (plane > SPLIT_LOADERS * (tp ? SPLIT_NT_TP : SPLIT_NT)) continue;
            {
                // LDS: two activation chunk buffers + a ring of three groups of G taps of weights
                static const int g_max = [] { const char *e = getenv("LAV_SPLIT_GMAX"); return e ? atoi(e) : 4; }();
                for (int G : {4, 2, 1}) {
                    if (G > g_max || (f_tg && G != f_tg)) continue;
                    if (G == 4 && MP * MC > 2 && !f_tg) continue;   // 2x2 tiles have 24 matrix instructions per tap: a barrier every 2 taps is amortised
                    if (tp && G != (MP * MC == 2 ? 4 : 1)) continue;   // built: 1x2 tiles with G = 4, 2x2 tiles with G = 1 (their 256-pixel patch leaves LDS for a 3-slot ring)
                    const int steps_per_chunk = tp ? (min_taps + 1) / 2 : min_taps;
                    if (2 * G > steps_per_chunk + 1 && G > 1) continue;   // 2G consecutive steps must touch at most two chunks (in every class)
                    const size_t lds_in = (size_t)2 * (tp ? 3 : 6) * plane * 16, wslot = (size_t)NBLK * 3 * 1024;
                    const size_t lds = lds_in + 3 * G * wslot;
                    if (lds > LDS_MAX) continue;
                    const long wgs1 = tiles * ((c.cout + NBLK * 32 - 1) / (NBLK * 32)) * c.batch * p.nclasses;
                    const int ks_max = f_ks ? f_ks : (nchunks >= 4 ? std::min(16, nchunks / 2) : 1);
                    const double slab_us = (double)c.batch * c.cout * p.OH * p.OW * 4.0 * 2.0 / 4e6;
                    for (int ks = f_ks ? f_ks : 1; ks <= ks_max; ++ks) {
                        const long wgs = wgs1 * ks;
                        const int nch = (nchunks + ks - 1) / ks;
                        // per chunk: matrix work + one barrier per G taps, and the loaders' floor (a chunk's loads + conversion)
                        const double chunk_us = std::max(p.taps_per_class * (MP * MC * c_mma + c_stage / G), c_chunk * plane / 384.0);
                        // a K loop of very few steps (1x1 and stride-4 up-convolutions, parity classes of one tap) never fills the
                        // loader / compute pipeline: measured 21-35 us where the direct fp32 kernel takes 17-29 (round-4 sweep)
                        const double short_loop_us = nchunks * min_taps <= 8 ? 6.0 : 0.0;
                        const double t = (double)((wgs + ncu - 1) / ncu) * (c_fixed + short_loop_us + nch * chunk_us) + (ks > 1 ? 6.0 + ks * slab_us : 0.0);
                        if (t < best.cost * (ks > 1 ? 0.97 : 1.0) - 1e-9) {
                            best = SplitPlan{true, MP, MC, WPX, tw, th, tiles_x, (int)tiles, Wst, Wsub, ROWS, plane, G, ks, 3 * G, lds, t, tp};
                        }
                    }
                }
            }
        }
    }
    // Stream-K (k_conv_split_sk) where whole-tile rounds waste a large part of the chip: a single image, one class, the 2x2/w2 G = 2
    // kernel (the head convolution: 400 tiles on 256 CUs = two rounds, the second 56 % full), at least one tile's worth of K per
    // workgroup (a tile is then cut at most once).  OPT-IN (LAV_SPLIT_SK=1, read at every plan): measured no faster than whole tiles -
    // 282 vs 281 us on the head convolution, with and without the XCD-aware order - because the chip is POWER bound under this kernel
    // (tools/clock_probe.py: 1330 W of the 1400 W socket limit at 2.11 GHz; the half-empty second round of whole tiles simply runs at a
    // higher clock), so evening out the work buys nothing.  Kept for parts / clocks where it is not (profiles/r05_clock_power.txt).
    const char *sk_env = getenv("LAV_SPLIT_SK");
    const bool sk_on = sk_env && atoi(sk_env) != 0;
    if (sk_on && best.ok && !best.tp && best.MP == 2 && best.MC == 2 && best.WPX == 2 && best.tap_group == 2 && best.ksplit == 1 && best.tw > 0 &&
        c.batch == 1 && p.nclasses == 1 && ncu == 256 && !(c.target_cus >= 16 && c.target_cus < 256)) {
        const int NBLK = (4 / best.WPX) * best.MC;
        const long wgs = (long)best.tiles * ((c.cout + NBLK * 32 - 1) / (NBLK * 32));
        const long rounds = (wgs + ncu - 1) / ncu;
        if (wgs > ncu && rounds * ncu * 100 >= wgs * 115 && rounds <= 4) best.sk_w = (int)ncu;
    }
    return best;
}

template <int MP, int MC, int WPX, int NT, int G, bool TP = false>
int launch_split_g(const SplitArgs &sa, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split<MP, MC, WPX, NT, G, TP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // Round 4 made this workgroup take the CU's whole LDS whatever its tiles need, to keep LDS-using kernels off its CUs: the
    // finite-but-wrong results beside it were taken for an LDS effect.  Round 5 found the cause - packed fp32 instructions with an
    // op_sel bit in the VICTIMS (common.hpp) - and removed those instructions from the library, so the launch asks for what it needs
    // again (+0.5 % frames/s: the side streams' kernels overlap with it once more).  LAV_LDS_EXCLUSIVE=1 / LAV_SPLIT_LDS_EXCLUSIVE=1
    // bring the claim back (tools/coresidency.py measures both settings).
    static const bool exclusive = [] {
        const char *e = getenv("LAV_SPLIT_LDS_EXCLUSIVE"), *g = getenv("LAV_LDS_EXCLUSIVE");
        return (e && atoi(e) != 0) || (g && atoi(g) != 0);
    }();
    hipLaunchKernelGGL((k_conv_split<MP, MC, WPX, NT, G, TP>), grid, dim3(512), exclusive ? (size_t)160 * 1024 : lds, st, sa);
    return LAV_OK;
}
template <int MP, int MC, int WPX>
int launch_split_t(const SplitArgs &sa, dim3 grid, size_t lds, hipStream_t st) {
    const bool small = sa.plane <= SPLIT_LOADERS * 2;
    if (sa.tap_group == 4) return small ? launch_split_g<MP, MC, WPX, 2, 4>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 4>(sa, grid, lds, st);
    if (sa.tap_group == 2) return small ? launch_split_g<MP, MC, WPX, 2, 2>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 2>(sa, grid, lds, st);
    return small ? launch_split_g<MP, MC, WPX, 2, 1>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 1>(sa, grid, lds, st);
}

// `a`: the epilogue / output description already filled in by lav_conv2d (pointers, sizes, flags, partial slab)
inline int launch_split(const lav_conv &c, const Plan &p, const SplitPlan &sp, const ConvArgs &a, const unsigned char *w_split, hipStream_t st,
                        const unsigned char *w_f16 = nullptr) {
    SplitArgs s;
    s.x = a.x; s.w = w_split; s.bias = a.bias; s.scale = a.scale; s.shift = a.shift; s.res = a.res; s.n_valid = a.n_valid;
    s.y = a.y; s.partial = a.partial;
    s.in_c_total = a.in_c_total; s.in_c_offset = a.in_c_offset; s.cin = a.cin; s.H = a.H; s.W = a.W;
    s.cout = a.cout; s.out_c_total = a.out_c_total; s.out_c_offset = a.out_c_offset; s.OH = a.OH; s.OW = a.OW;
    s.nchunks = p.cin_pad / 16; s.nblk_total = p.cout_pad / 32;
    s.QH = p.QH; s.QW = p.QW; s.in_s = p.in_s; s.out_s = p.out_s; s.nclasses = p.nclasses; s.ksplit = sp.ksplit;
    s.tw = sp.tw; s.th = sp.th; s.tiles_x = sp.tiles_x;
    s.Wst = sp.Wst; s.Wsub = sp.Wsub; s.ROWS = sp.ROWS; s.plane = sp.plane;
    s.tap_group = sp.tap_group; s.taps_per_class = p.taps_per_class; s.wring = sp.wring;
    s.relu_pre = a.relu_pre; s.relu_post = a.relu_post; s.sigmoid = a.sigmoid; s.pad_value = a.pad_value;
    long woff = 0;
    for (int i = 0; i < MAX_CLASSES; ++i) {
        const bool live = i < p.nclasses;
        s.cls_ntaps[i] = live ? (int)p.taps[i].size() : 0;
        s.cls_in_oy[i] = live ? p.in_oy[i] : 0; s.cls_in_ox[i] = live ? p.in_ox[i] : 0;
        s.cls_out_oy[i] = live ? p.out_oy[i] : 0; s.cls_out_ox[i] = live ? p.out_ox[i] : 0;
        s.cls_woff[i] = woff;
        if (live) woff += (long)p.taps[i].size() * s.nblk_total * s.nchunks * 3 * 1024;
    }
    for (int i = 0; i < MAX_TAPS; ++i) s.toff[i] = 0;
    for (int cl = 0; cl < p.nclasses; ++cl)
        for (size_t t = 0; t < p.taps[cl].size(); ++t) {
            const int dy = p.taps[cl][t].dy, dx = p.taps[cl][t].dx;
            s.toff[cl * p.taps_per_class + t] = dy * sp.Wst + (dx % p.in_s) * sp.Wsub + dx / p.in_s;
        }
    s.trace = nullptr;
    s.f16_parts = nullptr; s.f16_wscale = nullptr; s.f16_nparts = 0;
    const int NBLK = (4 / sp.WPX) * sp.MC;
    static const bool want_trace = getenv("LAV_SPLIT_TRACE") != nullptr;
    static long long *d_trace = nullptr;
    if (want_trace) {
        if (!d_trace) LAV_HIP(hipMalloc(&d_trace, (size_t)65536 * 8 * sizeof(long long)));
        LAV_HIP(hipMemsetAsync(d_trace, 0, (size_t)65536 * 8 * sizeof(long long), st));
        s.trace = d_trace;
    }
    // x = cout tile (fastest): workgroup b lands on XCD b % 8, so an XCD's L2 streams the weights of few cout tiles
    dim3 grid((c.cout + NBLK * 32 - 1) / (NBLK * 32), sp.tiles, c.batch * p.nclasses * sp.ksplit);
    const int tok = timer_begin("conv2d", st);
    int rc = LAV_EINVAL;
    if (sp.f16) {
        // a.partial = the layer's workspace: F16_PARTS floats of per-workgroup maxima; w_f16: the fp16 section of the packed weights
        s.w = w_f16; s.f16_parts = a.partial; s.f16_nparts = F16_PARTS;
        s.f16_wscale = reinterpret_cast<const float *>(w_f16 + split_weight_bytes_f16(p));
        s.partial = nullptr;
        s.cls_woff[0] = 0;
        hipLaunchKernelGGL(k_absmax_parts, dim3(F16_PARTS), dim3(256), 0, st, a.x, c.batch, a.in_c_total, a.in_c_offset, a.cin, (long)a.H * a.W, a.partial);
        const bool small = s.plane <= SPLIT_LOADERS * 2;
        static bool attr = false;
        if (!attr) {
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_f16<2, 2, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_f16<2, 2, 2, SPLIT_NT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_f16<2, 2, 2, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_f16<2, 2, 2, SPLIT_NT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        // taps per barrier: a tap is 12 matrix instructions here (24 with bf16 pieces), so the barrier's share doubles at G = 2; with two
        // pieces a ring of 3 x 4 taps of weights fits beside the activation buffers (LAV_F16_TAP_GROUP=2 restores)
        static const int g_env = [] { const char *e = getenv("LAV_F16_TAP_GROUP"); return e ? atoi(e) : 4; }();
        const int G16 = (g_env == 4 && 2 * 4 <= p.taps_per_class + 1 && (size_t)2 * 4 * sp.plane * 16 + (size_t)3 * 4 * NBLK * 2 * 1024 <= 160 * 1024) ? 4 : 2;
        s.tap_group = G16; s.wring = 3 * G16;
        const size_t lds16 = (size_t)2 * 4 * sp.plane * 16 + (size_t)3 * G16 * NBLK * 2 * 1024;
        const dim3 g16(grid.x, grid.y, c.batch);
        if (G16 == 4) {
            if (small) hipLaunchKernelGGL((k_conv_split_f16<2, 2, 2, 2, 4>), g16, dim3(512), lds16, st, s);
            else hipLaunchKernelGGL((k_conv_split_f16<2, 2, 2, SPLIT_NT, 4>), g16, dim3(512), lds16, st, s);
        } else {
            if (small) hipLaunchKernelGGL((k_conv_split_f16<2, 2, 2, 2, 2>), g16, dim3(512), lds16, st, s);
            else hipLaunchKernelGGL((k_conv_split_f16<2, 2, 2, SPLIT_NT, 2>), g16, dim3(512), lds16, st, s);
        }
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    }
    if (sp.sk_w) {
        const size_t lds_sk = sp.lds;
        const int nbx = (int)grid.x, ntiles = (int)(grid.x * grid.y);
        const bool small = s.plane <= SPLIT_LOADERS * 2;
        static bool attr = false;
        if (!attr) {
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_sk<2, 2, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_split_sk<2, 2, 2, SPLIT_NT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        if (small) hipLaunchKernelGGL((k_conv_split_sk<2, 2, 2, 2, 2>), dim3(sp.sk_w), dim3(512), lds_sk, st, s, nbx, ntiles);
        else hipLaunchKernelGGL((k_conv_split_sk<2, 2, 2, SPLIT_NT, 2>), dim3(sp.sk_w), dim3(512), lds_sk, st, s, nbx, ntiles);
        SkFixupArgs f;
        f.partial = a.partial; f.bias = a.bias; f.scale = a.scale; f.shift = a.shift; f.res = a.res; f.n_valid = a.n_valid; f.y = a.y;
        f.cout = a.cout; f.out_c_total = a.out_c_total; f.out_c_offset = a.out_c_offset; f.OH = a.OH; f.OW = a.OW;
        f.tw = sp.tw; f.th = sp.th; f.tiles_x = sp.tiles_x; f.couts_per_tile = NBLK * 32; f.nbx = nbx; f.nck = s.nchunks; f.ntiles = ntiles; f.W = sp.sk_w;
        f.relu_pre = a.relu_pre; f.relu_post = a.relu_post; f.sigmoid = a.sigmoid;
        const long total = (long)a.cout * p.OH * p.OW;
        hipLaunchKernelGGL(k_conv_sk_fixup, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, f);
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        return LAV_OK;
    }
    if (sp.tp) {
        if (sp.MP == 1 && sp.MC == 2 && sp.WPX == 4 && sp.tap_group == 4) rc = launch_split_g<1, 2, 4, SPLIT_NT_TP, 4, true>(s, grid, sp.lds, st);
        else if (sp.MP == 2 && sp.MC == 2 && sp.WPX == 4 && sp.tap_group == 1) rc = launch_split_g<2, 2, 4, SPLIT_NT_TP, 1, true>(s, grid, sp.lds, st);
        else return fail(LAV_EINVAL, "lav_conv2d: tap-pair split tile %dx%d/%d G%d not built", sp.MP, sp.MC, sp.WPX, sp.tap_group);
    } else
    switch (sp.MP * 100 + sp.MC * 10 + sp.WPX) {
        case 224: rc = launch_split_t<2, 2, 4>(s, grid, sp.lds, st); break;
        case 124: rc = launch_split_t<1, 2, 4>(s, grid, sp.lds, st); break;
        case 114: rc = launch_split_t<1, 1, 4>(s, grid, sp.lds, st); break;
        case 222: rc = launch_split_t<2, 2, 2>(s, grid, sp.lds, st); break;
        case 122: rc = launch_split_t<1, 2, 2>(s, grid, sp.lds, st); break;
        case 112: rc = launch_split_t<1, 1, 2>(s, grid, sp.lds, st); break;
        default: return fail(LAV_EINVAL, "lav_conv2d: split tile %dx%d/%d not built", sp.MP, sp.MC, sp.WPX);
    }
    if (rc) return rc;
    if (sp.ksplit > 1) {
        const long total = (long)c.batch * a.cout * p.OH * p.OW;
        hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, c.batch);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    static int runs = 0;
    if (s.trace && ++runs % 8 == 0) {   // debug: where the workgroups' time goes (cycles of the shader clock)
        const size_t nwg = (size_t)grid.x * grid.y * grid.z;
        if (nwg <= 65536) {
            std::vector<long long> h(nwg * 8);
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                double pro = 0, loop = 0, cw = 0, lconv = 0, lw = 0, dw = 0;
                long long t0 = h[0], t1 = 0;
                for (size_t i = 0; i < nwg; ++i) {
                    pro += (double)(h[i * 8 + 1] - h[i * 8]); loop += (double)(h[i * 8 + 2] - h[i * 8 + 1]);
                    cw += (double)h[i * 8 + 4]; lconv += (double)h[i * 8 + 5]; lw += (double)h[i * 8 + 6]; dw += (double)h[i * 8 + 7];
                    t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 2]);
                }
                const int nst = (s.nchunks / sp.ksplit) * p.taps_per_class;
                fprintf(stderr, "[split trace] %zu wgs %dx%d/w%d tw%d ring %d ks%d, %d steps: span %.0f kcyc | per wg: prologue wait %.0f, loop %.0f cyc (%.0f per step) | barrier wait: compute %.0f, loaders %.0f (convert %.0f) %.0f\n",
                        nwg, sp.MP, sp.MC, sp.WPX, sp.tw, sp.wring, sp.ksplit, nst, (double)(t1 - t0) / 1e3, pro / nwg, loop / nwg, loop / nwg / std::max(nst, 1),
                        cw / nwg, lw / nwg, lconv / nwg, dw / nwg);
            }
        }
    }
    return LAV_OK;
}

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
#include "conv_split_kernel.hpp"

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

// LAV_CONV_F16X3: the same body on two fp16 pieces per operand and three products lives in conv_f16.hip (k_conv_split_f16, k_absmax_parts).
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

// LAV_CONV_F16X3 packing: [class][cout block][tap][chunk][piece 2][lane = khalf*32 + cout%32][8 channels] fp16 of w / s_w (the bf16
// packing's order with two pieces), then the scale s_w (a power of two that puts the largest |w| into [16384, 32768)) as one float at
// a 16-byte aligned offset
inline size_t split_weight_bytes_f16(const Plan &p) {
    size_t taps = 0;
    for (auto &t : p.taps) taps += t.size();
    return taps * (size_t)(p.cout_pad / 32) * (p.cin_pad / 16) * 2 * 1024;
}
// The layers the mode takes (round 6: whatever the split kernel takes - any stride, tile, tap group, split-K, tap pairs, the parity
// classes of a transposed convolution).  LAV_F16X3_LAYERS=head restores round 5's rule (stride-1 single-class layers of >= 64 -> >= 128 channels).
inline bool f16x3_layer(const lav_conv &c, const Plan &p) {
    static const bool head_only = [] { const char *e = getenv("LAV_F16X3_LAYERS"); return e && !strcmp(e, "head"); }();
    if (head_only) return !c.transposed && p.nclasses == 1 && c.stride == 1 && c.cin % 16 == 0 && c.cin >= 64 && c.cout >= 128;
    return c.cin >= 16;
}
inline void split_pack_weights_f16(const lav_conv &c, const Plan &p, const float *h_weight, unsigned char *out) {
    const int nblk = p.cout_pad / 32, nchunks = p.cin_pad / 16;
    const size_t nw = (size_t)c.cout * c.cin * c.kh * c.kw;
    float m = 0.f;
    for (size_t i = 0; i < nw; ++i) { const float v = fabsf(h_weight[i]); if (v <= 3.4028235e38f && v > m) m = v; }
    int e = 0;
    (void)frexpf(m, &e);
    const float sw = ldexpf(1.f, m > 0.f ? std::max(e, -100) - 15 : 0), inv = 1.f / sw;   // (floor: as the activations' scale in split_body)
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    size_t cls_off = 0;   // in halves
    for (int cls = 0; cls < p.nclasses; ++cls) {
        const auto &t = p.taps[cls];
        parallel_for(nblk, [&, cls_off](int blk) {
            for (size_t ti = 0; ti < t.size(); ++ti)
                for (int ch = 0; ch < nchunks; ++ch)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int el = 0; el < 8; ++el) {
                            const int co = blk * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + el;
                            float w = 0.f;
                            if (co < c.cout && ci < c.cin)
                                w = (c.transposed ? h_weight[(((size_t)ci * c.cout + co) * c.kh + t[ti].ky) * c.kw + t[ti].kx]
                                                  : h_weight[(((size_t)co * c.cin + ci) * c.kh + t[ti].ky) * c.kw + t[ti].kx]) * inv;
                            const _Float16 h0 = (_Float16)w, h1 = (_Float16)(w - (float)h0);
                            const size_t frag = cls_off + ((((size_t)blk * t.size() + ti) * nchunks + ch) * 2) * 512;
                            o[frag + lane * 8 + el] = h0;
                            o[frag + 512 + lane * 8 + el] = h1;
                        }
        });
        cls_off += t.size() * (size_t)nblk * nchunks * 2 * 512;
    }
    const float tail[4] = {sw, 0.f, 0.f, 0.f};   // (the scale and its 12 bytes of padding: the whole buffer is defined - device re-packs compare equal)
    memcpy(out + split_weight_bytes_f16(p), tail, sizeof(tail));
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
    const size_t LDS_MAX = 160 * 1024 - 16;   // (16 bytes behind the tile buffers: the workgroup's output maximum + arrival count, SplitArgs::sync_off)
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
                        static const double sks_cost = [] { const char *e = getenv("LAV_SPLIT_KS_COST"); return e ? atof(e) : 6.0; }();   // price of the reduce launch
                        const double t = (double)((wgs + ncu - 1) / ncu) * (c_fixed + short_loop_us + nch * chunk_us) + (ks > 1 ? sks_cost + ks * slab_us : 0.0);
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
    hipLaunchKernelGGL((k_conv_split<MP, MC, WPX, NT, G, TP>), grid, dim3(512), exclusive ? (size_t)160 * 1024 : (lds + 15) / 16 * 16 + 16, st, sa);   // (+ the two sync words, SplitArgs::sync_off)
    return LAV_OK;
}
template <int MP, int MC, int WPX>
int launch_split_t(const SplitArgs &sa, dim3 grid, size_t lds, hipStream_t st) {
    const bool small = sa.plane <= SPLIT_LOADERS * 2;
    if (sa.tap_group == 4) return small ? launch_split_g<MP, MC, WPX, 2, 4>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 4>(sa, grid, lds, st);
    if (sa.tap_group == 2) return small ? launch_split_g<MP, MC, WPX, 2, 2>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 2>(sa, grid, lds, st);
    return small ? launch_split_g<MP, MC, WPX, 2, 1>(sa, grid, lds, st) : launch_split_g<MP, MC, WPX, SPLIT_NT, 1>(sa, grid, lds, st);
}

// Scale hand-off of LAV_CONV_F16X3 (conv_f16.hip): what lav_conv2d_amax was given.
struct AmaxIO {
    const float *in;    // maxima of the finite |x| left by the launches that wrote x (null: measured by a launch into `scratch`)
    int n_in;
    float *out;         // where this layer leaves the maxima of |y| (null: not wanted); lav_conv_amax_count() floats
    float *scratch;     // F16_PARTS floats of the layer's workspace
};

// `a`: the epilogue / output description already filled in by lav_conv2d (pointers, sizes, flags, partial slab)
inline int launch_split(const lav_conv &c, const Plan &p, const SplitPlan &sp, const ConvArgs &a, const unsigned char *w_split, hipStream_t st,
                        const unsigned char *w_f16 = nullptr, const AmaxIO &io = AmaxIO{nullptr, 0, nullptr, nullptr}) {
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
    // the workgroups' output maxima come from the kernel's epilogue (whole-K launches) or from k_conv_reduce (split-K); the stream-K
    // launch has neither (lav_conv2d measures its output with a launch)
    s.amax_out = sp.ksplit == 1 && !sp.sk_w ? io.out : nullptr;
    s.sync_off = (int)((sp.lds + 15) / 16 * 16);
    auto print_trace = [&]() {
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
    };
    if (sp.f16) {
        // w_f16: the fp16 section of the packed weights (class offsets in two-piece units); the scale of x from the producers' maxima,
        // else measured here
        s.w = w_f16;
        s.f16_wscale = reinterpret_cast<const float *>(w_f16 + split_weight_bytes_f16(p));
        long woff16 = 0;
        int min_taps = p.taps_per_class;
        for (int i = 0; i < p.nclasses; ++i) {
            s.cls_woff[i] = woff16;
            woff16 += (long)p.taps[i].size() * s.nblk_total * s.nchunks * 2 * 1024;
            min_taps = std::min<int>(min_taps, (int)p.taps[i].size());
        }
        if (io.in) {
            s.f16_parts = io.in; s.f16_nparts = io.n_in;
        } else {
            LAV_REQUIRE(io.scratch, "lav_conv2d: LAV_CONV_F16X3 without producer maxima needs its workspace");
            int rc16 = launch_absmax_parts(a.x, c.batch, a.in_c_total, a.in_c_offset, a.cin, (long)a.H * a.W, io.scratch, a.n_valid, st);
            if (rc16) return rc16;
            s.f16_parts = io.scratch; s.f16_nparts = F16_PARTS;
        }
        // taps per barrier: a tap is half the matrix instructions of the bf16 kernel, so the barrier's share doubles - four taps per
        // barrier wherever the two-piece ring of 3 x 4 taps fits beside the activation buffers and 2G steps stay within two chunks
        // (LAV_F16_TAP_GROUP=2 caps it); the tap-pair kernels exist for their bf16 tap groups only
        static const int g_env = [] { const char *e = getenv("LAV_F16_TAP_GROUP"); return e ? atoi(e) : 4; }();
        const size_t lds_in16 = (size_t)2 * (sp.tp ? 2 : 4) * sp.plane * 16;
        int G16 = sp.tap_group;
        if (!sp.tp) {
            G16 = 1;
            for (int g : {2, 4})
                if (g <= g_env && 2 * g <= min_taps + 1 && lds_in16 + (size_t)3 * g * NBLK * 2 * 1024 + 16 <= 160 * 1024) G16 = g;
            if (G16 < sp.tap_group) G16 = sp.tap_group;   // (never fewer taps per barrier than the bf16 plan, which holds 3/2 of the bytes)
        }
        s.tap_group = G16; s.wring = 3 * G16;
        const size_t lds16 = lds_in16 + (size_t)3 * G16 * NBLK * 2 * 1024;
        s.sync_off = (int)((lds16 + 15) / 16 * 16);
        rc = launch_split_f16(&s, sizeof(s), sp.MP, sp.MC, sp.WPX, sp.tp, grid.x, grid.y, grid.z, (size_t)s.sync_off + 16, st);
        if (rc) return rc;
        if (sp.ksplit > 1) {
            const long total = (long)c.batch * a.cout * p.OH * p.OW;
            hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, c.batch, io.out);
        }
        timer_end(tok, st);
        LAV_LAUNCH_CHECK();
        print_trace();
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
        hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, c.batch, io.out);
    }
    timer_end(tok, st);
    LAV_LAUNCH_CHECK();
    print_trace();
    return LAV_OK;
}

// Convolutions with a handful of input channels on the vector ALU (round 5) - included by conv.hip (inside its namespace).
//
// The camera stems (brake net: 3 -> 64, 7x7 stride 2 on 288 x 768 and 192 x 480; ERFNet's first block: 3 -> 13, 3x3 stride 2 on three
// 288 x 256 images) have K = cin * taps = 147 resp. 27.  The matrix kernels stage 16-channel chunks: 13 of 16 k-slots of every matrix
// instruction multiply zeros (measured 104.7 us = 9.9 TFLOP/s for the wide stem, 36.7 us = 1.1 TFLOP/s for ERFNet's), and their
// prologues are sized for deep layers.  Here the contraction runs on packed fp32 FMAs:
//   workgroup   128 x CS threads, a tile of 8 x 32 output pixels; a thread owns two pixels (rows ty and ty + 4) x COUT / CS output channels
//               (wave pair g the g-th slice of the channels: with all 64 channels per thread the wide stem was 432 waves of 75 k VALU
//               cycles on 1024 SIMDs - 78 us; two slices 53 us; four slices the figure in DESIGN 4.3);
//   LDS         the layer's weights as [tap][ci][cout] rows (read as 16-byte broadcasts: every lane of a wave wants the same four
//               output channels' weights) gathered once per workgroup from lav_conv_pack_weights' fp32 layout, and the input patch of
//               the tile ((8 - 1) s + k rows x (32 - 1) s + k columns x cin, out-of-image positions hold pad_value);
//   arithmetic  acc[co, co+1] += x * w[co, co+1] as one v_pk_fma_f32 in the form that is safe on this part (broadcast operand in the low
//               half of the FIRST source, no op_sel bit: common.hpp), channels and taps in ascending order: an fp32 FMA chain per output.
// Same epilogue as the other kernels (bias, ReLU, folded BatchNorm, residual, ReLU, sigmoid), channel windows on both sides, batch limit.
typedef float sc_v2f __attribute__((ext_vector_type(2)));

struct SmallCinArgs {
    const float *x, *w, *bias, *scale, *shift, *res;
    const int *n_valid;
    float *y;
    int in_c_total, in_c_offset, H, W;
    int cout, out_c_total, out_c_offset, OH, OW;
    int cin_pad, ntaps, pad_h, pad_w;
    int tiles_x, tiles_y;
    int relu_pre, relu_post, sigmoid;
    float pad_value;
};

constexpr int SC_TW = 32;   // output columns of a tile; rows: PPT x 4 (a thread owns PPT pixels: rows ty and, PPT = 2, ty + 4)

template <int CIN, int K, int S, int COUT, int CS, int PPT>
__global__ __launch_bounds__(128 * CS) void k_conv_smallcin(SmallCinArgs a) {
    constexpr int SC_TH = 4 * PPT;
    constexpr int PW = (SC_TW - 1) * S + K, PH = (SC_TH - 1) * S + K;      // patch
    constexpr int PWS = PW | 1;                                            // odd row stride: the stride-2 column reads of a wave spread over the banks
    constexpr int NT = K * K;
    extern __shared__ __attribute__((aligned(16))) float sc_smem[];
    float *s_w = sc_smem;                              // [NT * CIN][COUT]
    float *s_in = sc_smem + NT * CIN * COUT;           // [CIN][PH][PWS]
    float *s_ep = s_in + CIN * PH * PWS;               // [3][COUT]: bias, scale, shift (read per channel behind the stores to y, a scalar
    //                                                    load of them was a memory round trip per channel: 130 us of epilogue)
    constexpr int CH = COUT / CS, NTH = 128 * CS;                          // output channels per thread; threads
    const int tid = threadIdx.x, tx = tid & 31, ty = (tid >> 5) & 3;      // ty 0..3
    const int c_lo = __builtin_amdgcn_readfirstlane(tid >> 7) * CH;       // this wave's channel half (wave-uniform)
    const int n = blockIdx.z;
    if (a.n_valid && n >= *a.n_valid) return;
    const int ox0 = blockIdx.x * SC_TW, oy0 = blockIdx.y * SC_TH;
    // weights: s_w[(t * CIN + ci) * COUT + co] from the packed fp32 layout [cout block][tap][8-channel group][lane = parity * 32 + cout % 32][pair].
    // Eight independent loads per thread are issued before the first is stored (from clamped indices, selected afterwards): taken one
    // at a time the 74 gathers of a thread were 74 memory round trips - the first version of this kernel spent > 100 us in its prologue.
    constexpr int NWT = NT * CIN * COUT, UW = 8;
    for (int e0 = tid; e0 < NWT; e0 += NTH * UW) {
        float v[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int e = min(e0 + u * NTH, NWT - 1);
            const int co = e % COUT, r = e / COUT, ci = r % CIN, t = r / CIN;
            const int cc = min(co, a.cout - 1);
            const float w = a.w[(size_t)(cc / 32) * a.ntaps * a.cin_pad * 32 + (size_t)((t * a.cin_pad + ci) / 8) * 256 + ((ci & 1) * 32 + cc % 32) * 4 + (ci % 8) / 2];
            v[u] = co < a.cout ? w : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UW; ++u)
            if (e0 + u * NTH < NWT) s_w[e0 + u * NTH] = v[u];
    }
    // input patch (rows of PW consecutive pixels: coalesced), the same way
    const int iy0 = oy0 * S - a.pad_h, ix0 = ox0 * S - a.pad_w;
    const float *xn = a.x + ((long)n * a.in_c_total + a.in_c_offset) * a.H * a.W;
    constexpr int NPT = CIN * PH * PW;
    for (int e0 = tid; e0 < NPT; e0 += NTH * UW) {
        float v[UW];
        bool ok[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int e = min(e0 + u * NTH, NPT - 1);
            const int px = e % PW, r = e / PW, py = r % PH, ci = r / PH;
            const int iy = iy0 + py, ix = ix0 + px;
            ok[u] = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            v[u] = xn[(long)ci * a.H * a.W + (long)min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)];
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int e = e0 + u * NTH;
            if (e < NPT) {
                const int px = e % PW, r = e / PW;   // (r = ci * PH + py)
                s_in[r * PWS + px] = ok[u] ? v[u] : a.pad_value;
            }
        }
    }
    if (tid < COUT) {
        const int cc = min(tid, a.cout - 1);
        s_ep[tid] = a.bias ? a.bias[cc] : 0.f;
        s_ep[COUT + tid] = a.scale ? a.scale[cc] : 1.f;
        s_ep[2 * COUT + tid] = a.scale ? a.shift[cc] : 0.f;
    }
    __syncthreads();
    sc_v2f acc[PPT][CH / 2];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int q = 0; q < CH / 2; ++q) acc[p][q] = sc_v2f{0.f, 0.f};
    const float *in0 = s_in + (ty * S) * PWS + tx * S;            // pixel (ty, tx); the second pixel sits 4 S rows further down
    // Weight rows travel LDS -> registers in chunks of CQ 16-byte reads, the next chunk requested before the current one is multiplied
    // (two register sets): read right before their use - what the compiler makes of the plain loop - every four FMAs waited a full
    // LDS round trip (107 us for the wide stem).
    constexpr int QN = CH / 4, CQ = QN < 8 ? QN : 8, CPT = QN / CQ, NCH = K * CPT;   // 16-byte reads per tap, per chunk; chunks per tap, per kernel row
    float4 wbuf[2][CQ];
    for (int ci = 0; ci < CIN; ++ci) {
        for (int ky = 0; ky < K; ++ky) {
            const float *r0 = in0 + (ci * PH + ky) * PWS, *r1 = r0 + 4 * S * PWS;
            const float *wrow = s_w + ((ky * K) * CIN + ci) * COUT + c_lo;
            float xa[K], xb[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) { xa[kx] = r0[kx]; xb[kx] = PPT == 2 ? r1[kx] : 0.f; }
#pragma unroll
            for (int q = 0; q < CQ; ++q) wbuf[0][q] = reinterpret_cast<const float4 *>(wrow)[q];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int kx = c / CPT, h = c % CPT;
                if (c + 1 < NCH) {
                    const float4 *nx = reinterpret_cast<const float4 *>(wrow + ((c + 1) / CPT) * CIN * COUT) + ((c + 1) % CPT) * CQ;   // the same address in every lane: broadcast reads
#pragma unroll
                    for (int q = 0; q < CQ; ++q) wbuf[(c + 1) & 1][q] = nx[q];
                }
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks each read to four FMAs before its use: one round trip per read)
                const sc_v2f x0 = sc_v2f{xa[kx], 0.f}, x1 = sc_v2f{xb[kx], 0.f};
#pragma unroll
                for (int q = 0; q < CQ; ++q) {
                    const float4 t = wbuf[c & 1][q];
                    const sc_v2f wa = sc_v2f{t.x, t.y}, wb = sc_v2f{t.z, t.w};
                    const int o = 2 * (h * CQ + q);
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0][o]) : "v"(x0), "v"(wa));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0][o + 1]) : "v"(x0), "v"(wb));
                    if constexpr (PPT == 2) {
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[1][o]) : "v"(x1), "v"(wa));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[1][o + 1]) : "v"(x1), "v"(wb));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // epilogue (as k_conv_direct): a wave's 32 lanes of one row write 128 contiguous bytes per channel
    const long plane_o = (long)a.OH * a.OW;
    const int ox = ox0 + tx;
    bool pok[PPT];
    long pix[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int oy = oy0 + ty + 4 * p;
        pok[p] = oy < a.OH && ox < a.OW;
        pix[p] = (long)min(oy, a.OH - 1) * a.OW + min(ox, a.OW - 1);
    }
#pragma unroll
    for (int c4 = 0; c4 < CH / 4; ++c4) {
        const float4 b4 = reinterpret_cast<const float4 *>(s_ep + c_lo)[c4], sc4 = reinterpret_cast<const float4 *>(s_ep + COUT + c_lo)[c4],
                     sh4 = reinterpret_cast<const float4 *>(s_ep + 2 * COUT + c_lo)[c4];
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, hh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = 4 * c4 + j, co = c_lo + cl;
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                if (co >= a.cout || !pok[p]) continue;
                float v = acc[p][cl >> 1][cl & 1] + bb[j];
                if (a.relu_pre) v = v > 0.f ? v : 0.f;
                v = fmaf(v, ss[j], hh[j]);
                const long idx = ((long)n * a.out_c_total + a.out_c_offset + co) * plane_o + pix[p];
                if (a.res) v += a.res[idx];
                if (a.relu_post) v = v > 0.f ? v : 0.f;
                if (a.sigmoid && co >= a.sigmoid - 1) v = 1.f / (1.f + expf(-v));
                a.y[idx] = v;
            }
        }
    }
}

// Which layers the kernel is built for (LAV_CONV_SMALLCIN=0 switches it off)
inline bool smallcin_applies(const lav_conv &c) {
    const char *e = getenv("LAV_CONV_SMALLCIN");
    if (e && atoi(e) == 0) return false;
    if (c.transposed || c.dil_h != 1 || c.dil_w != 1 || c.kh != c.kw || c.stride != 2) return false;
    // (ERFNet's second downsampler, 16 -> 48 with K = 144, was tried on an instance <16, 3, 2, 48, 4, 1> of this kernel: 33.3 us against 25.4 us
    // on the split kernel - the 37 KB patch per 4 x 32 tile costs more than the matrix kernel's prologue; not kept.)
    if (c.cin != 3) return false;
    return (c.kh == 7 && c.cout <= 64 && c.cout > 16) || (c.kh == 3 && c.cout <= 16);
}

inline int smallcin_tile_rows(const lav_conv &) { return 8; }   // 4 x pixels per thread (PPT) of the instance that takes the layer

template <int CIN, int K, int S, int COUT, int CS, int PPT>
int launch_smallcin_t(const SmallCinArgs &a, dim3 grid, hipStream_t st) {
    constexpr int SC_TH = 4 * PPT;
    constexpr int PW = (SC_TW - 1) * S + K, PH = (SC_TH - 1) * S + K, PWS = PW | 1;
    constexpr size_t lds = (size_t)(K * K * CIN * COUT + CIN * PH * PWS + 3 * COUT) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        LAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_smallcin<CIN, K, S, COUT, CS, PPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((k_conv_smallcin<CIN, K, S, COUT, CS, PPT>), grid, dim3(128 * CS), lds, st, a);
    return LAV_OK;
}

inline int launch_smallcin(const lav_conv &c, const Plan &p, const ConvArgs &ca, hipStream_t st) {
    SmallCinArgs a;
    a.x = ca.x; a.w = ca.w; a.bias = ca.bias; a.scale = ca.scale; a.shift = ca.shift; a.res = ca.res; a.n_valid = ca.n_valid; a.y = ca.y;
    a.in_c_total = c.in_c_total; a.in_c_offset = c.in_c_offset; a.H = c.h; a.W = c.w;
    a.cout = c.cout; a.out_c_total = c.out_c_total; a.out_c_offset = c.out_c_offset; a.OH = p.OH; a.OW = p.OW;
    a.cin_pad = p.cin_pad; a.ntaps = c.kh * c.kw; a.pad_h = c.pad_h; a.pad_w = c.pad_w;
    const int th = smallcin_tile_rows(c);
    a.tiles_x = (p.OW + SC_TW - 1) / SC_TW; a.tiles_y = (p.OH + th - 1) / th;
    a.relu_pre = c.relu_pre; a.relu_post = c.relu_post; a.sigmoid = c.sigmoid; a.pad_value = c.pad_value;
    const dim3 grid(a.tiles_x, a.tiles_y, c.batch);
    const int tok = timer_begin("conv2d", st);
    int rc;
    if (c.kh == 7) rc = launch_smallcin_t<3, 7, 2, 64, 4, 2>(a, grid, st);
    else rc = launch_smallcin_t<3, 3, 2, 16, 2, 2>(a, grid, st);
    timer_end(tok, st);
    if (rc) return rc;
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

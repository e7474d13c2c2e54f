// Minimal victims for the co-residency effect of DESIGN 4.4c (tools/lds_hazard.py runs them beside tools/probes/lds_hog.hip).
// Each pattern is one hand-written instruction sequence that hipcc is ALLOWED to emit for ordinary C++ (its hazard recognizer pads
// none of them), repeated `iters` times by every lane on its own LDS words with fresh values; a lane counts the results that are not
// what the ISA manual promises.  The workgroup is small (256 threads, 8 KB of LDS, < 64 VGPRs) so that it fits a CU beside a
// stem-shaped neighbour (512 threads, 150 KB of LDS, <= 184 VGPRs).
//   0  control: write, wait, read, wait
//   1  store-DATA write-after-read: ds_write_b32 a, v ; v_mov v, junk          (does the store see the old v?)
//   2  store-ADDRESS write-after-read: ds_write_b32 a, v ; v_mov a, other
//   3  load-ADDRESS write-after-read: ds_read_b32 r, a ; v_mov a, other
//   4  two loads in flight, every register distinct and untouched until lgkmcnt(0)
//   5  eight loads in flight, every register distinct and untouched until lgkmcnt(0)
//   6  in-order return: two loads, s_waitcnt lgkmcnt(1), consume the first
//   7  store then load of the same word, back to back (LDS executes a wave's operations in order)
//   8  store to word A and load of word B in flight together, everything pinned
//   9  ds_write_b64 a, v[0:1] ; v_mov_b64 v[0:1], junk
//  10  ds_write2_b64 a, v[0:1], v[2:3] ; v_mov_b64 v[2:3], junk      (the pair hipcc produced in round 4: 128 bits of data)
//  11  the same with one independent instruction (s_nop 0) in between
//  12  ds_bpermute_b32 and a load in flight together (what __shfl_xor beside an LDS read looks like), everything pinned
//  13  compiled C++ (no inline asm): six interleaved xor-butterfly sums per wave on __shfl_xor (= ds_bpermute_b32), as in rounds 2-4's plan
//      kernel, compared with the same sums on v_permlane32_swap / v_permlane16_swap / DPP (integer adds: any order gives the same bits)
//  14  13 + the plan kernel's hand-off: lane 0 of every wave stores its six sums and a tag to LDS, LDS-only barrier, 24 threads read them
//      back and compare with a copy stored beside them (counts stale tags and mismatching copies), barrier
//  15  NO LDS at all: the same chain of vector-ALU work (floor, min / max, compares + selects, a division, sinf / cosf, fused
//      multiply-adds - what a kernel's per-thread set-up looks like) evaluated twice on identical inputs; the two results must have the
//      same bits.  errors[16 + q] counts the mismatches of lanes 16 q .. 16 q + 15 (a wave64 vector instruction runs as four passes
//      of 16 lanes).
//  16-21  15 split by instruction class: 16 fused multiply-adds only | 17 compares + selects (v_cmp -> mask -> v_cndmask) | 18 the
//      transcendental unit (v_exp_f32, v_rcp_f32, v_sin_f32, v_sqrt_f32) | 19 integer add / multiply / shift / min / max | 20 floor,
//      float <-> int conversions | 21 IEEE division (v_div_scale / v_div_fmas / v_div_fixup: the mask travels through VCC)
//  22-26  22 v_cmp -> vcc -> v_cndmask back to back (inline asm) | 23 divergent branches (s_and_saveexec / s_or exec around asm) |
//      24 two v_cmp -> s_and_b64 -> v_cndmask (a lane mask through the scalar ALU) | 25 precise sinf + cosf | 26 precise expf
//  27-29  PACKED fp32 (two floats per lane and instruction, gfx90a+): 27 v_pk_fma_f32 | 28 v_pk_mul_f32 + v_pk_add_f32 |
//      29 the same arithmetic as 27 written with scalar v_fma_f32 (inline asm, so that the compiler cannot pack it)
//  40  registers at REST: 24 VGPRs are written once (lane-dependent values), the wave then sleeps (s_sleep loop) and re-reads them: a
//      mismatch means that a value changed while its wave did not touch it.  41: the same while the wave runs a multiply-add chain on
//      other registers.  errors[48 + q] counts lanes 16 q .. 16 q + 15, errors[56 + (register index & 7)] which registers.
//  42-44  vector instructions with a SCALAR-register source: 42 packed fp32 with an SGPR PAIR (v_pk_mul_f32 v, v, s[2:3] /
//      v_pk_add_f32 v, v, s[4:5]: what hipcc makes of pairs of lane values times uniform constants) | 43 v_mul_f32 / v_add_f32 with one
//      32-bit SGPR | 44 v_pk_fma_f32 v, v, s[2:3], v
//  45  15 evaluated twice with every intermediate kept: errors[64 + i] counts how often intermediate i is the FIRST one to differ
//      (0 cs, 1 sn, 2 k, 3 gx, 4 gy, 5 ix, 6 iy, 7 fx, 8 fy, 9 x0, 10 y0, 11 wx1, 12 wy1, 13 w00, 14 w01, 15 w10, 16 w11, 17 cx0, 18 cy1, 19 exp term)
//  46-49  SCALAR-register write-after-read: a vector instruction reads an SGPR, the scalar ALU overwrites that SGPR right behind it (what
//      hipcc emits when it recycles the register of a uniform constant: `v_pk_add_f32 v[4:5], v[10:11], s[0:1]` ... `s_mov_b32 s0, 0x431f0000`
//      in pattern 15): 46 v_pk_add_f32 v, v, s[2:3] ; s_mov_b32 s2 / s3 | 47 the same with two independent vector instructions in
//      between (pattern 15's distance) | 48 v_add_f32 v, s2, v ; s_mov_b32 s2 (32-bit source) | 49 46 with s_nop 4 in between
//  50-53  pattern 15's own instructions around the first value that goes wrong (gx, tools/lds_hazard.py pattern 45), as inline asm:
//      v_cndmask x2 -> v_pk_mul_f32 op_sel_hi:[0,1] -> v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] -> v_pk_fma_f32 neg_lo / neg_hi ->
//      v_pk_fma_f32 op_sel_hi:[0,1,1] -> v_pk_add_f32 v, v, s[2:3] (s2 / s3 set two instructions earlier).  50 as compiled |
//      51 s_nop 3 between all of them | 52 without the v_cndmask producers | 53 without the SGPR add
//  54-59  one packed fp32 instruction each, 48 in a dependent chain: 54 v_pk_mul_f32 op_sel_hi:[0,1] | 55 v_pk_mul_f32 op_sel:[0,1]
//      op_sel_hi:[0,0] | 56 v_pk_fma_f32 neg_lo:[0,0,1] neg_hi:[0,0,1] | 57 v_pk_fma_f32 op_sel_hi:[0,1,1] | 58 v_pk_add_f32 op_sel_hi:[1,0]
//      (second source: low half for both) | 59 v_pk_fma_f32 without modifiers (= 27, the control)
//  60-63  the other packed fp32 forms with an op_sel bit: 60 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] | 61 v_pk_fma_f32 op_sel:[0,1,0]
//      op_sel_hi:[1,0,1] | 62 v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (first source swapped) | 63 v_pk_mul_f32 op_sel:[0,1]
//      op_sel_hi:[1,1] (both results from the second source's HIGH register)
//  30-31  a lane mask in a GENERAL SGPR pair: 30 v_cmp_gt_f32 s[2:3] -> v_cndmask_b32 ..., s[2:3] back to back | 31 two masks alive at once
#include <hip/hip_runtime.h>

constexpr int SLOTS = 8;

__device__ __forceinline__ unsigned ifold32(unsigned x, unsigned y) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); return r[0] + r[1]; }
__device__ __forceinline__ unsigned ifold16(unsigned x, unsigned y) { const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); return r[0] + r[1]; }
template <int CTRL> __device__ __forceinline__ unsigned idpp_add(unsigned x) { return x + (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned wave_isum_valu(unsigned v) {
    v = ifold16(ifold32(v, v), ifold32(v, v));
    v = idpp_add<0x128>(v); v = idpp_add<0x124>(v); v = idpp_add<0x122>(v); v = idpp_add<0x121>(v);
    return v;
}

__global__ __launch_bounds__(256) void k_hazard(int pattern, int iters, unsigned *__restrict__ errors, unsigned *__restrict__ checks) {
    __shared__ unsigned lds[SLOTS][256];
    const unsigned tid = threadIdx.x;
    unsigned a[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) a[s] = (unsigned)(size_t)&lds[s][tid];
    unsigned err = 0, n = 0;
    const unsigned salt = blockIdx.x * 2654435761u + tid * 40503u;
    auto wr = [](unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory"); };
    auto rd = [](unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; };
    if ((pattern >= 16 && pattern <= 31) || (pattern >= 42 && pattern <= 44) || (pattern >= 46 && pattern <= 63)) {
        auto kind = [](int pat, float x, float y) __attribute__((noinline)) {
            float r = x;
            if (pat == 16) {
#pragma unroll
                for (int i = 0; i < 48; ++i) r = fmaf(r, 0.75f + 0.001f * i, y * (0.01f * i) - 0.3f);
            } else if (pat == 17) {
#pragma unroll
                for (int i = 0; i < 48; ++i) { const float c = 0.02f * i; r = (r > c) ? r - y * c : ((r < -c) ? r + y : r + 0.37f); }
            } else if (pat == 18) {
#pragma unroll
                for (int i = 0; i < 12; ++i) r = __builtin_amdgcn_rcpf(1.5f + __builtin_amdgcn_exp2f(-r)) + __builtin_amdgcn_sinf(r * 0.1f + y) + __builtin_amdgcn_sqrtf(r * r + 0.5f);
            } else if (pat == 19) {
                int a = __float_as_int(x) >> 7, b = __float_as_int(y) >> 9;
#pragma unroll
                for (int i = 0; i < 48; ++i) { a = min(max(a * 3 + b, -1000000), 1000000) ^ (b >> (i & 7)); b = b + (a << (i & 3)) - i; }
                r = (float)(a & 0xffff) + (float)(b & 0xffff) * 65536.f;
            } else if (pat == 20) {
#pragma unroll
                for (int i = 0; i < 24; ++i) { const float f = floorf(r * 37.5f + y * i); const int q = (int)f; r = (float)(q % 97) * 0.013f + (r * 37.5f - f); }
            } else if (pat == 22) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const float c = 0.02f * i, a = r - y * c, b = r + 0.37f;
                    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(r), "v"(c), "v"(b), "v"(a) : "vcc");
                }
            } else if (pat == 23) {
#pragma unroll 1
                for (int i = 0; i < 48; ++i) {
                    if (r > 0.02f * i) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(y), "v"(-0.02f * i)); }
                    else { asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(0.37f)); }
                }
            } else if (pat == 24) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const float c = 0.02f * i, a = r - y * c, b = r + 0.37f;
                    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 s[2:3], %5, %1\n\ts_and_b64 vcc, vcc, s[2:3]\n\tv_cndmask_b32 %0, %3, %4, vcc"
                                 : "=v"(r) : "v"(r), "v"(c), "v"(b), "v"(a), "v"(y - 3.f) : "vcc", "s2", "s3");
                }
            } else if (pat == 27 || pat == 28 || pat == 29) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f a = v2f{x, y}, b = v2f{y * 0.5f + 0.25f, x * 0.5f - 0.25f};
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const v2f c = v2f{0.75f + 0.001f * i, 0.6f - 0.002f * i};
                    if (pat == 27) {
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));
                    } else if (pat == 28) {
                        asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(c), "v"(b));
                    } else {
                        float a0 = a[0], a1 = a[1];
                        asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5" : "+v"(a0), "+v"(a1) : "v"(c[0]), "v"(c[1]), "v"(b[0]), "v"(b[1]));
                        a = v2f{a0, a1};
                    }
                }
                r = a[0] + a[1];
            } else if (pat >= 42 && pat <= 44) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f a = v2f{x, y};
                const v2f b = v2f{y * 0.5f + 0.25f, x * 0.5f - 0.25f};
                asm volatile("s_mov_b32 s2, 0x3f7d70a4\n\ts_mov_b32 s3, 0x3f7ae148\n\ts_mov_b32 s4, 0x3c23d70a\n\ts_mov_b32 s5, 0xbc23d70a" ::: "s2", "s3", "s4", "s5");   // 0.99, 0.98, +-0.01
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    if (pat == 42) {
                        asm volatile("v_pk_mul_f32 %0, %0, s[2:3]\n\tv_pk_add_f32 %0, %0, s[4:5]" : "+v"(a) : : "s2", "s3", "s4", "s5");
                    } else if (pat == 43) {
                        float a0 = a[0], a1 = a[1];
                        asm volatile("v_mul_f32 %0, s2, %0\n\tv_mul_f32 %1, s3, %1\n\tv_add_f32 %0, s4, %0\n\tv_add_f32 %1, s5, %1" : "+v"(a0), "+v"(a1) : : "s2", "s3", "s4", "s5");
                        a = v2f{a0, a1};
                    } else {
                        asm volatile("v_pk_fma_f32 %0, %0, s[2:3], %1" : "+v"(a) : "v"(b) : "s2", "s3");
                    }
                }
                r = a[0] + a[1];
            } else if (pat >= 46 && pat <= 49) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f a = v2f{x, y};
                float t0 = x, t1 = y;
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    // constants 0.25 / -0.75 into s2 / s3, used once, then the registers take 159.0 / 3.0e5 (a stale read would show)
                    if (pat == 46)
                        asm volatile("s_mov_b32 s2, 0x3e800000\n\ts_mov_b32 s3, 0xbf400000\n\tv_pk_add_f32 %0, %0, s[2:3]\n\ts_mov_b32 s2, 0x431f0000\n\ts_mov_b32 s3, 0x48927c00\n\tv_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]"
                                     : "+v"(a) : : "s2", "s3");
                    else if (pat == 47)
                        asm volatile("s_mov_b32 s2, 0x3e800000\n\ts_mov_b32 s3, 0xbf400000\n\tv_pk_add_f32 %0, %0, s[2:3]\n\tv_mov_b32 %1, %2\n\tv_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]\n\ts_mov_b32 s2, 0x431f0000\n\ts_mov_b32 s3, 0x48927c00"
                                     : "+v"(a), "+v"(t0) : "v"(t1) : "s2", "s3");
                    else if (pat == 48) {
                        float a0 = a[0];
                        asm volatile("s_mov_b32 s2, 0x3e800000\n\tv_add_f32 %0, s2, %0\n\ts_mov_b32 s2, 0x431f0000\n\tv_mul_f32 %0, 0.5, %0" : "+v"(a0) : : "s2");
                        a = v2f{a0, a[1] * 0.5f + 0.25f};
                    } else
                        asm volatile("s_mov_b32 s2, 0x3e800000\n\ts_mov_b32 s3, 0xbf400000\n\tv_pk_add_f32 %0, %0, s[2:3]\n\ts_nop 4\n\ts_mov_b32 s2, 0x431f0000\n\ts_mov_b32 s3, 0x48927c00\n\tv_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]"
                                     : "+v"(a) : : "s2", "s3");
                }
                r = a[0] + a[1] + t0;
            } else if (pat >= 50 && pat <= 53) {
                float x0 = x, x1 = y, t0 = y, t1 = x, g0 = 0.f, g1 = 0.f;
                const float klo = 0.6f - 0.003f * y, nanv = __uint_as_float(0x7fc00000u);
                float cs = 1.f - 4.8f * x * x, sn = 3.1f * x - 4.9f * x * x * x;
                // fixed registers: xy = v[40:41], t = v[44:45], c = v[46:47], p = v[48:49], g = v[50:51], k = v[54:55]
#define SEQ(N, PRODUCERS, TAIL) asm volatile( \
                        "v_mov_b32 v40, %[x0]\n\tv_mov_b32 v41, %[x1]\n\tv_mov_b32 v44, %[t0]\n\tv_mov_b32 v45, %[t1]\n\tv_mov_b32 v54, %[klo]\n\tv_mov_b32 v55, 0\n\t" \
                        PRODUCERS N \
                        "v_pk_mul_f32 v[48:49], v[54:55], v[46:47] op_sel_hi:[0,1]\n\t" N \
                        "v_pk_mul_f32 v[44:45], v[44:45], v[48:49] op_sel:[0,1] op_sel_hi:[0,0]\n\t" N \
                        "v_pk_fma_f32 v[50:51], v[40:41], v[48:49], v[44:45] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" N \
                        "v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[44:45] op_sel_hi:[0,1,1]\n\t" N \
                        TAIL \
                        "\n\tv_mov_b32 %[x0], v40\n\tv_mov_b32 %[x1], v41\n\tv_mov_b32 %[t0], v44\n\tv_mov_b32 %[t1], v45\n\tv_mov_b32 %[g0], v50\n\tv_mov_b32 %[g1], v51" \
                        : [x0] "+v"(x0), [x1] "+v"(x1), [t0] "+v"(t0), [t1] "+v"(t1), [g0] "+v"(g0), [g1] "+v"(g1) \
                        : [klo] "v"(klo), [cs] "v"(cs), [sn] "v"(sn), [nanv] "v"(nanv) \
                        : "vcc", "s2", "s3", "s4", "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v54", "v55")
#define PROD "s_movk_i32 s4, 0x1f8\n\tv_cmp_class_f32 vcc, %[cs], s4\n\ts_mov_b32 s2, 0x3e800000\n\ts_mov_b32 s3, 0xbf400000\n\tv_cndmask_b32 v47, %[nanv], %[cs], vcc\n\tv_cndmask_b32 v46, %[nanv], %[sn], vcc\n\t"
#define NOPROD "s_mov_b32 s2, 0x3e800000\n\ts_mov_b32 s3, 0xbf400000\n\tv_mov_b32 v47, %[cs]\n\tv_mov_b32 v46, %[sn]\n\ts_nop 4\n\t"
#define TAILADD "v_pk_add_f32 v[44:45], v[50:51], s[2:3]\n\ts_mov_b32 s2, 0x431f0000"
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    if (pat == 50) SEQ("", PROD, TAILADD);
                    else if (pat == 51) SEQ("s_nop 3\n\t", PROD, TAILADD);
                    else if (pat == 52) SEQ("", NOPROD, TAILADD);
                    else SEQ("", PROD, "s_nop 0");
                    x0 = x0 * 0.25f + 0.1f; x1 = x1 * 0.25f + 0.2f; t0 = t0 * 0.01f + 0.3f; t1 = t1 * 0.01f + 0.1f;
                    cs = cs * 0.999f + 0.0001f; sn = sn * 0.998f + 0.0002f;
                }
                r = x0 + x1 + t0 + t1 + g0 + g1;
            } else if (pat >= 60 && pat <= 63) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f a = v2f{x, y}, b = v2f{0.99f - 0.01f * y, 0.98f + 0.01f * x}, c = v2f{0.01f * x, -0.01f * y};
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    if (pat == 60) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 61) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 62) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
                    else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
                }
                r = a[0] + a[1];
            } else if (pat >= 54 && pat <= 59) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f a = v2f{x, y}, b = v2f{0.99f - 0.01f * y, 0.98f + 0.01f * x}, c = v2f{0.01f * x, -0.01f * y};
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    if (pat == 54) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 55) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[0,0]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 56) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 57) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(b), "v"(c));
                    else if (pat == 58) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0]" : "+v"(a) : "v"(b), "v"(c));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                }
                r = a[0] + a[1];
            } else if (pat == 30) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const float c = 0.02f * i, a = r - y * c, b = r + 0.37f;
                    asm volatile("v_cmp_gt_f32 s[2:3], %1, %2\n\tv_cndmask_b32 %0, %3, %4, s[2:3]" : "=v"(r) : "v"(r), "v"(c), "v"(b), "v"(a) : "s2", "s3");
                }
            } else if (pat == 31) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const float c = 0.02f * i, a = r - y * c, b = r + 0.37f;
                    float t;
                    asm volatile("v_cmp_gt_f32 s[2:3], %2, %3\n\tv_cmp_lt_f32 s[4:5], %6, %2\n\tv_cndmask_b32 %1, %4, %5, s[4:5]\n\tv_cndmask_b32 %0, %1, %5, s[2:3]"
                                 : "=v"(r), "=&v"(t) : "v"(r), "v"(c), "v"(b), "v"(a), "v"(y - 0.5f) : "s2", "s3", "s4", "s5");
                }
            } else if (pat == 25) {
#pragma unroll 1
                for (int i = 0; i < 6; ++i) r = sinf(r * 3.1f + y) * 0.5f + cosf(r * 2.3f - y) * 0.5f;
            } else if (pat == 26) {
#pragma unroll 1
                for (int i = 0; i < 12; ++i) r = 1.f / (1.f + expf(-r * 1.7f + y));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) r = (r + 1.25f + 0.1f * i) / (y + 0.5f + 0.01f * i) - floorf((r + 1.25f + 0.1f * i) / (y + 0.5f + 0.01f * i)) + 0.125f;
            }
            return r;
        };
        unsigned qerr = 0;
        for (int it = 0; it < iters; ++it) {
            float x = (float)((salt + (unsigned)it * 2654435761u) >> 8) * (1.f / 16777216.f), y = (float)((salt * 31u + (unsigned)it * 40503u) >> 8) * (1.f / 16777216.f);
            const float r1 = kind(pattern, x, y);
            asm volatile("" : "+v"(x), "+v"(y));
            const float r2 = kind(pattern, x, y);
            qerr += __float_as_uint(r1) != __float_as_uint(r2);
        }
        if (qerr) { atomicAdd(errors + pattern, qerr); atomicAdd(errors + 48 + ((tid & 63) >> 4), qerr); }
        return;
    }
    if (pattern == 45) {
        auto stages = [](float x, float y, float (&o)[20]) __attribute__((noinline)) {
            const float cs = cosf(x * 3.1f), sn = sinf(x * 3.1f);
            const float k = 96.f / (160.f + y);
            float gx = k * cs * x - k * sn * y + 0.25f, gy = k * sn * x + k * cs * y - 0.75f;
            const float ix = (gx + 1.f) * 0.5f * 159.f, iy = (gy + 1.f) * 0.5f * 159.f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const bool vx0 = x0 >= 0 && x0 < 160, vx1 = x1 >= 0 && x1 < 160, vy0 = y0 >= 0 && y0 < 160, vy1 = y1 >= 0 && y1 < 160;
            const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
            const float w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
            const int cx0 = min(max(x0, 0), 159), cy1 = min(max(y1, 0), 159);
            o[0] = cs; o[1] = sn; o[2] = k; o[3] = gx; o[4] = gy; o[5] = ix; o[6] = iy; o[7] = fx; o[8] = fy; o[9] = (float)x0; o[10] = (float)y0;
            o[11] = wx1; o[12] = wy1; o[13] = w00; o[14] = w01; o[15] = w10; o[16] = w11; o[17] = (float)cx0; o[18] = (float)cy1;
            o[19] = 1.f / (1.f + expf(-gx));
        };
        unsigned qerr = 0;
        for (int it = 0; it < iters; ++it) {
            float x = (float)((salt + (unsigned)it * 2654435761u) >> 8) * (1.f / 16777216.f), y = (float)((salt * 31u + (unsigned)it * 40503u) >> 8) * (1.f / 16777216.f);
            float a[20], b[20];
            stages(x, y, a);
            asm volatile("" : "+v"(x), "+v"(y));
            stages(x, y, b);
            int first = -1;
#pragma unroll
            for (int i = 19; i >= 0; --i) if (__float_as_uint(a[i]) != __float_as_uint(b[i])) first = i;
            if (first >= 0) { ++qerr; atomicAdd(errors + 64 + first, 1); }
        }
        if (qerr) { atomicAdd(errors + pattern, qerr); atomicAdd(errors + 48 + ((tid & 63) >> 4), qerr); }
        return;
    }
    if (pattern == 40 || pattern == 41) {
        unsigned hold[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) { hold[j] = salt * (2u * j + 1u) + 0x01010101u * j; asm volatile("" : "+v"(hold[j])); }
        unsigned qerr = 0, rmask = 0;
        float busy = (float)tid;
        for (int it = 0; it < iters; ++it) {
            if (pattern == 40) {
                __builtin_amdgcn_s_sleep(20);
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) busy = fmaf(busy, 0.999f, 0.5f);
            }
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                asm volatile("" : "+v"(hold[j]));                                      // (the register itself is compared, not a recomputation)
                const bool bad = hold[j] != salt * (2u * j + 1u) + 0x01010101u * j;
                qerr += bad;
                rmask |= bad ? 1u << (j & 7) : 0u;
                if (bad) hold[j] = salt * (2u * j + 1u) + 0x01010101u * j;                // count an event once
            }
        }
        if (busy == 12345.f) errors[63] = 1;
        if (qerr) {
            atomicAdd(errors + pattern, qerr); atomicAdd(errors + 48 + ((tid & 63) >> 4), qerr);
            for (int j = 0; j < 8; ++j) if (rmask >> j & 1) atomicAdd(errors + 56 + j, 1);
        }
        return;
    }
    if (pattern == 15 || (pattern >= 32 && pattern <= 39)) {
        // 32-39: 15 with one ingredient removed (bit b of pattern - 32 + 1 ... see DROP_*): which part has to be there for the mismatches?
        auto chain = [](int drop, float x, float y) __attribute__((noinline)) {
            const bool no_trig = drop & 1, no_exp = drop & 2, no_weights = drop & 4, no_int = drop & 8, no_div = drop & 16;
            const float cs = no_trig ? 1.f - 4.8f * x * x : cosf(x * 3.1f), sn = no_trig ? 3.1f * x - 4.9f * x * x * x : sinf(x * 3.1f);
            const float k = no_div ? 0.6f - y * 0.003f : 96.f / (160.f + y);
            float gx = k * cs * x - k * sn * y + 0.25f, gy = k * sn * x + k * cs * y - 0.75f;
            const float ix = (gx + 1.f) * 0.5f * 159.f, iy = (gy + 1.f) * 0.5f * 159.f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            float wsum = wx1 + wy1;
            if (!no_weights) {
                const bool vx0 = x0 >= 0 && x0 < 160, vx1 = x1 >= 0 && x1 < 160, vy0 = y0 >= 0 && y0 < 160, vy1 = y1 >= 0 && y1 < 160;
                const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
                const float w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
                wsum = fmaf(w00, 1.5f, fmaf(w01, -2.5f, fmaf(w10, 3.5f, w11 * 4.5f)));
            }
            float r = wsum;
            if (!no_int) { const int cx0 = min(max(x0, 0), 159), cy1 = min(max(y1, 0), 159); r += (float)(cx0 * 160 + cy1) * 1e-3f; }
            if (!no_exp) r += no_div ? expf(-gx) : 1.f / (1.f + expf(-gx));
            return r;
        };
        static const int DROPS[8] = {1, 2, 4, 8, 16, 1 | 2, 1 | 2 | 16, 4 | 8};   // patterns 32 .. 39
        const int drop = pattern == 15 ? 0 : DROPS[pattern - 32];
        unsigned qerr = 0;
        for (int it = 0; it < iters; ++it) {
            float x = (float)((salt + (unsigned)it * 2654435761u) >> 8) * (1.f / 16777216.f), y = (float)((salt * 31u + (unsigned)it * 40503u) >> 8) * (1.f / 16777216.f);
            const float r1 = chain(drop, x, y);
            asm volatile("" : "+v"(x), "+v"(y));
            const float r2 = chain(drop, x, y);
            qerr += __float_as_uint(r1) != __float_as_uint(r2);
        }
        if (qerr) { atomicAdd(errors + pattern, qerr); atomicAdd(errors + 48 + ((tid & 63) >> 4), qerr); }
        return;
    }
    if (pattern == 13 || pattern == 14) {   // plain C++: what hipcc makes of it is the test
        __shared__ unsigned s_g[24], s_v[24], s_t[4];
        const unsigned lane = tid & 63, wid = tid >> 6;
        for (int it = 0; it < iters; ++it) {
            unsigned x[6], bp[6], va[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) { x[q] = (salt + q * 7919u) * ((unsigned)it * 2246822519u + 1u); bp[q] = x[q]; va[q] = wave_isum_valu(x[q]); }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
                for (int q = 0; q < 6; ++q) bp[q] += (unsigned)__shfl_xor((int)bp[q], d, 64);
#pragma unroll
            for (int q = 0; q < 6; ++q) err += bp[q] != va[q];
            n += 6;
            if (pattern == 14) {
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) { s_g[wid * 6 + q] = bp[q]; s_v[wid * 6 + q] = va[q]; }
                    s_t[wid] = (unsigned)it;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (tid < 24) { err += s_g[tid] != s_v[tid]; err += s_t[tid / 6] != (unsigned)it; n += 2; }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if (err) atomicAdd(errors + pattern, err);
        return;
    }
    for (int it = 0; it < iters; ++it) {
        const unsigned X = salt + (unsigned)it * 0x9E3779B9u, Y = ~X, J = X ^ 0x5a5a5a5au;
        switch (pattern) {
        case 0: {
            wr(a[0], X);
            err += rd(a[0]) != X; ++n;
        } break;
        case 1: {
            unsigned v = X;
            asm volatile("ds_write_b32 %1, %0\n\tv_mov_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a[0]), "v"(J) : "memory");
            err += rd(a[0]) != X; ++n;
        } break;
        case 2: {
            wr(a[1], Y);
            unsigned ad = a[0];
            asm volatile("ds_write_b32 %0, %1\n\tv_mov_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(ad) : "v"(X), "v"(a[1]) : "memory");
            err += rd(a[0]) != X; err += rd(a[1]) != Y; n += 2;
        } break;
        case 3: {
            wr(a[0], X); wr(a[1], Y);
            unsigned ad = a[0], r;
            asm volatile("ds_read_b32 %0, %1\n\tv_mov_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r), "+v"(ad) : "v"(a[1]) : "memory");
            err += r != X; ++n;
        } break;
        case 4: {
            wr(a[0], X); wr(a[1], Y);
            unsigned r0, r1;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(a[1]) : "memory");
            err += r0 != X; err += r1 != Y; n += 2;
        } break;
        case 5: {
#pragma unroll
            for (int s = 0; s < 8; ++s) wr(a[s], X + s);
            unsigned r[8];
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                         "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
#pragma unroll
            for (int s = 0; s < 8; ++s) err += r[s] != X + s;
            n += 8;
        } break;
        case 6: {
            wr(a[0], X); wr(a[1], Y);
            unsigned r0, r1, c;
            asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 %2, %0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(c) : "v"(a[0]), "v"(a[1]) : "memory");
            err += c != X; err += r1 != Y; n += 2;
        } break;
        case 7: {
            wr(a[0], J);
            unsigned r;
            asm volatile("ds_write_b32 %1, %2\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a[0]), "v"(X) : "memory");
            err += r != X; ++n;
        } break;
        case 8: {
            wr(a[0], J); wr(a[1], Y);
            unsigned r;
            asm volatile("ds_write_b32 %1, %3\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a[0]), "v"(a[1]), "v"(X) : "memory");
            err += r != Y; err += rd(a[0]) != X; n += 2;
        } break;
        case 9: {
            unsigned long long v = ((unsigned long long)Y << 32) | X, j = ((unsigned long long)J << 32) | J;
            const unsigned a64 = (unsigned)(size_t)&lds[0][2 * (tid & 127)] + (tid >> 7) * 1024 * 2;   // an 8-byte slot per lane (rows 0-1 / 2-3)
            asm volatile("ds_write_b64 %1, %0\n\tv_mov_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a64), "v"(j) : "memory");
            err += rd(a64) != X; err += rd(a64 + 4) != Y; n += 2;
        } break;
        case 10:
        case 11: {
            unsigned long long v0 = ((unsigned long long)Y << 32) | X, v1 = ((unsigned long long)(Y + 1) << 32) | (X + 1), j = ((unsigned long long)J << 32) | J;
            const unsigned a128 = (unsigned)(size_t)&lds[0][4 * (tid & 63)] + (tid >> 6) * 1024 * 2;       // a 16-byte slot per lane (two rows per wave)
            if (pattern == 10)
                asm volatile("ds_write2_b64 %2, %0, %1 offset1:1\n\tv_mov_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1) : "v"(a128), "v"(j) : "memory");
            else
                asm volatile("ds_write2_b64 %2, %0, %1 offset1:1\n\ts_nop 0\n\tv_mov_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1) : "v"(a128), "v"(j) : "memory");
            err += rd(a128) != X; err += rd(a128 + 4) != Y; err += rd(a128 + 8) != X + 1; err += rd(a128 + 12) != Y + 1; n += 4;
        } break;
        case 12: {
            wr(a[0], X);
            unsigned r, p;
            const unsigned src_lane4 = ((tid & 63) ^ 1) * 4;
            asm volatile("ds_bpermute_b32 %1, %3, %4\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r), "=&v"(p) : "v"(a[0]), "v"(src_lane4), "v"((unsigned)it * 977u + tid) : "memory");
            err += r != X; err += p != (unsigned)it * 977u + (tid ^ 1); n += 2;
        } break;
        default: break;
        }
    }
    if (err) atomicAdd(errors + pattern, err);
    if ((tid & 63) == 0) atomicAdd(checks + pattern, n * 64);
}

extern "C" int hazard_launch(int wgs, int pattern, int iters, unsigned *errors, unsigned *checks, void *stream) {
    hipLaunchKernelGGL(k_hazard, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), pattern, iters, errors, checks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

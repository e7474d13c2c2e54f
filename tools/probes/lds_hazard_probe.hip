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
    if (pattern == 15) {
        auto chain = [](float x, float y) __attribute__((noinline)) {
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
            return fmaf(w00, 1.5f, fmaf(w01, -2.5f, fmaf(w10, 3.5f, w11 * 4.5f))) + (float)(cx0 * 160 + cy1) * 1e-3f + 1.f / (1.f + expf(-gx));
        };
        unsigned qerr = 0;
        for (int it = 0; it < iters; ++it) {
            float x = (float)((salt + (unsigned)it * 2654435761u) >> 8) * (1.f / 16777216.f), y = (float)((salt * 31u + (unsigned)it * 40503u) >> 8) * (1.f / 16777216.f);
            const float r1 = chain(x, y);
            asm volatile("" : "+v"(x), "+v"(y));
            const float r2 = chain(x, y);
            qerr += __float_as_uint(r1) != __float_as_uint(r2);
        }
        if (qerr) { atomicAdd(errors + 15, qerr); atomicAdd(errors + 16 + ((tid & 63) >> 4), qerr); }
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

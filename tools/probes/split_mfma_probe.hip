// fp32 convolution contractions on the bf16 / f16 matrix cores by operand splitting - numerics and rate probe.
//   x = b0 + b1 + b2 exactly (three bf16 pieces, truncation split), a*b ~ b0b0' + b0b1' + b1b0' + b0b2' + b2b0' + b1b1'
//   (six v_mfma_f32_32x32x16_bf16 per k16 instead of eight v_mfma_f32_32x32x2_f32), or f16 hi/lo (three products).
// Checks (1) the A/B fragment layout assumed by conv_split.hip, (2) the error of each scheme against an fp64 sum for a
// conv-sized K with a wide dynamic range, (3) the issue rate of the six-product group.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/split_mfma_probe.hip -o tools/probes/split_mfma_probe.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned short bf_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short bf_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// A: [32][K] row-major, B: [K][32] row-major (K multiple of 16).  mode: 0 fp32 mfma, 1 bf16x6 trunc, 2 bf16x6 rne, 3 f16x3, 4 bf16x3 (2 pieces)
__global__ void k_gemm(const float *A, const float *B, float *D, int K, int mode) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + half], B[(k + half) * 32 + l31], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            float a[8], b[8];
            for (int i = 0; i < 8; ++i) { a[i] = A[l31 * K + k0 + 8 * half + i]; b[i] = B[(k0 + 8 * half + i) * 32 + l31]; }
            if (mode == 3) {
                f16x8 ah, al, bh, bl;
                for (int i = 0; i < 8; ++i) {
                    ah[i] = (_Float16)a[i]; al[i] = (_Float16)(a[i] - (float)ah[i]);
                    bh[i] = (_Float16)b[i]; bl[i] = (_Float16)(b[i] - (float)bh[i]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else {
                s16x8 a0, a1, a2, b0, b1, b2;
                for (int i = 0; i < 8; ++i) {
                    unsigned short p0, p1, p2;
                    float r;
                    if (mode == 1) { p0 = bf_trunc(a[i]); r = a[i] - bf_to_f(p0); p1 = bf_trunc(r); r -= bf_to_f(p1); p2 = bf_trunc(r); }
                    else { p0 = bf_rne(a[i]); r = a[i] - bf_to_f(p0); p1 = bf_rne(r); r -= bf_to_f(p1); p2 = bf_rne(r); }
                    a0[i] = (short)p0; a1[i] = (short)p1; a2[i] = (short)p2;
                    if (mode == 1) { p0 = bf_trunc(b[i]); r = b[i] - bf_to_f(p0); p1 = bf_trunc(r); r -= bf_to_f(p1); p2 = bf_trunc(r); }
                    else { p0 = bf_rne(b[i]); r = b[i] - bf_to_f(p0); p1 = bf_rne(r); r -= bf_to_f(p1); p2 = bf_rne(r); }
                    b0[i] = (short)p0; b1[i] = (short)p1; b2[i] = (short)p2;
                }
#define MM(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)
                if (mode == 4) { MM(a1, b0); MM(a0, b1); MM(a0, b0); }
                else { MM(a1, b1); MM(a2, b0); MM(a0, b2); MM(a1, b0); MM(a0, b1); MM(a0, b0); }
#undef MM
            }
        }
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
}

// rate: each wave issues `iters` groups of six independent-accumulator-friendly MFMAs on 4 accumulators
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    const float fa = (float)threadIdx.x, fb = 1.f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) {
#pragma unroll
                for (int s = 0; s < 6; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int s = 0; s < 3; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int K = 1152;
    std::vector<float> A(32 * K), B(K * 32);
    std::vector<double> ref(32 * 32), mag(32 * 32);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    // wide dynamic range: magnitudes spread over ~2^12, plus a few exact integers for the layout check
    for (auto &v : A) v = rnd() * ldexpf(1.f, rand() % 12 - 8);
    for (auto &v : B) v = rnd() * ldexpf(1.f, rand() % 12 - 4);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0, m = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
            ref[i * 32 + j] = s; mag[i * 32 + j] = m;
        }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    const char *names[5] = {"fp32 mfma 32x32x2", "bf16x6 trunc split", "bf16x6 rne split", "f16x3 hi/lo", "bf16x3 (two pieces)"};
    std::vector<float> D(32 * 32);
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, rms = 0;
        for (int e = 0; e < 1024; ++e) {
            const double err = std::fabs((double)D[e] - ref[e]) / mag[e];
            worst = std::max(worst, err); rms += err * err;
        }
        printf("%-22s K=%d  max |err| / sum|a b| = %.3e   rms = %.3e\n", names[mode], K, worst, std::sqrt(rms / 1024));
    }
    // subnormal / huge operands through the bf16 split (fp32 range): a*b with a = 1e-30, b = 1e30 and a = 3e38 * 1e-38
    {
        std::vector<float> A2(32 * 16, 0.f), B2(16 * 32, 0.f);
        A2[0] = 1.2345678e-30f; B2[0] = 7.6543211e29f;      // D[0][0]
        A2[16 + 1] = 3.1e38f; B2[32 + 1] = 1.07e-38f;        // D[1][1]
        A2[32 + 2] = 1.0e-41f; B2[64 + 2] = 1.0e38f;   // subnormal input
        CK(hipMemcpy(dA, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B2.data(), B2.size() * 4, hipMemcpyHostToDevice));
        for (int mode : {0, 1}) {
            hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dD, 16, mode);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            printf("range check %-20s: %.8e (exact %.8e)  %.8e (exact %.8e)  %.8e (exact %.8e)\n", names[mode], D[0], (double)A2[0] * B2[0], D[33],
                   (double)A2[17] * B2[33], D[66], (double)A2[34] * B2[66]);
        }
    }
    // rate
    float *dout;
    CK(hipMalloc(&dout, 1024 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(256), 0, 0, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(256), 0, 0, dout, iters);
            if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1024), dim3(256), 0, 0, dout, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // fp32-equivalent flops: one k16 step of a 32x32 tile = 2*32*32*16
        const double eq = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("rate %-8s: %.3f ms -> %.1f fp32-equivalent TFLOP/s\n", mode == 0 ? "bf16x6" : mode == 1 ? "f16x3" : "fp32", ms, eq / ms / 1e9);
    }
    return 0;
}

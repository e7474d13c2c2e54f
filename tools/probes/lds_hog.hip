// A synthetic neighbour for the persistent plan kernel (tools/plan_stress.py, PLAN_STRESS_HOG=synth:<mode>): workgroups with the
// resource shape of the tap-pair stem kernel (512 threads, ~150 KB of dynamic LDS, <= 184 VGPRs so that two of its waves and one plan
// wave fit a SIMD) that keep the matrix pipes busy.  mode 0: never touches its LDS; 1: reads and writes it (16-byte accesses over
// the whole allocation); 2: no matrix instructions either (s_sleep loop); 3: LDS traffic without matrix instructions.  Built by the script: hipcc -shared -fPIC.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_hog(int iters, int mode, int lds_bytes, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    f32x16 acc[8];
    const float seed = (float)tid;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = seed;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + e + (tid & 3)); b[e] = (__bf16)(0.5f * e); }
    const int n16 = lds_bytes / 16;
    float4 carry = make_float4(tid, 1.f, 2.f, 3.f);
    if (mode == 2)
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
    else
    for (int it = 0; it < iters; ++it) {
        if (mode != 3)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        if (mode == 1 || mode == 3) {
            const int i0 = (tid + it * 512) % n16, i1 = (tid * 7 + it * 131) % n16;
            reinterpret_cast<float4 *>(smem)[i0] = carry;
            const float4 v = reinterpret_cast<float4 *>(smem)[i1];
            carry.x += v.y * 1e-9f;
        }
    }
    float s = carry.x;
    for (int k = 0; k < 8; ++k)
        for (int r = 0; r < 16; r += 5) s += acc[k][r];
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int hog_launch(int wgs, int lds_bytes, int mode, int iters, float *sink, void *stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        attr = true;
    }
    hipLaunchKernelGGL(k_hog, dim3(wgs), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), iters, mode, lds_bytes, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

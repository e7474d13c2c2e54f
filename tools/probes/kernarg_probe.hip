// How much of a small kernel's duration is the fetch of its argument segment?  Three variants of the same kernel:
//   k_struct  - arguments in one by-value struct (what liblav_amd's kernels do): read with s_load after launch
//   k_flat    - the same arguments as scalars / pointers, no preload
//   k_preload - k_flat compiled in a translation unit with -mllvm -amdgpu-kernarg-preload-count=16 (see build line)
// Back-to-back launches on one stream, HIP events, many workgroups that each do one dependent load + store.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/probes/kernarg_probe.hip -o /tmp/kernarg_probe
#include <hip/hip_runtime.h>

#include <cstdio>

struct Args {
    const float *x;
    float *y;
    int n;
    float s;
    int pad[8];
};

__global__ void k_struct(Args a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.y[i] = a.x[i] * a.s;
}

__global__ void k_flat(const float *x, float *y, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * s;
}

template <typename F>
float time_us(F launch, int reps, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int n = 256 * 512;
    float *x, *y;
    hipMalloc(&x, n * 4);
    hipMalloc(&y, n * 4);
    hipMemset(x, 0, n * 4);
    hipStream_t st;
    hipStreamCreate(&st);
    Args a{x, y, n, 2.f, {0}};
    for (int wgs : {1, 512}) {
        const float t_s = time_us([&] { hipLaunchKernelGGL(k_struct, dim3(wgs), dim3(256), 0, st, a); }, 2000, st);
        const float t_f = time_us([&] { hipLaunchKernelGGL(k_flat, dim3(wgs), dim3(256), 0, st, (const float *)x, y, n, 2.f); }, 2000, st);
        printf("%4d workgroups: struct args %.2f us/launch, flat args (preload build) %.2f us/launch\n", wgs, t_s, t_f);
    }
    // the same inside a graph of 200 kernel nodes
    for (int flat = 0; flat < 2; ++flat) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) {
            if (flat) hipLaunchKernelGGL(k_flat, dim3(64), dim3(256), 0, st, (const float *)x, y, n, 2.f);
            else hipLaunchKernelGGL(k_struct, dim3(64), dim3(256), 0, st, a);
        }
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const float t = time_us([&] { hipGraphLaunch(ge, st); }, 50, st);
        printf("graph of 200 x 64-workgroup kernels, %s args: %.2f us per kernel\n", flat ? "flat/preload" : "struct", t / 200);
    }
    return 0;
}

// Point painting: LiDAR -> camera projection + semantic gather, all cameras in one pass.
//
// Replaces InferModel.forward_paint / point_painting / CoordConverter.forward of the reference
// (team_code_v2/model_inference.py:44-50, 75-93, 280-297): per camera three skinny matmuls, a
// divide, a .long() truncation, a masked gather and a masked scatter - ~30 small launches - become one
// kernel: one thread per point, cameras unrolled, later cameras overwrite earlier ones.
//
// Arithmetic contract (shared with oracle/paint.py): float32, every product and every sum rounded
// separately (no FMA contraction), terms in k order:  ((m0*x + m1*y) + m2*z) + m3*w.
// HBM traffic: N*(lidar_dim + lidar_dim + sem_c)*4 bytes streamed; the 3x5x288x256 probability maps
// (4.4 MB) are gathered through L2.
#include "common.hpp"

// parity-critical float32 arithmetic: no fused multiply-add contraction anywhere in this file
// (HIP's __fadd_rn/__fmul_rn are plain operators that clang would otherwise fuse)
#pragma clang fp contract(off)

namespace {
using namespace lav;
constexpr int MAX_CAM = 4;

struct PaintArgs {
    lav_camera cam[MAX_CAM];
    int ncam, n, lidar_dim, sem_c, h, w;
};

__device__ __forceinline__ float mv4(const float *m, float x, float y, float z, float w) {
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), __fmul_rn(m[3], w));
}
__device__ __forceinline__ float mv3(const float *m, float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z));
}
// Tensor.long() of a float32, narrowed to int32: INT32_MIN stands for "out of range / non finite"
__device__ __forceinline__ int to_long(float v) {
    if (!(fabsf(v) < 2147483520.f)) return INT32_MIN;  // also catches NaN and inf
    return (int)v;                                      // truncation toward zero
}

template <int SEM_C>
__global__ __launch_bounds__(256) void k_paint(PaintArgs a, const float *__restrict__ lidar, const float *__restrict__ sem,
                                               float *__restrict__ fused, int *__restrict__ uvz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float *p = lidar + (long)i * a.lidar_dim;
    const float x = p[0], y = p[1], z = p[2];
    float painted[SEM_C];
#pragma unroll
    for (int c = 0; c < SEM_C; ++c) painted[c] = 0.f;
    const long plane = (long)a.h * a.w;
    for (int cam = 0; cam < a.ncam; ++cam) {
        const lav_camera &cm = a.cam[cam];
        const float wx = mv4(cm.l2w + 0, x, y, z, 1.f), wy = mv4(cm.l2w + 4, x, y, z, 1.f);
        const float wz = mv4(cm.l2w + 8, x, y, z, 1.f), ww = mv4(cm.l2w + 12, x, y, z, 1.f);
        const float cx = mv4(cm.w2c + 0, wx, wy, wz, ww), cy = mv4(cm.w2c + 4, wx, wy, wz, ww);
        const float cz = mv4(cm.w2c + 8, wx, wy, wz, ww);
        const float X = cy, Y = -cz, Z = cx;  // model_inference.py:289
        const float p0 = mv3(cm.K + 0, X, Y, Z), p1 = mv3(cm.K + 3, X, Y, Z), p2 = mv3(cm.K + 6, X, Y, Z);
        const float den = __fadd_rn(1e-5f, p2);
        const int u = to_long(__fdiv_rn(p0, den)), v = to_long(__fdiv_rn(p1, den)), d = to_long(p2);
        if (uvz) {
            int *o = uvz + ((long)cam * a.n + i) * 3;
            o[0] = u; o[1] = v; o[2] = d;
        }
        if (d >= 0 && u >= 0 && u < a.w && v >= 0 && v < a.h) {
            const float *s = sem + (long)cam * (SEM_C + 1) * plane + (long)v * a.w + u;
            const float keep = __fsub_rn(1.f, s[0]);  // forward_paint: sem[:,1:] * (1 - sem[:,:1])
#pragma unroll
            for (int c = 0; c < SEM_C; ++c) painted[c] = __fmul_rn(s[(c + 1) * plane], keep);
        }
    }
    float *o = fused + (long)i * (a.lidar_dim + SEM_C);
    for (int d = 0; d < a.lidar_dim; ++d) o[d] = p[d];
#pragma unroll
    for (int c = 0; c < SEM_C; ++c) o[a.lidar_dim + c] = painted[c];
}
}  // namespace

extern "C" int lav_paint(const float *lidar, int n, int lidar_dim, const float *sem, int ncam, int sem_c, int h, int w,
                         const lav_camera *h_cams, float *fused, int *uvz, void *stream) {
    LAV_REQUIRE(n >= 0 && lidar_dim >= 3 && sem && h_cams && (fused || n == 0), "lav_paint: bad argument");
    LAV_REQUIRE(ncam >= 1 && ncam <= MAX_CAM, "lav_paint: ncam %d outside [1,%d]", ncam, MAX_CAM);
    LAV_REQUIRE(sem_c == 4, "lav_paint: sem_c %d not instantiated (4)", sem_c);
    if (n == 0) return LAV_OK;
    PaintArgs a;
    for (int c = 0; c < ncam; ++c) a.cam[c] = h_cams[c];
    for (int c = ncam; c < MAX_CAM; ++c) a.cam[c] = h_cams[0];
    a.ncam = ncam; a.n = n; a.lidar_dim = lidar_dim; a.sem_c = sem_c; a.h = h; a.w = w;
    const int tok = timer_begin("paint", static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL((k_paint<4>), dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a, lidar, sem, fused, uvz);
    timer_end(tok, static_cast<hipStream_t>(stream));
    LAV_LAUNCH_CHECK();
    return LAV_OK;
}

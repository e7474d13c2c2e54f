// Entry points of conv_f16.hip (the LAV_CONV_F16X3 kernels) for conv.hip: the two translation units include
// conv_split_kernel.hpp in their own anonymous namespaces, so the argument block crosses the boundary as bytes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace lav {
constexpr int F16_PARTS = 512;     // maxima written by launch_absmax_parts
constexpr int AMAX_MAX = 16384;    // most per-workgroup maxima a layer may hand to its consumer (beyond: measured by a launch)

// parts[0 .. F16_PARTS) = largest finite |x| of the channel window (images below *n_valid, when given)
int launch_absmax_parts(const float *x, int batch, int in_c_total, int in_c_offset, int cin, long plane, float *parts, const int *n_valid, hipStream_t st);

// k_conv_split_f16<mp, mc, wpx, NT by plane, G = args.tap_group, tp>
int launch_split_f16(const void *split_args, size_t args_bytes, int mp, int mc, int wpx, int tp, unsigned gx, unsigned gy, unsigned gz, size_t lds, hipStream_t st);
}  // namespace lav

// One step of a GRU over R rows (gru_seq.hip): h_out = GRUCell(x or W_ih u + b_ih, h_prev), recurrent GEMM on MFMA.
// Shared by the training layer (lav_gru_seq_forward) and the many-row path of the plan decoder (gru.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace lav {

struct GruFwdArgs {
    const float *h_prev;        // row r at h_prev + r * h_prev_stride  (h0, or the previous step's output)
    const float *x;             // input-side pre-activations (b_ih included): row r at x + r * x_stride, [3H] = (r, z, n); used when u == nullptr
    const float *u;             // or the raw input: row r at u + r * u_stride, [I], projected in the kernel with w_ih [3H][I], b_ih [3H]
    const float *w_ih, *b_ih;
    const float *w_hh, *b_hh;   // [3H][H], [3H]
    float *h_out;               // row r at h_out + r * h_out_stride
    float *tape;                // (r, z, n, W_hn h + b_hn) of this step: row r at tape + r * tape_stride, [4][H]; may be null
    long h_prev_stride, x_stride, u_stride, h_out_stride, tape_stride;
    int R, H, I;
};

// H must be a multiple of 16, (R + 15) / 16 <= 65535.  Only enqueues.
void launch_gru_fwd_step(const GruFwdArgs &a, hipStream_t st);

}  // namespace lav

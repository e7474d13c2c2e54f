/*
 * lav_amd.h - C ABI of liblav_amd.so: the MI355X (gfx950) kernels of the LAV per-frame
 * perception -> prediction -> planning forward path.
 *
 * The reference (dotchen/LAV) has no native code and no FFI: its "operator boundary" for this
 * path is a set of third-party/torch ops called from Python (level B3).
 * Each entry point below replaces one such call site; the reference file:line it replaces is
 * cited on the declaration.  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add, and lav_amd/_lib.py is that stub as shipped here.
 *
 * Conventions
 *   - plain C: device pointers are raw `void*`/`float*` into HBM, sizes are ints, `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream).  No torch types.
 *   - every call only ENQUEUES work on `stream`; nothing synchronises the host.
 *   - return value: 0 on success, a negative LAV_E* code otherwise; lav_last_error() returns a
 *     thread-local message.  Inputs are never written.
 *   - tensors are dense row-major float32 unless stated; index outputs are int32.
 *   - there is no CPU fallback: on a box without a gfx950 device every entry point fails.
 */
#ifndef LAV_AMD_H
#define LAV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAV_ABI_VERSION 28

#define LAV_OK 0
#define LAV_EINVAL (-1)    /* bad argument / unsupported shape */
#define LAV_EHIP (-2)      /* a HIP runtime call failed */
#define LAV_EWORKSPACE (-3)/* workspace too small */

int lav_abi_version(void);
const char *lav_last_error(void);
/* number of HIP devices visible, or a negative error.  Used by the loader to fail loudly. */
int lav_device_count(void);

/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for
 * the live roofline figure).  lav_profile_enable(slots>0) arms it: the first `slots` launches of each
 * tracked kernel ("pointnet_scatter", "pillar_prep", "conv2d", "paint", "gru_cast", "gru_plan") get an
 * event pair; lav_profile_enable(0) disarms.  lav_profile_read synchronises those events and returns
 * the summed duration and the launch count.  Not part of the reference's surface. */
int lav_profile_enable(int slots);
/* forget recorded launches but keep the (already created) events: call after a warm-up, before the timed region */
int lav_profile_reset(void);
int lav_profile_read(const char *kernel, double *total_ms, int *launches);

/* ------------------------------------------------------------------------------------------
 * 1. PointPillars: dynamic voxelisation + decoration + PointNet + scatter-max + dense canvas.
 *    Replaces PointPillarNet.forward, lav/models/point_pillar.py:92-116, i.e. grid_locations
 *    (:70-79), coords.unique(dim=0) (:82), decorate incl. torch_scatter.scatter_mean (:55-68),
 *    DynamicPointNet incl. torch_scatter.scatter_max (:28-35) and scatter_points (:87-90).
 * ------------------------------------------------------------------------------------------ */
typedef struct lav_grid {
    float min_x, max_x, min_y, max_y; /* metres; keep x in [min_x,max_x), y in [min_y,max_y) */
    float ppm;                        /* pixels per metre */
    int nx, ny;                       /* (max_x-min_x)*ppm, (max_y-min_y)*ppm : number of xi / yi cells */
} lav_grid;

/* PointNet weights with eval-mode BatchNorm1d folded in (host does the folding once):
 *   h1 = relu(f @ w1 + b1),  h2 = relu(h1 @ w2 + b2);  w1 [D+5][C], w2 [C][C], row-major. */
typedef struct lav_pointnet {
    const float *w1, *b1, *w2, *b2;
    int num_input; /* D+5 (16 for the v2 agent) */
    int channels;  /* C (64) */
} lav_pointnet;

/* Bytes of scratch lav_pillar_scatter needs for `batch` clouds of at most `max_points` points.
 *
 * Workspace contract: the head of a pillar workspace holds state that is ZERO AT REST (two sets of arrival counters
 * used alternately and two epoch words), which is what lets a call run as two launches with no memset.  Zero-fill a
 * workspace ONCE after allocating it (lav_pillar_workspace_init, or any memset), keep it for ONE (batch, grid) geometry
 * (max_points may vary up to the size it was allocated for) and do not write to it between calls; every call leaves it
 * clean for the next one.  Re-run lav_pillar_workspace_init to reuse the memory for another geometry or after a
 * failed launch.  Calls sharing a workspace must be ordered on one stream. */
size_t lav_pillar_workspace_bytes(int batch, int max_points, const lav_grid *grid);
int lav_pillar_workspace_init(void *workspace, size_t workspace_bytes, void *stream);

/*
 * points      [batch][max_points][D]   (a single cloud: batch=1, max_points=N)
 * h_num_points host array [batch]: only the first h_num_points[b] rows of cloud b are used (:98)
 * canvas      out [batch][C][ny][nx]; every element is written (empty cells = 0)
 * Optional index outputs (NULL to skip; they cost extra passes and exist for parity/training):
 *   unique_coords out [<= total kept][3] rows (b, xi, yi) sorted lexicographically  (:82)
 *   inverse       out [total kept]  pillar row of each kept point, in input order   (:82)
 *   counts        out [2] = { number of pillars P, number of kept points }
 */
int lav_pillar_scatter(const float *points, const int *h_num_points, int batch, int max_points, int D,
                       const lav_grid *grid, const lav_pointnet *net, float *canvas,
                       int *unique_coords, int *inverse, int *counts,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The same call leaving a BOUND of the canvas for the layer that reads it (round 6): amax_parts[0 .. lav_pillar_amax_count(batch, grid))
 * receives, per workgroup of the canvas kernel, the largest finite value its PointNet produced (>= every value it wrote; the
 * canvas holds ReLU outputs, so this bounds |canvas|).  LAV_CONV_F16X3 layers take their activation scale from such parts
 * (lav_conv2d_amax) instead of measuring their input with a launch of their own: ConvBackbone's first convolution,
 * team_code_v2/models/lidar.py:57.  amax_parts NULL = lav_pillar_scatter. */
int lav_pillar_amax_count(int batch, const lav_grid *grid);
int lav_pillar_scatter_amax(const float *points, const int *h_num_points, int batch, int max_points, int D,
                            const lav_grid *grid, const lav_pointnet *net, float *canvas,
                            int *unique_coords, int *inverse, int *counts, float *amax_parts,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * 2. Point painting: LiDAR->camera projection + semantic gather for up to 4 cameras.
 *    Replaces InferModel.forward_paint / point_painting / CoordConverter.forward,
 *    team_code_v2/model_inference.py:44-50, 75-93, 280-297.
 * ------------------------------------------------------------------------------------------ */
typedef struct lav_camera {
    float K[9];    /* row-major 3x3 intrinsics                       (model_inference.py:259-262) */
    float l2w[16]; /* row-major 4x4 lidar_to_world                    (:264-266) */
    float w2c[16]; /* row-major 4x4 world_to_cam                      (:268-271) */
} lav_camera;

/*
 * lidar    [n][lidar_dim]  (xyz in columns 0..2; lidar_dim >= 3)
 * sem      [ncam][sem_c+1][h][w] softmax probabilities; channel 0 is the "other" class
 * fused    out [n][lidar_dim + sem_c] = cat(lidar, painted) where, per camera in order (later
 *          cameras overwrite), painted = sem[cam][1+c][v][u] * (1 - sem[cam][0][v][u]) for points
 *          whose truncated (u, v, depth) satisfy depth>=0, 0<=u<w, 0<=v<h; else 0.
 * uvz      optional out [ncam][n][3] int32 truncated projections (INT32_MIN where the reference's
 *          .long() is out of range); NULL to skip.
 */
int lav_paint(const float *lidar, int n, int lidar_dim, const float *sem, int ncam, int sem_c, int h, int w,
              const lav_camera *h_cams, float *fused, int *uvz, void *stream);

/* ------------------------------------------------------------------------------------------
 * 3. GRU waypoint decoders.  Replaces UniPlanner.cast (team_code_v2/models/uniplanner.py:288-308),
 *    UniPlanner.plan/_plan (:255-286) and the cuDNN GRU calls inside them.
 *    PyTorch GRU parameter layout: w_ih [3H][I], w_hh [3H][H], b_ih [3H], b_hh [3H], gates r,z,n.
 * ------------------------------------------------------------------------------------------ */
/*
 * cast: for each of num_cmds GRUs (I=embd_dim, H=hidden): constant input embd[b] at every step,
 *       h0 = 0, T steps; locs = cumsum_t( mlp_w @ h_t + mlp_b ).
 * embd [B][embd_dim]; w_ih [num_cmds][3H][embd_dim]; w_hh [num_cmds][3H][H]; b_ih,b_hh [num_cmds][3H];
 * mlp_w [num_cmds][2][H]; mlp_b [num_cmds][2];  out [B][num_cmds][T][2]
 */
int lav_gru_cast(const float *embd, int B, int embd_dim, int H, int num_cmds, int T,
                 const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                 const float *mlp_w, const float *mlp_b, float *out,
                 void *workspace, size_t workspace_bytes, void *stream);
size_t lav_gru_cast_workspace_bytes(int B, int embd_dim, int H, int num_cmds, int T);
/*
 * embed + cast: the tail of the embedder and everything that hangs on it, in one launch per batch
 * (uniplanner.py:36-40 AdaptiveAvgPool2d + Flatten, :50-53 cast_cmd_pred, :288-308 cast;
 * model_inference.py:164-165, 240-251 transform_points + translate).
 * feat [B][embd_dim][hw]: the embedder's last feature map (hw pixels per channel; hw = 1: the embedding itself);
 * embd_out [B][embd_dim] or NULL: its spatial mean; cmd_w [num_cmds][embd_dim], cmd_b [num_cmds], cmds_out [B][num_cmds]
 * (all three or none): sigmoid(cmd_w . embd + cmd_b); oris [B], locs [B][2] (each optional): every decoded waypoint
 * (x, y) becomes (x cos o - y sin o, x sin o + y cos o) + loc.  Everything else as lav_gru_cast; honours lav_batch_limit.
 */
int lav_embed_cast(const float *feat, int B, int embd_dim, int hw, float *embd_out, int H, int num_cmds, int T,
                   const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                   const float *mlp_w, const float *mlp_b, const float *cmd_w, const float *cmd_b, float *cmds_out,
                   const float *oris, const float *locs, float *out, void *stream);

/*
 * plan: GRU(I=4 -> H), h0 = embd[b]; iteration it, command c:
 *         u_t = [ nxp[b]*ppm/crop_size*2-1 , loc_{it-1}[b][c][t] ],  loc_{-1} = cast_locs
 *         loc_it[b][c] = cumsum_t(mlp(h_t)) + loc_{it-1}[b][c]
 * cmd >= 0: only that command branch is evaluated (branches never interact) and `out` is
 *           [B][iters][1][T][2];  cmd = -1: all branches, out [B][iters][num_cmds][T][2].
 * embd [B][H]; nxp [B][2]; cast_locs [B][num_cmds][T][2]; w_ih [3H][4]; w_hh [3H][H]; mlp_w [2][H].
 */
int lav_gru_plan(const float *embd, const float *nxp, const float *cast_locs, int B, int H, int num_cmds, int T,
                 int iters, int cmd, float pixels_per_meter, float crop_size,
                 const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                 const float *mlp_w, const float *mlp_b, float *out,
                 void *workspace, size_t workspace_bytes, void *stream);
size_t lav_gru_plan_workspace_bytes(int B, int H, int num_cmds, int T);
/* A plan workspace must be ZERO before its first use and belongs to one stream.  Its first 512 bytes are a control block:
 * [0, 256) sticky counters {aborted persistent launches, persistent launches} since it was zero-filled, [256, 512) the status
 * and diagnosis words of the last persistent launch (cleared at every launch). */
/*
 * lav_gru_plan runs all iters*T dependent steps in ONE persistent launch when B*(1 or num_cmds) <= 6: its H/8 workgroups
 * exchange the hidden state through HBM and must be co-resident.  Every wait is bounded; a launch that times out (the chip
 * oversubscribed by other streams) fills the WHOLE `out` with NaN and raises the status word of its workspace:
 *   lav_gru_plan_status  copies that word to *h_status (0 = completed, 1 = aborted) - it SYNCHRONISES `stream`;
 *   lav_gru_plan_steps   same contract as lav_gru_plan on the step-per-launch path (no co-residency requirement, ~10 % slower):
 *                        what a caller runs after an abort.
 */
int lav_gru_plan_status(const void *workspace, size_t workspace_bytes, int B, int H, int num_cmds, int cmd,
                        int *h_status, void *stream);
/* lav_gru_plan_diag: the 16 diagnosis words of the last persistent launch (SYNCHRONISES `stream`): [0] status, [1] workgroups
 * that entered the kernel, [2] 1 + workgroup / [3] wave / [4] step epoch of the first wave that gave up, [5] its spin count,
 * [6] the state granule it was waiting for, [7] the tag it last saw there, [8] microseconds from its kernel entry to the abort,
 * [9] workgroups that ran to completion (H/8 after a good launch); sticky since the workspace was zero-filled: [10] aborted
 * launches, [11] all persistent launches (a plan workspace must be ZERO before its first use).  No counterpart in the reference (cuDNN's GRU cannot
 * time out); it exists so that an abort can be told apart from lost co-residency vs a lost hand-off. */
int lav_gru_plan_diag(const void *workspace, size_t workspace_bytes, int B, int H, int num_cmds, int cmd,
                      int *h_words16, void *stream);
int lav_gru_plan_steps(const float *embd, const float *nxp, const float *cast_locs, int B, int H, int num_cmds, int T,
                       int iters, int cmd, float pixels_per_meter, float crop_size,
                       const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                       const float *mlp_w, const float *mlp_b, float *out,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * 3b. GRU over a sequence with a saved tape - the TRAINING side of the same decoders: what nn.GRU (cuDNN there, MIOpen
 *     here) does under UniPlanner.cast / _plan / BEVPlanner in train mode (team_code_v2/models/uniplanner.py:255-308,
 *     lav/models/bev_planner_v2.py, called from lav/lav_final_v2.py:140-259 and lav/lav_privileged_v2.py:110-159).
 *     One launch per time step (recurrent GEMM on MFMA fused with the gates); the input projection and the weight
 *     gradients are plain GEMMs over all (row, step) pairs and are the caller's (see lav_amd/ops.py:gru_seq).
 *
 *     x      : input-side pre-activations W_ih u + b_ih, gate order (r, z, n): [R][T][3H] when x_per_step, else [R][3H]
 *              (the same input at every step - the cast decoders)
 *     h0     : [R][H];  w_hh [3H][H], b_hh [3H];  out [R][T][H];  tape [R][T][4][H] = (r, z, n, W_hn h + b_hn) per step
 *              (may be NULL for an inference-only forward)
 *   backward : dout [R][T][H] -> dx [R][T][3H] (gradient of x per step; sum over T for a shared x), dgh [R][T][3H]
 *              (gradient of the recurrent-side pre-activations: dW_hh = dgh^T [h0, out[:, :-1]], db_hh = sum dgh), dh0 [R][H].
 *              w_hh_t = W_hh transposed, [H][3H].  H must be a multiple of 16.  No atomics: results are bit-reproducible.
 */
int lav_gru_seq_forward(const float *x, int x_per_step, const float *h0, const float *w_hh, const float *b_hh,
                        int R, int T, int H, float *out, float *tape, void *stream);
size_t lav_gru_seq_backward_workspace_bytes(int R, int H);
int lav_gru_seq_backward(const float *dout, const float *tape, const float *out, const float *h0, const float *w_hh_t,
                         int R, int T, int H, float *dx, float *dgh, float *dh0,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * 4. 2-D convolutions on the matrix cores.  fp32 in, fp32 out, fp32 accumulation; the products are either exact fp32
 *    (v_mfma_f32_32x32x2_f32; precision = LAV_CONV_F32) or - the default, LAV_CONV_BF16X6 - each operand is split exactly into
 *    three bf16 pieces and the six leading partial products run on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): the same
 *    accuracy class as the fp32 chain (max error 3e-7 vs 6e-7 of sum |a||b| at K = 1152) at up to twice its rate; the launch
 *    plan picks per layer and shape among the tiled, direct and split-operand kernels.  Replaces the cuDNN calls behind ConvBackbone / Head
 *    (team_code_v2/models/lidar.py:48-161) and, with the same kernel, the ResNet-18 embedder
 *    (lav/models/resnet.py:39-82,235-247).  NCHW activations.
 *
 *    y = epilogue( conv(x, w) ),  epilogue in this order (each optional):
 *        + bias[co] -> ReLU (relu_pre) -> * scale[co] + shift[co] -> + residual -> ReLU (relu_post)
 *        -> sigmoid
 *    (the reference's BEV blocks are Conv -> ReLU -> BatchNorm, lidar.py:57-60, i.e. relu_pre +
 *    scale/shift; its ResNet blocks are Conv -> BatchNorm (-> +identity) -> ReLU, i.e. scale/shift
 *    (+residual) + relu_post.)
 * ------------------------------------------------------------------------------------------ */
typedef struct lav_conv {
    int batch;
    int in_c_total, in_c_offset, cin; /* x is [batch][in_c_total][h][w]; channels [in_c_offset, +cin) are read */
    int h, w;
    int cout, kh, kw;
    int stride, pad_h, pad_w, dil_h, dil_w;
    int transposed, out_pad;             /* 1: ConvTranspose2d(stride, padding=pad_h/pad_w, output_padding) */
    int out_c_total, out_c_offset;       /* y is [batch][out_c_total][oh][ow]; channels [out_c_offset, +cout)
                                            are written (fused torch.cat, lidar.py:143) */
    int relu_pre, relu_post;
    int sigmoid;       /* 0: none; k > 0: sigmoid on output channels >= k-1 of this convolution (1 = all of them) */
    int target_cus;    /* 0 = plan for the whole chip (256 CUs); n: plan tiles / split-K to fill n CUs - for layers of a
                          network that runs on a side stream next to another network's kernels */
    float pad_value;   /* what out-of-image taps read (0 = zero padding).  A network whose input normalisation x' = s*x + t
                          has been folded into its first convolution pads the RAW image with -t/s instead */
    int precision;     /* how the fp32 contraction is evaluated: LAV_CONV_F32 = v_mfma_f32_32x32x2_f32 only (bit-for-bit an
                          fmaf chain); LAV_CONV_BF16X6 = layers whose plan favours it run on the bf16 matrix cores with every
                          fp32 operand split exactly into three bf16 pieces and the six leading partial products accumulated
                          in fp32 (error of an fp32 dot product, 2.4x the matrix rate; fp32 subnormal inputs are flushed; every
                          finite input up to FLT_MAX is split exactly; an Inf or NaN activation / weight gives NaN in the
                          outputs it reaches, where LAV_CONV_F32 and the reference's cuDNN path propagate Inf as Inf);
                          0 = the library default (environment LAV_CONV_PRECISION = f32 | bf16x6, default bf16x6).  The
                          packed weights of a layer depend on it: pack and run with the same descriptor.  Launch note: a
                          workgroup of the split-operand kernel claims its CU's whole 160 KB of LDS whatever its tiles need
                          (one workgroup per CU either way), so that no kernel that uses LDS runs beside it on a CU - such
                          neighbours were measured to compute wrong results on gfx950 (DESIGN 4.4c;
                          LAV_SPLIT_LDS_EXCLUSIVE=0 restores the exact size for experiments) */
} lav_conv;
#define LAV_CONV_F32 1
#define LAV_CONV_BF16X6 2
#define LAV_CONV_F16X3 3   /* as BF16X6, and wherever the plan is the split kernel (round 5: the head convolution's plan only; round 6:
                              every split plan - any stride, tile, split-K, tap pairs, the classes of a transposed convolution) each
                              operand is TWO fp16 pieces scaled by a power of two taken from the tensor's largest finite magnitude,
                              three products: half the matrix instructions at 22 bits per operand - the error of the dot product
                              stays at the level of its fp32 accumulation.  The activations' magnitude comes from the launches that
                              wrote them (lav_conv2d_amax below) or, without that, from one measuring launch in front of the
                              convolution.  Re-packing on the device: lav_conv_repack_scratch (the weights' magnitude is measured first) */

/* output spatial size of the convolution */
int lav_conv_out_hw(const lav_conv *c, int *oh, int *ow);
/* number of floats of the packed weight buffer */
size_t lav_conv_packed_weight_floats(const lav_conv *c);
/* host-side repack of a PyTorch-layout weight (Conv2d: [cout][cin][kh][kw]; ConvTranspose2d:
 * [cin][cout][kh][kw]) into the kernel's layout [class][cout block of 32][tap][8-channel group][lane][channel pair] -
 * the order in which the MFMA A operands are consumed, so both kernels fetch 16 bytes per lane.  Pure host code. */
int lav_conv_pack_weights(const lav_conv *c, const float *h_weight, float *h_packed);
/* Re-packing on the device, for callers whose weights change in HBM (a training run that evaluates its student through these
 * kernels after every optimiser step, lav/lav_final_v2.py:228-236): lav_conv_pack_map (host, once per layer) fills
 * lav_conv_pack_map_ints(c) ints - for every slot of the packed buffer the index of the PyTorch-layout weight it holds, -1 for
 * padding; lav_conv_repack gathers (and, for LAV_CONV_BF16X6, splits) the packed buffer from the device-resident weight with
 * that map in one launch: bit-identical to lav_conv_pack_weights + upload.  lav_bn_fold: the eval-mode BatchNorm affine
 * scale = gamma / sqrt(var + eps), shift = beta - mean * scale (float64 arithmetic) of n channels, on the device. */
size_t lav_conv_pack_map_ints(const lav_conv *c);
int lav_conv_pack_map(const lav_conv *c, int *h_map);
int lav_conv_repack(const lav_conv *c, const float *d_weight, const int *d_map, float *d_packed, void *stream);
/* the same with 512 floats of device scratch (round 6): what a LAV_CONV_F16X3 layer needs - its fp16 section is scaled by the weights'
 * largest magnitude, which is measured on the device first (lav_conv_repack refuses such a layer) */
int lav_conv_repack_scratch(const lav_conv *c, const float *d_weight, const int *d_map, float *d_packed, float *d_parts, size_t parts_floats,
                            void *stream);
int lav_bn_fold(const float *mean, const float *var, const float *gamma, const float *beta, double eps, int n, float *scale,
                float *shift, void *stream);
/* introspection of the launch plan (host only, no device access): info[0..8] = { MP, MC, row-blocked tiles,
 * staged tile width, staged tile rows, LDS bytes, split-K factor, taps per weight slab, chunks per stage }; small layers
 * that take the direct kernel (whole batch in one GEMM dimension, operands straight from L2, no LDS staging) report
 * info[0] = 0 and info[1] = waves per workgroup.  Fails exactly when lav_conv2d would reject the shape. */
int lav_conv_tile_info(const lav_conv *c, int *info);
/* scratch for the split-K partial sums of small-output / deep-channel layers (0 when the layer is not split) */
size_t lav_conv_workspace_bytes(const lav_conv *c);
/* x, y, w_packed: device.  bias / scale / shift: [cout] device or NULL.  residual: same shape and
 * channel window as y, or NULL. */
int lav_conv2d(const lav_conv *c, const float *x, const float *w_packed, const float *bias, const float *scale,
               const float *shift, const float *residual, float *y, void *workspace, size_t workspace_bytes,
               void *stream);
/* lav_conv2d with the scale hand-off of LAV_CONV_F16X3 (round 6; replaces nothing in the reference - its cuDNN layers compute in
 * fp32 - it is what lets a CHAIN of layers, team_code_v2/models/lidar.py:110-143, lav/models/resnet.py:41-85, run on fp16 pieces
 * without a pass over the activations per layer).  amax_out (device, lav_conv_amax_count(c) floats, or NULL): the launch leaves
 * the largest finite |y| of each of its workgroups there (from its epilogue; from the split-K reduce pass; for the few plans with
 * neither, measured by one more launch) - rows beyond lav_batch_limit contribute 0, nothing needs zeroing.  amax_in / amax_in_count
 * (device, or NULL / 0): such maxima of the tensor(s) x was assembled from - any values whose maximum bounds max |x| (e.g. the
 * maxima of a tensor that x is a max-pooling, a crop or a bilinear resampling of); a LAV_CONV_F16X3 layer on the split kernel takes
 * its power-of-two scale from them instead of measuring x (other layers ignore them).  A bound that is too small would overflow
 * fp16 (Inf / NaN in y); one that is 2^k too large costs k of the 22 operand bits. */
#define LAV_AMAX_PARTS 512   /* floats lav_absmax_parts writes */
int lav_absmax_parts(const float *x, long n, float *parts, void *stream);   /* maxima of the finite |x[0 .. n)| in LAV_AMAX_PARTS parts: one launch */
int lav_conv_amax_count(const lav_conv *c);
int lav_conv2d_amax(const lav_conv *c, const float *x, const float *w_packed, const float *bias, const float *scale,
                    const float *shift, const float *residual, float *y, void *workspace, size_t workspace_bytes,
                    const float *amax_in, int amax_in_count, float *amax_out, void *stream);

/* Grouped ConvTranspose2d with few output channels (memory bound, vector-ALU kernel): group g maps input channels
 * [g*cin/groups, (g+1)*cin/groups) to cout_per_group[g] (1..8) output channels; y is [batch][sum cout][oh][ow] with the
 * groups' channels in order.  weight: the groups' PyTorch-layout ConvTranspose2d weights [cin/groups][cout_g][k][k]
 * concatenated (device).  bias: [sum cout] device or NULL.  sigmoid_from: output channels >= it get a sigmoid, -1 = none,
 * -2 = softmax over each group's channels (the class probabilities the agent takes from ERFNet, lav_agent_fast.py:264).
 * (kernel, stride) = (3, 2) or (2, 2).  cout_per_group is a host array.  Replaces the four head tails
 * ConvTranspose2d(64, out, 3, 2, 1, 1) of team_code_v2/models/lidar.py:30-33 (one launch for all heads) and ERFNet's
 * output layer lav/models/erfnet.py:137. */
int lav_deconv_grouped(int batch, int cin, int h, int w, int groups, const int *cout_per_group, int kernel, int stride,
                       int pad, int out_pad, const float *x, const float *weight, const float *bias, int sigmoid_from,
                       float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * 5. Rotated crop of the BEV feature map around each actor.  Replaces crop_feature
 *    (team_code_v2/model_inference.py:204-238; same maths team_code_v2/models/uniplanner.py:310-352):
 *    theta = k*R(ori) about the (offset_x, offset_y) pivot, k = crop/H, affine_grid + bilinear grid_sample,
 *    zeros outside, align_corners=True.
 *    feat [feat_batch][C][H][W] (feat_batch = 1: the one map is shared by all n crops, the reference's
 *    features.expand(N, ...)); locs [n][2] metres, oris [n] radians (device); out [n][C][crop][crop].
 * ------------------------------------------------------------------------------------------ */
int lav_crop_rotate(const float *feat, int feat_batch, int C, int H, int W, const float *locs, const float *oris, int n,
                    float pixels_per_meter, int crop, float offset_x, float offset_y, float *out, void *stream);
/* Training form (lav/models/uniplanner.py:84-95, bev_planner_v2.py:91-104): crop n is taken from map map_index[n]
 * (int32, device) of feat [num_maps][C][H][W] - the reference materialises features.expand(N, ...)[mask], one 39 MB
 * copy of the feature map per sampled vehicle, before cropping.  The backward spreads grad_out [n][C][crop][crop] over
 * the source pixels of grad_feat [num_maps][C][H][W] (zeroed by the call; fp32 atomics, like torch's grid_sampler). */
int lav_crop_rotate_indexed(const float *feat, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                            const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                            float *out, void *stream);
int lav_crop_rotate_backward(const float *grad_out, int num_maps, const int *map_index, int C, int H, int W, const float *locs,
                             const float *oris, int n, float pixels_per_meter, int crop, float offset_x, float offset_y,
                             float *grad_feat, void *stream);

/* ------------------------------------------------------------------------------------------
 * 6. Per-frame glue of LAVAgent.run_step between the big kernels (one launch each).
 *
 * lav_merge_ticks: cur = cat([tick, prev]) with the ego-vehicle box removed, then prev = tick
 *    (team_code_v2/lav_agent_fast.py:240-247 and preprocess() :450-452).  Removed points are marked x = NaN
 *    instead of being compacted: they fail every range test downstream exactly like absent points, and the
 *    row count stays static.  tick, prev [rows][4], cur [2*rows][4] (device).
 *
 * lav_stack_sweeps: get_stacked_lidar() (lav_agent_fast.py:363-383) with move_lidar_points() (:547-565).
 *    fused [rows][8] is the newly painted sweep; ring [slots][rows][8] the history; d_slot (1 int64, device):
 *    ring slot the new sweep is stored to; d_sweeps [num_sweeps] int64 (device): ring slots of sweeps t, t-5, t-10
 *    (d_sweeps[0] is ignored: sweep 0 is `fused`); d_R [num_sweeps][3][3], d_t [num_sweeps][3] (device): xyz' = xyz@R + t.
 *    out [num_sweeps*rows][8 + num_sweeps] = (xyz', features, one-hot sweep index).
 *
 * lav_extract_peaks: extract_peak() (team_code_v2/model_inference.py:189-202: 7x7 max-pool NMS, top-15 by
 *    score) over `ncls` heat-map planes [ncls][h][w] (apply_sigmoid: logits in, scores out) plus the gathers of
 *    det_inference() (:95-121) from size [size_c][h][w] and ori [ori_c][h][w].
 *    out [ncls][max_det][3 + size_c + ori_c] = (score, x, y, size..., ori...), rows by descending score, ties by
 *    ascending pixel index; if a plane has fewer than max_det non-suppressed pixels the remaining rows carry
 *    score -1e5.  The last 256 bytes of the workspace are counters: zero before the first use, left zero.
 * ------------------------------------------------------------------------------------------ */
int lav_merge_ticks(const float *tick, float *prev, int rows, int dim, float *cur, void *stream);
int lav_stack_sweeps(const float *fused, float *ring, const long *d_slot, const long *d_sweeps, const float *d_R,
                     const float *d_t, int num_sweeps, int rows, int dim, float *out, void *stream);
size_t lav_extract_peaks_workspace_bytes(int ncls, int h, int w);
int lav_extract_peaks(const float *heat, int ncls, int h, int w, int ks, int max_det, int apply_sigmoid,
                      const float *size, int size_c, const float *ori, int ori_c, float *out, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * 6b. Single-query attention pooling of the brake net.  Replaces Attention.forward, lav/models/attention.py:21-38
 *     (called twice per frame from RGBBrakePredictionModel.forward, team_code_v2/models/rgb.py:69-70): Linear(C -> 2C) key/value
 *     projection, positional encoding on the keys, one learned query per head, soft-max over the h*w tokens, weighted sum.
 *     With one fixed query per head the key projection folds into the query and the value projection moves behind the
 *     weighted sum (exact algebra, prepared once from the weights on the host):
 *       u         [heads][C]  = scale * W_k,h^T q_h
 *       dots_bias [heads][N]  = scale * q_h . (b_k,h + PE[n])
 *       out[b][h*dh + d]      = W_v[h*dh + d] . (sum_n softmax_n(u_h . x[b][:, n] + dots_bias[h][n]) x[b][:, n]) + b_v[h*dh + d]
 *     x [batch][C][N] (an NCHW map, N = h*w tokens), w_v [C][C] (rows = outputs, PyTorch Linear layout), out [batch][C].
 * ------------------------------------------------------------------------------------------ */
int lav_attn_pool(const float *x, int batch, int C, int N, int heads, const float *u, const float *dots_bias,
                  const float *w_v, const float *b_v, float *out, void *stream);

/*
 * Health counter of a frame: counter2[0] += how many of the n (<= 8) float tensors hold a NaN or an Inf, counter2[1] += 1.
 * Sticky and device resident: enqueued behind a frame's last kernels (it is part of the HIP graphs), read by the host when it
 * likes.  The reference checks its waypoints on the host every tick (lav_agent_fast.py:325-328, np.isnan after .cpu()); this is
 * the same question asked without a device->host copy per tensor, so that a benchmark loop can prove every timed frame finite.
 */
int lav_nonfinite_count(int n, const float *const *tensors, const long *numel, int *counter2, void *stream);

/* Frame glue that replaces library launches inside the frame graphs:
 * lav_maxpool3x3s2: nn.MaxPool2d(3, 2, 1) of the ResNet stems (lav/models/resnet.py:161,236), x [batch][channels][h][w] ->
 *     y [batch][channels][(h-1)/2+1][(w-1)/2+1]; honours lav_batch_limit.
 * lav_channel_affine: y = x * scale[c] + shift[c] on [batch][channels][plane] (plane % 4 == 0): the brake net's
 *     `normalize(rgb / 255)` (team_code_v2/models/rgb.py:71-72) in one pass.
 * lav_copy_many: up to 8 device-to-device copies (16-byte aligned pointers, sizes multiples of 4) in ONE launch: a tick's sensor tensors
 *     into the static buffers the frame graphs read (lav_agent_fast.py:233-277 hands them over as fresh tensors). */
int lav_maxpool3x3s2(const float *x, int batch, int channels, int h, int w, float *y, void *stream);
int lav_channel_affine(const float *x, int batch, int channels, long plane, const float *scale, const float *shift, float *y, void *stream);
int lav_copy_many(int n, const void *const *src, void *const *dst, const size_t *bytes, void *stream);
/* lav_stage_many: up to 8 strided tensors of at most 4 dimensions (float32, or uint8 converted to float32) -> contiguous float32
 *     buffers in ONE launch.  dims[4 n] (outermost first, leading 1s), strides[4 n] in source elements.  The camera tensors
 *     of a tick are channels-last views (lav_agent_fast.py:252-277: stack / permute / float). */
int lav_stage_many(int n, const void *const *src, float *const *dst, const int *dims, const long *strides, const int *src_is_u8,
                   void *stream);
/* lav_stage_many_block (ABI 28): the same launch, with up to 256 bytes of HOST data (a multiple of 4) riding in its kernel arguments and
 *     written to block_dst in HBM by the kernel: the per-tick sweep indices and float32 poses that get_stacked_lidar computes on the host
 *     (lav_agent_fast.py:363-383) - an upload of its own would be one more packet in front of the frame's first graph. */
int lav_stage_many_block(int n, const void *const *src, float *const *dst, const int *dims, const long *strides, const int *src_is_u8,
                         const void *block, int block_bytes, void *block_dst, void *stream);

/* Small dense layer out[b][o] = act(bias[o] + sum_k weight[o][k] x[b][k]) (weight in nn.Linear layout [out][in], bias or NULL;
 * act 0 = none, 1 = sigmoid): the brake classifier nn.Sequential(Linear(1024, 1), Sigmoid) of team_code_v2/models/rgb.py:62,79. */
int lav_linear_act(const float *x, int batch, int in_features, const float *weight, const float *bias, int out_features,
                   int act, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * 6c. Fixed-capacity, device-resident "other vehicles" batch.  The reference reads every detection back
 *     to the host (team_code_v2/model_inference.py:101-108: up to ~120 scalar device->host syncs) and runs the others
 *     branch on a data-dependent batch (:123-187).  Here the count stays in HBM:
 *
 * lav_det_decode: from the peak rows of lav_extract_peaks ([ncls][max_det][7] = score, x, y, size0, size1, ori0, ori1) of
 *     class `cls`, keep a row when score > min_score, near_px < |(x,y) - (ego_x,ego_y)| < far_px and not
 *     max(size0,size1) < min_box (det_inference, model_inference.py:95-121), and |(x,y) - (cx,cy)| > skip_px (the ego's own
 *     box, :131-133); survivors, in score order, give actors[2 i], actors[2 i + 1] = ((x - cx)/ppm, (y - cy)/ppm) (ego-frame
 *     metres) and actors[2 max_det + i] = atan2(ori1, ori0); *n_out = their number.  Entries beyond it are zero.
 * lav_batch_limit: from now on the batch-aware launches enqueued BY THIS THREAD - lav_conv2d, lav_crop_rotate, lav_gru_cast -
 *     skip the rows (images, crops, samples) >= *d_rows, read on the device when the kernel runs: a capacity-sized batch
 *     costs what its live rows cost, with no host round trip in between.  Skipped rows of the outputs are left untouched.
 *     NULL switches it off.  (A pointer, not a value: the launches can be captured in a HIP graph.)
 * ------------------------------------------------------------------------------------------ */
int lav_det_decode(const float *rows, int ncls, int max_det, int cls, double min_score, double ego_x, double ego_y,
                   double near_px, double far_px, double min_box, double cx, double cy, double skip_px, double ppm,
                   float *actors, int *n_out, void *stream);
int lav_batch_limit(const int *d_rows);
/* lav_det_decode_report (ABI 28): lav_det_decode, and the same launch also writes the [ncls][max_det][7] peak rows to host_rows, the
 *     count to host_n and then increments *host_seq (system-scope release) - three pointers into pinned, device-mapped HOST memory
 *     (hipHostMalloc).  The reference reads every detection back to the host (model_inference.py:101-108); here the host polls
 *     host_seq instead of fetching rows and count with two device->host copies and an event between the heads and the others
 *     graph.  All three NULL = lav_det_decode. */
int lav_det_decode_report(const float *rows, int ncls, int max_det, int cls, double min_score, double ego_x, double ego_y,
                          double near_px, double far_px, double min_box, double cx, double cy, double skip_px, double ppm,
                          float *actors, int *n_out, float *host_rows, int *host_n, unsigned *host_seq, void *stream);

/* ------------------------------------------------------------------------------------------
 * 7. Training-side pillar ops: what PointPillarNet needs in train mode, where BatchNorm1d uses batch
 *    statistics and therefore cannot be folded into the fused inference kernel.  They replace the two
 *    torch_scatter calls and coords.unique(dim=0) of lav/models/point_pillar.py:55-68, 81-85, 33 and give
 *    autograd what it needs (arg-max for the backward of scatter_max).
 *
 * lav_pillar_decorate: grid_locations + pillar_generation + decorate (point_pillar.py:55-85).
 *    points [batch][max_points][D]; outputs sized for the worst case (batch*max_points rows):
 *    unique_coords [P][3] = (cloud, xi, yi) sorted like torch.unique(dim=0); inverse [N_kept] pillar of each kept
 *    point; kept_src [N_kept] flat index (cloud*max_points + row) of each kept point - kept points stay in input
 *    order; decorated [N_kept][D+5] = (point, xyz - pillar mean, x - cell_x, y - cell_y) with the reference's swapped
 *    cell origin; counts [2] (device) = {P, N_kept}.  Pillar means are the same order-independent fixed-point sums as
 *    in lav_pillar_scatter.  Workspace: lav_pillar_decorate_workspace_bytes (no at-rest state; a lav_pillar_scatter workspace of
 *    the same geometry also serves and stays valid).  Any output except decorated/counts may be NULL.
 *
 * lav_scatter_max: out[s][c] = max over rows i with index[i] == s of src[i][c]; argmax[s][c] = the lowest such row
 *    (n for an empty segment, whose out is 0 - torch_scatter's convention).  src [n][channels], index [n] int32.
 * lav_scatter_max_backward: grad_src[argmax[s][c]][c] = grad_out[s][c], zero elsewhere.
 * ------------------------------------------------------------------------------------------ */
size_t lav_pillar_decorate_workspace_bytes(int batch, int max_points, const lav_grid *grid);
int lav_pillar_decorate(const float *points, const int *h_num_points, int batch, int max_points, int D,
                        const lav_grid *grid, int *unique_coords, int *inverse, int *kept_src, float *decorated,
                        int *counts, void *workspace, size_t workspace_bytes, void *stream);
int lav_scatter_max(const float *src, const int *index, int n, int channels, int num_segments, float *out, int *argmax,
                    void *stream);
int lav_scatter_max_backward(const float *grad_out, const int *argmax, int n, int channels, int num_segments,
                             float *grad_src, void *stream);

/* ------------------------------------------------------------------------------------------
 * 7b. Train-mode BatchNorm2d (batch statistics) with the neighbouring ReLU / residual add fused, forward and backward, NCHW
 *     float32.  Replaces the nn.BatchNorm2d + nn.ReLU (+ add) launches of the student networks in
 *     lav/lav_final_v2.py:140-259: ConvBackbone's Conv -> ReLU -> BatchNorm (team_code_v2/models/lidar.py:57-108; relu_pre)
 *     and ResNet-18's Conv -> BatchNorm (-> + identity) -> ReLU (lav/models/resnet.py; relu_post, residual).
 *
 *     forward:   t = relu_pre ? max(x, 0) : x;   mean/var over (batch, plane) per channel (biased var, float64 sums);
 *                y = (t - mean) / sqrt(var + eps) * gamma + beta (+ residual) (-> ReLU if relu_post)
 *                save_mean / save_var (biased) / save_rstd [channels] are written for the backward and the running statistics
 *                (the caller updates those: running = (1 - m) * running + m * (mean, var * n / (n - 1))).
 *     backward:  g = relu_post ? dy * [y > 0] : dy  (written to dres when the forward had a residual: that branch's gradient);
 *                dbeta = sum g, dgamma = sum g * xhat, dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n) (* [x > 0] if relu_pre)
 *     relu_pre excludes relu_post / residual.  y is only read when relu_post.  Two launches each way; sums are reduced in a
 *     fixed order (bit-reproducible).  workspace: lav_bn_train_workspace_bytes(channels) bytes, no at-rest state.
 * ------------------------------------------------------------------------------------------ */
size_t lav_bn_train_workspace_bytes(int channels);
int lav_bn_train_forward(const float *x, const float *residual, float *y, int batch, int channels, long plane, const float *gamma,
                         const float *beta, double eps, int relu_pre, int relu_post, float *save_mean, float *save_var,
                         float *save_rstd, void *workspace, size_t workspace_bytes, void *stream);
int lav_bn_train_backward(const float *x, const float *y, const float *dy, int batch, int channels, long plane, const float *gamma,
                          const float *save_mean, const float *save_rstd, int relu_pre, int relu_post, float *dx, float *dres,
                          float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream);
/* The same pair leaving the BOUND of what it writes (round 6): amax_y / amax_dx (device, lav_bn_train_amax_count floats, or NULL) receive
 * the largest finite |y| resp. |dx| of every workgroup of the normalising launch - in the training graph (lav/lav_final_v2.py:140-259)
 * the next convolution's input is a BatchNorm's output and a convolution's output gradient is a BatchNorm's input gradient, so that a
 * LAV_CONV_F16X3 step takes its scales from these parts (lav_conv2d_amax, lav_conv_wgrad_amax) instead of measuring every activation and
 * gradient tensor with a launch of its own (173 lav_absmax_parts launches per train_full step). */
int lav_bn_train_amax_count(int batch, int channels, long plane);
int lav_bn_train_forward_amax(const float *x, const float *residual, float *y, int batch, int channels, long plane, const float *gamma,
                              const float *beta, double eps, int relu_pre, int relu_post, float *save_mean, float *save_var,
                              float *save_rstd, float *amax_y, void *workspace, size_t workspace_bytes, void *stream);
int lav_bn_train_backward_amax(const float *x, const float *y, const float *dy, int batch, int channels, long plane, const float *gamma,
                               const float *save_mean, const float *save_rstd, int relu_pre, int relu_post, float *dx, float *dres,
                               float *dgamma, float *dbeta, float *amax_dx, void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradient of a 3x3, padding-1 convolution of stride 1 or 2, or of a 7x7, padding-3, stride-2 convolution (round 5): the weight half of torch.autograd's convolution backward for
 * the stage convolutions of ConvBackbone (team_code_v2/models/lidar.py:57-108) and the four heads' first convolutions (lidar.py:147-161)
 * inside LAV.train_lidar's backward (lav/lav_final_v2.py:140-259), which the reference runs on cuDNN (here: MIOpen's igemm_wrw).
 *     dw[co][ci][ky][kx] = sum over (n, y, x) of dy[n][co][y][x] * x[n][ci][stride y + ky - pad][stride x + kx - pad]      (zero padding, pad = ksize / 2)
 * x [batch][cin][h][w], dy [batch][cout][h / stride][w / stride], dw [cout][cin][ksize][ksize] (PyTorch layouts, float32, device, 16-byte
 * aligned); cin and cout multiples of 64, w a multiple of 4 (stride 2: h even, w a multiple of 8).  bf16x6 arithmetic on the matrix cores (operands split exactly into three bf16
 * pieces, six partial products, float32 accumulation), partial sums added in a fixed order: bit-reproducible.  The forward and the
 * data gradient of the same layers are lav_conv2d launches (the data gradient as the transposed convolution with the same weight
 * tensor); lav_amd/train/hipnn.py wires the three into one torch.autograd.Function.
 * workspace: lav_conv_wgrad_workspace_bytes (0 = shape not supported), private to the stream. */
size_t lav_conv_wgrad_workspace_bytes(int batch, int cin, int cout, int h, int w, int ksize, int stride);
int lav_conv_wgrad(const float *x, const float *dy, int batch, int cin, int cout, int h, int w, int ksize, int stride, float *dw,
                   void *workspace, size_t workspace_bytes, void *stream);
/* the same on TWO fp16 pieces per operand and three partial products (round 6, the weight-gradient side of LAV_CONV_F16X3): amax_x /
 * amax_dy (device, n_amax_* floats each) hold maxima of the finite |x| / |dy| in parts - lav_absmax_parts, or what lav_conv2d_amax
 * left - from which every task takes the two power-of-two scales; NULL, NULL: lav_conv_wgrad's bf16x6 arithmetic. */
int lav_conv_wgrad_amax(const float *x, const float *dy, int batch, int cin, int cout, int h, int w, int ksize, int stride, float *dw,
                        void *workspace, size_t workspace_bytes, const float *amax_x, int n_amax_x, const float *amax_dy, int n_amax_dy,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * 7c. Transposed convolutions with kernel == stride ("pointwise" up-convolutions, ABI 28): every output pixel has one contributing
 *     input pixel and tap, so the layer is k*k independent [cout x cin] x [cin x pixels] products - two of the BEV backbone's three
 *     up-convolutions (team_code_v2/models/lidar.py:114-131: ConvTranspose2d(64, 128, 1, 1) and ConvTranspose2d(128, 128, 4, 4, 1, 2),
 *     each + ReLU + eval BatchNorm into a slice of the 384-channel feature map):
 *     y[n][out_c_offset + co][k iy + ky - pad][k ix + kx - pad] = scale[co] * relu?(sum_ci w[ci][co][ky][kx] x[n][ci][iy][ix]) + shift[co];
 *     output positions without a contributing input pixel (output_padding) hold scale * relu?(0) + shift.  Exact fp32 products and
 *     accumulation on v_mfma_f32_32x32x2_f32.  Supported: k in {1, 4}, cin in {64, 128}, cout a multiple of 128 (k = 1) / 32 (k = 4),
 *     pad < k; k = 1 without padding.  x [batch][cin][ih][iw], y [batch][out_c_total][OH][OW], OH = (ih - 1) k - 2 pad + k + out_pad.
 *     lav_upconv_pointwise_pack repacks the PyTorch ConvTranspose2d weight [cin][cout][k][k] on the host
 *     (lav_upconv_pointwise_packed_floats floats).  amax_out (or NULL): lav_upconv_pointwise_parts floats, the largest finite |y| each
 *     workgroup wrote - what lav_conv2d_amax leaves for LAV_CONV_F16X3 readers of the feature map.
 * ------------------------------------------------------------------------------------------ */
size_t lav_upconv_pointwise_packed_floats(int cin, int cout, int k);
int lav_upconv_pointwise_pack(int cin, int cout, int k, const float *h_weight, float *h_packed);
int lav_upconv_pointwise_parts(int batch, int cin, int cout, int ih, int iw, int k, int pad, int out_pad);
int lav_upconv_pointwise(int batch, int cin, int cout, int ih, int iw, int k, int pad, int out_pad, const float *x, const float *w_packed,
                         const float *scale, const float *shift, int relu_pre, int out_c_total, int out_c_offset, float *y,
                         float *amax_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * 8. Two stacked 1-D convolutions in one launch - one half of ERFNet's non_bottleneck_1d block
 *    (lav/models/erfnet.py:45-56):  y = relu?( (conv1x3_dB( relu( conv3x1_dA(x) + bias_a ) ) + bias_b) * scale + shift
 *    + residual ).  x, y, residual [batch][channels][h][w] (same shape), stride 1, "same" padding (pad = dilation).
 *    One workgroup per image row; the intermediate stays in LDS.  Supported: w in {32, 64, 128},
 *    channels a multiple of 16 with ceil(channels/32) <= 128/w (ERFNet: 16@128, 64@64, 128@32).
 *    Weights: lav_conv1d_pair_pack_weights repacks a PyTorch [cout][cin][3] tensor (the unit kernel dimension
 *    squeezed) on the host into all the forms the kernels read - exact fp32, three bf16 pieces, two scaled fp16 pieces with
 *    their scale and the largest L1 norm of a filter behind them (round 6) -; scale/shift/residual may be NULL.
 * ------------------------------------------------------------------------------------------ */
size_t lav_conv1d_pair_packed_weight_floats(int channels);
int lav_conv1d_pair_pack_weights(int channels, const float *h_weight, float *h_packed);
/*
 * lav_conv1d_pair_chain: a RUN of such pairs over one map - the non_bottleneck_1d blocks of an ERFNet stage
 * (lav/models/erfnet.py:25-62: 5 blocks at 64 channels, 8 dilated ones at 128, 2 + 2 in the decoder) - as ONE persistent launch:
 * a workgroup per image row walks all pairs; between two pairs only its two neighbour rows (y -+ d_a) travel through memory
 * (write-through stores + a per-row progress counter), its own row, the block residual and the next pair's first weights never
 * leave the CU.  residual[i] != 0: pair i adds the INPUT of pair i-1 (the block's input) before its ReLU; the run starts at a
 * block boundary.  out[i]: output buffer of pair i ([batch][channels][h][w] each, all distinct; out[npairs-1] is the result).
 * batch * h must not exceed the CU count (every row's workgroup waits for its neighbours'); three bf16 pieces per operand (lav_conv1d_pair_chain_f16: two fp16 pieces).  Waits are
 * bounded: a workgroup that gives up raises the launch's abort word (its peers poll it and stop waiting too), fills ITS row of
 * out[npairs-1] with NaN - a voided launch cannot be mistaken for a result: every row of the output is either the complete result or
 * NaN (round 5) - and raises a sticky counter: lav_conv1d_pair_chain_status copies {workgroups that gave up, launches}
 * since the workspace was zero-filled (it SYNCHRONISES `stream`; 0 = every result was valid).
 * workspace: lav_conv1d_pair_chain_workspace_bytes (sticky counters | one progress counter per row, capacity rounded up to 64 rows | the fp16
 * runs' 16 x capacity count-and-maximum words), ZERO before its first use, private to the stream.
 */
size_t lav_conv1d_pair_chain_workspace_bytes(int batch, int h);
size_t lav_conv1d_pair_chain_lds_bytes(int channels, int w, int d_b_max);
int lav_conv1d_pair_chain(int batch, int channels, int h, int w, int npairs, const int *d_a, const int *d_b, const int *residual,
                          const int *relu_post, const float *x, const float *const *wa_packed, const float *const *bias_a,
                          const float *const *wb_packed, const float *const *bias_b, const float *const *scale,
                          const float *const *shift, float *const *out, void *workspace, size_t workspace_bytes, void *stream);
/* lav_conv1d_pair_chain_f16 (ABI 28): the same run, same arguments and contract, on TWO power-of-two-scaled fp16 pieces per operand and
 * three matrix products instead of three bf16 pieces and six (LAV_CONV_F16X3 for ERFNet's runs: half the matrix instructions, 2/3 of the
 * weight bytes a row streams per pair).  The weights' scale comes from lav_conv1d_pair_pack_weights (which packs all three forms); the
 * activations' scale is derived per workgroup and pair from the largest finite magnitude of the three rows it multiplies (the
 * neighbours' maxima travel IN the hand-off's flag: one 64-bit word {pairs done | row maximum} per pair and row), the intermediate row's from a bound (largest input x largest L1 norm of a filter +
 * largest |bias|).  Error: that of an fp32 dot product (tests/test_gpu_conv.py, against float64); not bit-identical to the bf16 run. */
int lav_conv1d_pair_chain_f16(int batch, int channels, int h, int w, int npairs, const int *d_a, const int *d_b, const int *residual,
                              const int *relu_post, const float *x, const float *const *wa_packed, const float *const *bias_a,
                              const float *const *wb_packed, const float *const *bias_b, const float *const *scale,
                              const float *const *shift, float *const *out, void *workspace, size_t workspace_bytes, void *stream);
/* lav_conv1d_pair_chain_region (ABI 28): several runs that follow each other on one stream share ONE cleaning of their progress counters.
 * The runs THIS THREAD enqueues from now on keep their counters at rows [row_offset, row_offset + batch * h) of the workspace's counter
 * array (capacity: what lav_conv1d_pair_chain_workspace_bytes was asked for, rounded up to 64 rows) and, in front of the run, zero the
 * first clean_rows counters of the array (clean_rows > 0: the first run of a sequence, for all of them), nothing (0: a launch earlier on the
 * stream cleaned this run's rows - each region is good for ONE run per cleaning), or their own rows (-1 with row_offset 0: the default,
 * one small launch in front of every run).  ERFNet's four runs per frame: one cleaning launch instead of four (lav/models/erfnet.py:109-131).
 * A pointer-free thread-local setting like lav_batch_limit: captured launches keep what was set when they were enqueued. */
int lav_conv1d_pair_chain_region(int row_offset, int clean_rows);
int lav_conv1d_pair_chain_status(const void *workspace, int *h_timeouts_launches2, void *stream);
size_t lav_conv1d_pair_lds_bytes(int channels, int w, int d_b);
int lav_conv1d_pair(int batch, int channels, int h, int w, int d_a, int d_b, const float *x, const float *wa_packed,
                    const float *bias_a, const float *wb_packed, const float *bias_b, const float *scale, const float *shift,
                    const float *residual, int relu_post, float *y, void *stream);

/* 2x2/stride-2 max pooling + per-channel affine (+ ReLU) written into a channel window of y: the pooled branch of
 * ERFNet's DownsamplerBlock, cat([conv(x), pool(x)]) -> BatchNorm -> ReLU (lav/models/erfnet.py:9-23), whose
 * convolution branch is lav_conv2d writing the other window of the same tensor.
 * x [batch][channels][h][w] (h, w even); y [batch][out_c_total][h/2][w/2]. */
int lav_pool_affine(const float *x, int batch, int channels, int h, int w, const float *scale, const float *shift, int relu,
                    float *y, int out_c_total, int out_c_offset, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LAV_AMD_H */

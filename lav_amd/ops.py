"""Tensor-level wrappers over the C ABI (include/lav_amd.h).

torch is used here only as the owner of HBM buffers and of the HIP stream; every
function hands raw device pointers to liblav_amd.so and enqueues on torch's
current stream.  No function here has a torch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os as _os
import threading as _threading
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import Camera, Conv, Grid, PointNet, check

_workspaces: dict = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor in HBM (cuda/hip device), got {t.device}; lav_amd has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


_retired = []


def _workspace(key, nbytes: int, device) -> torch.Tensor:
    """Scratch buffer per (kind, device, stream): kernels enqueued on different streams may run concurrently
    (the frame graph forks the brake net and the ego branch onto side streams), so they never share scratch."""
    k = (key, device, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(k)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            # a captured HIP graph may have baked the old buffer's address into its kernel nodes: keep it alive (it stays
            # large enough for the launches that were captured with it) instead of returning it to the allocator
            _retired.append(ws)
        # zero-filled: kernels with arrival counters (lav_extract_peaks) expect zeros before their first launch
        ws = torch.zeros(max(nbytes, 256), dtype=torch.uint8, device=device)
        _workspaces[k] = ws
    return ws


def _drop_workspace(key, device):
    ws = _workspaces.pop((key, device, torch.cuda.current_stream().cuda_stream), None)
    if ws is not None:
        _retired.append(ws)   # (captured graphs may still hold its address)


# ------------------------------------------------------------------------------------------ pillar
def make_grid(min_x, max_x, min_y, max_y, ppm) -> Grid:
    nx = int((max_x - min_x) * ppm)
    ny = int((max_y - min_y) * ppm)
    return Grid(float(min_x), float(max_x), float(min_y), float(max_y), float(ppm), nx, ny)


def pillar_scatter(points: torch.Tensor, num_points: Sequence[int], grid: Grid, w1, b1, w2, b2,
                   want_indices: bool = False, amax: Optional["Amax"] = None):
    """points (B, Nmax, D) or (N, D) f32 in HBM -> canvas (B, C, ny, nx)
    [+ unique_coords (P,3) int32, inverse (N_kept,) int32 when want_indices].
    amax: an Amax that receives the canvas kernel's per-workgroup maxima (lav_pillar_scatter_amax) and is attached to the canvas
    (amax_of): the first BEV convolution then takes its fp16 scale from them instead of measuring the canvas with a launch."""
    lib = _lib.load()
    if points.dim() == 2:
        points = points[None]
    points = _f32c(points, "points")
    B, nmax, D = points.shape
    if len(num_points) != B:
        raise RuntimeError("num_points must have one entry per cloud")
    Cc = w2.shape[1]
    net = PointNet(_ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), D + 5, Cc)
    dev = points.device
    canvas = torch.empty((B, Cc, grid.ny, grid.nx), dtype=torch.float32, device=dev)
    nbytes = lib.lav_pillar_workspace_bytes(B, nmax, C.byref(grid))
    # zero-filled once and kept per geometry: the head of a pillar workspace is state that every call leaves clean
    # (include/lav_amd.h, workspace contract)
    ws = _workspace(("pillar", B, grid.nx, grid.ny), nbytes, dev)
    h_num = (C.c_int * B)(*[int(n) for n in num_points])
    uc = inv = cnt = None
    if want_indices:
        total = B * nmax
        uc = torch.empty((max(total, 1), 3), dtype=torch.int32, device=dev)
        inv = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
        cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    parts = amax.reset().take(lib.lav_pillar_amax_count(B, C.byref(grid))) if amax is not None else None
    rc = lib.lav_pillar_scatter_amax(_ptr(points) if nmax > 0 else None, h_num, B, nmax, D, C.byref(grid), C.byref(net),
                                     _ptr(canvas), _ptr(uc), _ptr(inv), _ptr(cnt), _ptr(parts), _ptr(ws), ws.numel(), _stream())
    if amax is not None:
        canvas._lav_amax = amax
    if rc != 0:
        # the workspace's zero-at-rest state (arrival counters, epoch words) may be half-updated: never reuse it (lav_amd.h, workspace
        # contract) - the next call zero-fills a fresh one
        _drop_workspace(("pillar", B, grid.nx, grid.ny), dev)
    check(rc, "lav_pillar_scatter")
    if want_indices:
        p, k = [int(v) for v in cnt.tolist()]
        return canvas, uc[:p], inv[:k]
    return canvas


def pillar_decorate(points: torch.Tensor, num_points: Sequence[int], grid: Grid):
    """Training-side front end of PointPillarNet: points (B, Nmax, D) -> decorated (N_kept, D+5), unique_coords (P,3),
    inverse (N_kept,), kept_src (N_kept,) (int32, all in HBM).  One device->host copy (the two counts)."""
    lib = _lib.load()
    if points.dim() == 2:
        points = points[None]
    points = _f32c(points.detach(), "points")
    B, nmax, D = points.shape
    dev = points.device
    total = max(B * nmax, 1)
    uc = torch.empty((total, 3), dtype=torch.int32, device=dev)
    inv = torch.empty((total,), dtype=torch.int32, device=dev)
    src = torch.empty((total,), dtype=torch.int32, device=dev)
    dec = torch.empty((total, D + 5), dtype=torch.float32, device=dev)
    cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbytes = lib.lav_pillar_decorate_workspace_bytes(B, nmax, C.byref(grid))
    ws = _workspace("pillar_decorate", nbytes, dev)
    h_num = (C.c_int * B)(*[int(n) for n in num_points])
    check(lib.lav_pillar_decorate(_ptr(points) if nmax > 0 else None, h_num, B, nmax, D, C.byref(grid), _ptr(uc), _ptr(inv),
                                  _ptr(src), _ptr(dec), _ptr(cnt), _ptr(ws), ws.numel(), _stream()), "lav_pillar_decorate")
    p, k = [int(v) for v in cnt.tolist()]
    return dec[:k], uc[:p], inv[:k], src[:k]


class _ScatterMax(torch.autograd.Function):
    """torch_scatter.scatter_max(src, index, dim=0) on liblav_amd, differentiable in src."""

    @staticmethod
    def forward(ctx, src, index, num_segments):
        lib = _lib.load()
        src = _f32c(src, "src")
        if index.dtype != torch.int32 or not index.is_cuda:
            raise RuntimeError("scatter_max: index must be an int32 tensor in HBM")
        n, ch = src.shape
        out = torch.empty((num_segments, ch), dtype=torch.float32, device=src.device)
        arg = torch.empty((num_segments, ch), dtype=torch.int32, device=src.device)
        check(lib.lav_scatter_max(_ptr(src), _ptr(index.contiguous()), n, ch, num_segments, _ptr(out), _ptr(arg), _stream()),
              "lav_scatter_max")
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        lib = _lib.load()
        grad_out = _f32c(grad_out, "grad_out")
        num_segments, ch = grad_out.shape
        grad_src = torch.empty((ctx.n, ch), dtype=torch.float32, device=grad_out.device)
        check(lib.lav_scatter_max_backward(_ptr(grad_out), _ptr(arg), ctx.n, ch, num_segments, _ptr(grad_src), _stream()),
              "lav_scatter_max_backward")
        return grad_src, None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, num_segments: int):
    """(out (S, C), argmax (S, C) int32) - torch_scatter.scatter_max(src, index, dim=0) semantics."""
    return _ScatterMax.apply(src, index, int(num_segments))


# ------------------------------------------------------------------------------------------ paint
def make_cameras(mats) -> C.Array:
    """mats: sequence of (K 3x3, lidar_to_world 4x4, world_to_cam 4x4) float32 arrays."""
    arr = (Camera * len(mats))()
    for cam, (K, l2w, w2c) in zip(arr, mats):
        cam.K[:] = [float(v) for v in np.asarray(K, np.float32).reshape(-1)]
        cam.l2w[:] = [float(v) for v in np.asarray(l2w, np.float32).reshape(-1)]
        cam.w2c[:] = [float(v) for v in np.asarray(w2c, np.float32).reshape(-1)]
    return arr


def paint(lidar: torch.Tensor, sem: torch.Tensor, cams, want_uvz: bool = False):
    """lidar (N, Dl) f32, sem (ncam, 1+Cs, H, W) softmax maps -> fused (N, Dl+Cs) [+ uvz (ncam, N, 3) int32]."""
    lib = _lib.load()
    lidar = _f32c(lidar, "lidar")
    sem = _f32c(sem, "sem")
    n, dl = lidar.shape
    ncam, cs1, h, w = sem.shape
    fused = torch.empty((n, dl + cs1 - 1), dtype=torch.float32, device=lidar.device)
    uvz = torch.empty((ncam, n, 3), dtype=torch.int32, device=lidar.device) if want_uvz else None
    check(lib.lav_paint(_ptr(lidar), n, dl, _ptr(sem), ncam, cs1 - 1, h, w, cams, _ptr(fused), _ptr(uvz), _stream()),
          "lav_paint")
    return (fused, uvz) if want_uvz else fused


# ------------------------------------------------------------------------------------------ GRU decoders
def gru_cast(embd, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, T: int):
    """embd (B, E); stacked per-command GRU/MLP weights -> (B, num_cmds, T, 2)."""
    lib = _lib.load()
    embd = _f32c(embd, "embd")
    B, E = embd.shape
    ncmd, g3, _ = w_ih.shape
    H = g3 // 3
    out = torch.empty((B, ncmd, T, 2), dtype=torch.float32, device=embd.device)
    check(lib.lav_gru_cast(_ptr(embd), B, E, H, ncmd, T, _ptr(w_ih), _ptr(w_hh), _ptr(b_ih), _ptr(b_hh), _ptr(mlp_w),
                           _ptr(mlp_b), _ptr(out), None, 0, _stream()), "lav_gru_cast")
    return out


def embed_cast(feat, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, T: int, cmd_w=None, cmd_b=None, oris=None, locs=None, want_embd=True):
    """feat (B, E, h, w) (the embedder's last map) or (B, E) -> (embd (B,E) or None, cast (B,num_cmds,T,2), cmds (B,num_cmds) or
    None): spatial mean, the six cast GRUs, the command scores and the rotation / translation of the waypoints into the ego
    frame in ONE launch (lav_embed_cast)."""
    lib = _lib.load()
    feat = _f32c(feat, "feat")
    B, E = feat.shape[0], feat.shape[1]
    hw = feat.numel() // max(B * E, 1) if B else 1
    ncmd, g3, _ = w_ih.shape
    dev = feat.device
    out = torch.empty((B, ncmd, T, 2), dtype=torch.float32, device=dev)
    embd = torch.empty((B, E), dtype=torch.float32, device=dev) if want_embd else None
    cmds = torch.empty((B, ncmd), dtype=torch.float32, device=dev) if cmd_w is not None else None
    check(lib.lav_embed_cast(_ptr(feat), B, E, hw, _ptr(embd), g3 // 3, ncmd, T, _ptr(w_ih), _ptr(w_hh), _ptr(b_ih), _ptr(b_hh),
                             _ptr(mlp_w), _ptr(mlp_b), _ptr(None if cmd_w is None else _f32c(cmd_w, "cmd_w")),
                             _ptr(None if cmd_b is None else _f32c(cmd_b, "cmd_b")), _ptr(cmds),
                             _ptr(None if oris is None else _f32c(oris.reshape(-1), "oris")),
                             _ptr(None if locs is None else _f32c(locs.reshape(-1, 2), "locs")), _ptr(out), _stream()), "lav_embed_cast")
    return embd, out, cmds


def gru_plan(embd, nxp, cast_locs, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, iters: int, cmd: int, ppm: float,
             crop_size: float, impl: str = "auto"):
    """embd (B,H), nxp (B,2), cast_locs (B,num_cmds,T,2) -> (B, iters, num_cmds or 1, T, 2).
    impl "auto": the persistent one-launch kernel when it applies (a launch that could not complete returns NaN and
    raises gru_plan_status); "steps": one launch per GRU step, no co-residency requirement."""
    lib = _lib.load()
    embd = _f32c(embd, "embd")
    nxp = _f32c(nxp, "nxp")
    cast_locs = _f32c(cast_locs, "cast_locs")
    B, H = embd.shape
    _, ncmd, T, _ = cast_locs.shape
    nc = 1 if cmd >= 0 else ncmd
    out = torch.empty((B, iters, nc, T, 2), dtype=torch.float32, device=embd.device)
    nbytes = lib.lav_gru_plan_workspace_bytes(B, H, ncmd, T)
    ws = _workspace("plan", nbytes, embd.device)
    fn = {"auto": lib.lav_gru_plan, "steps": lib.lav_gru_plan_steps}[impl]
    check(fn(_ptr(embd), _ptr(nxp), _ptr(cast_locs), B, H, ncmd, T, iters, cmd, float(ppm),
             float(crop_size), _ptr(w_ih), _ptr(w_hh), _ptr(b_ih), _ptr(b_hh), _ptr(mlp_w), _ptr(mlp_b),
             _ptr(out), _ptr(ws), ws.numel(), _stream()), "lav_gru_plan")
    return out


class _GruSeq(torch.autograd.Function):
    """GRU recurrence over a sequence on lav_gru_seq_forward / lav_gru_seq_backward (one launch per step, MFMA recurrent
    GEMM fused with the gates).  Differentiable in x (the input-side pre-activations), h0, w_hh and b_hh; the weight
    gradients are two GEMMs over the tape (rocBLAS through torch)."""

    @staticmethod
    def forward(ctx, x, h0, w_hh, b_hh, T):
        lib = _lib.load()
        x, h0 = _f32c(x, "x"), _f32c(h0, "h0")
        w_hh, b_hh = _f32c(w_hh, "w_hh"), _f32c(b_hh, "b_hh")
        R, H = h0.shape
        per_step = x.dim() == 3
        if (per_step and tuple(x.shape) != (R, T, 3 * H)) or (not per_step and tuple(x.shape) != (R, 3 * H)):
            raise RuntimeError(f"gru_seq: x has shape {tuple(x.shape)}, expected ({R}, {T}, {3 * H}) or ({R}, {3 * H})")
        if tuple(w_hh.shape) != (3 * H, H) or b_hh.numel() != 3 * H:
            raise RuntimeError("gru_seq: w_hh must be (3H, H) and b_hh (3H,)")
        out = torch.empty((R, T, H), dtype=torch.float32, device=x.device)
        need_tape = any(ctx.needs_input_grad[:4])
        tape = torch.empty((R, T, 4, H), dtype=torch.float32, device=x.device) if need_tape else None
        train_work["gru_seq_forward_flops"] += 2 * R * H * 3 * H * T
        check(lib.lav_gru_seq_forward(_ptr(x), int(per_step), _ptr(h0), _ptr(w_hh), _ptr(b_hh), R, T, H, _ptr(out),
                                      _ptr(tape) if need_tape else None, _stream()), "lav_gru_seq_forward")
        if need_tape:
            ctx.save_for_backward(tape, out, h0, w_hh)
            ctx.per_step = per_step
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        tape, out, h0, w_hh = ctx.saved_tensors
        R, T, H = out.shape
        dout = _f32c(dout, "dout")
        dx = torch.empty((R, T, 3 * H), dtype=torch.float32, device=out.device)
        dgh = torch.empty((R, T, 3 * H), dtype=torch.float32, device=out.device)
        dh0 = torch.empty((R, H), dtype=torch.float32, device=out.device)
        if R == 0:
            return (dx if ctx.per_step else dx.sum(1)), dh0, torch.zeros_like(w_hh), w_hh.new_zeros(3 * H), None
        ws = _workspace("gru_seq_bwd", lib.lav_gru_seq_backward_workspace_bytes(R, H), out.device)
        w_hh_t = w_hh.t().contiguous()
        train_work["gru_seq_backward_flops"] += 2 * R * 3 * H * H * T
        check(lib.lav_gru_seq_backward(_ptr(dout), _ptr(tape), _ptr(out), _ptr(h0), _ptr(w_hh_t), R, T, H, _ptr(dx), _ptr(dgh),
                                       _ptr(dh0), _ptr(ws), ws.numel(), _stream()), "lav_gru_seq_backward")
        h_prev = torch.cat([h0[:, None], out[:, :-1]], dim=1)                       # (R, T, H): the state each step started from
        dw_hh = dgh.view(R * T, 3 * H).t() @ h_prev.reshape(R * T, H)
        db_hh = dgh.sum(dim=(0, 1))
        return (dx if ctx.per_step else dx.sum(1)), dh0, dw_hh, db_hh, None


def gru_seq(x, h0, w_hh, b_hh, T: int):
    """h_t = GRUCell(x_t, h_{t-1}) for t < T with the input side already projected: x (R,T,3H) or (R,3H) (the same input at
    every step) = W_ih u + b_ih in torch's gate order (r, z, n); h0 (R,H) -> (R,T,H).  Same arithmetic as nn.GRU."""
    return _GruSeq.apply(x, h0, w_hh, b_hh, int(T))


def gru_plan_status(B: int, H: int, num_cmds: int, cmd: int, device, stream=None) -> int:
    """Status word of the last gru_plan launch on `stream` (default: the current one) with these sizes: 0 = completed,
    1 = the persistent kernel gave up waiting for its peers and returned NaN.  Synchronises that stream."""
    lib = _lib.load()
    st = stream if stream is not None else torch.cuda.current_stream()
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ws = _workspaces.get(("plan", device, st.cuda_stream))
    if ws is None:
        return 0
    status = C.c_int(0)
    check(lib.lav_gru_plan_status(_ptr(ws), ws.numel(), B, H, num_cmds, cmd, C.byref(status), st.cuda_stream), "lav_gru_plan_status")
    return int(status.value)


def gru_plan_diag(B: int, H: int, num_cmds: int, cmd: int, device, stream=None) -> dict:
    """The diagnosis words of the last persistent gru_plan launch on `stream` (include/lav_amd.h: lav_gru_plan_diag)."""
    lib = _lib.load()
    st = stream if stream is not None else torch.cuda.current_stream()
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ws = _workspaces.get(("plan", device, st.cuda_stream))
    names = ("status", "entered", "abort_wg_plus1", "abort_wave", "abort_epoch", "abort_spins", "missing_granule", "tag_seen",
             "abort_us", "completed", "aborted_launches", "launches")
    if ws is None:
        return dict.fromkeys(names, 0)
    words = (C.c_int * 16)()
    check(lib.lav_gru_plan_diag(_ptr(ws), ws.numel(), B, H, num_cmds, cmd, words, st.cuda_stream), "lav_gru_plan_diag")
    return dict(zip(names, list(words)))


# ------------------------------------------------------------------------------------------ conv
_precision_stack = _threading.local()


def infer_precision() -> int:
    """lav_conv.precision of the eval-mode engines built from now on by this thread: the innermost `with precision(...)`, else
    LAV_INFER_PRECISION (f16x3 | bf16x6 | f32, default: the library's, i.e. LAV_CONV_PRECISION / bf16x6).  The frame pipelines and
    InferModel ask for LAV_CONV_F16X3 (round 6: every split-kernel layer on two fp16 pieces, the scale handed from layer to layer);
    an engine a trainer builds by calling a module directly keeps the default (its layers are re-packed on the device after every
    step, which the fp16 packing does not support)."""
    st = getattr(_precision_stack, "v", None)
    if st:
        return st[-1]
    return {"f16x3": _lib.CONV_F16X3, "bf16x6": _lib.CONV_BF16X6, "f32": _lib.CONV_F32, "fp32": _lib.CONV_F32}.get(
        _os.environ.get("LAV_INFER_PRECISION", ""), 0)


class precision:
    """with ops.precision(_lib.CONV_F16X3): ... - the eval engines used inside are the ones packed for that precision (engines are
    cached per precision on their modules: nothing is written onto a module that somebody else owns)."""

    def __init__(self, p: int):
        self.p = int(p)

    def __enter__(self):
        st = getattr(_precision_stack, "v", None)
        if st is None:
            st = _precision_stack.v = []
        st.append(self.p)
        return self

    def __exit__(self, *exc):
        _precision_stack.v.pop()
        return False


def frame_precision() -> int:
    """What the inference pipelines run at: LAV_CONV_F16X3 unless LAV_CONV_PRECISION asks for the exact-fp32 kernels or
    LAV_INFER_PRECISION / LAV_HEADS_PRECISION (round 5's knob) say bf16x6."""
    if _os.environ.get("LAV_CONV_PRECISION", "") in ("f32", "fp32"):
        return 0
    want = _os.environ.get("LAV_INFER_PRECISION", _os.environ.get("LAV_HEADS_PRECISION", "f16x3"))
    return _lib.CONV_F16X3 if want == "f16x3" else 0


class Amax:
    """Maxima of the finite |values| of a tensor, in parts, as the launches that wrote it left them (lav_conv2d_amax): the next
    LAV_CONV_F16X3 layer takes its power-of-two activation scale from them instead of measuring its input.  One fixed buffer;
    `take(n)` hands out the next n slots (the same addresses on every pass over the same layers: HIP graphs may hold them)."""
    CAPACITY = 20480

    def __init__(self, device, capacity: Optional[int] = None, zeroed: bool = True):
        self.CAPACITY = int(capacity) if capacity else Amax.CAPACITY
        self.buf = (torch.zeros if zeroed else torch.empty)(self.CAPACITY, dtype=torch.float32, device=device)
        self.count = 0

    def reset(self):
        self.count = 0
        return self

    def take(self, n: int) -> torch.Tensor:
        if self.count + n > self.CAPACITY:
            raise RuntimeError(f"Amax: {self.count} + {n} parts exceed the buffer ({self.CAPACITY})")
        v = self.buf[self.count:self.count + n]
        self.count += n
        return v


def amax_of(t):
    """The Amax a producer attached to the tensor OBJECT it returned (engine-internal hand-off between modules: backbone -> heads /
    crops); None for anything else.  A bound, not a measurement: valid for the tensor and for whatever is a max-pool, a crop or a
    bilinear resampling of it."""
    return getattr(t, "_lav_amax", None)


class ConvLayer:
    """One fused convolution of the C ABI: packed weights + epilogue vectors resident in HBM.

    Built once from PyTorch-layout parameters (host-side repack in the library), then
    `__call__(x, out=None, residual=None)` enqueues lav_conv2d.
    """

    def __init__(self, weight: torch.Tensor, *, stride=1, padding=(0, 0), dilation=(1, 1), transposed=False,
                 output_padding=0, bias: Optional[torch.Tensor] = None, bn=None, bn_eps: float = 1e-5,
                 relu_pre=False, relu_post=False, sigmoid=False, in_c_total=None, in_c_offset=0, out_c_total=None, target_cus=0, pad_value=0.0,
                 out_c_offset=0, precision=0, device=None):
        lib = _lib.load()
        # the sources are kept by reference (tensors, or callables that assemble them): refresh() re-packs from their current values
        self._src = dict(weight=weight, bias=bias, bn=bn, bn_eps=float(bn_eps))
        self._map = None
        src_dev = None
        if callable(weight):
            weight = weight()
        if callable(bias):
            bias = bias()
        if callable(bn):
            bn = bn()
        src_dev = weight.device
        w = weight.detach().to("cpu", torch.float32).contiguous()
        if transposed:
            cin, cout, kh, kw = w.shape
        else:
            cout, cin, kh, kw = w.shape
        if isinstance(padding, int):
            padding = (padding, padding)
        if isinstance(dilation, int):
            dilation = (dilation, dilation)
        self.cin, self.cout = cin, cout
        self.in_c_total = in_c_total if in_c_total is not None else cin
        self.out_c_total = out_c_total if out_c_total is not None else cout
        self.desc = Conv(1, self.in_c_total, in_c_offset, cin, 0, 0, cout, kh, kw, int(stride), int(padding[0]),
                         int(padding[1]), int(dilation[0]), int(dilation[1]), int(bool(transposed)),
                         int(output_padding), self.out_c_total, out_c_offset, int(relu_pre), int(relu_post),
                         int(sigmoid), int(target_cus), float(pad_value), int(precision))
        probe = Conv.from_buffer_copy(self.desc)
        probe.h, probe.w = 64, 64
        nfl = lib.lav_conv_packed_weight_floats(C.byref(probe))
        if nfl == 0:
            raise RuntimeError("lav_conv_packed_weight_floats: " + lib.lav_last_error().decode())
        packed = torch.empty(nfl, dtype=torch.float32)
        check(lib.lav_conv_pack_weights(C.byref(probe), w.data_ptr(), packed.data_ptr()), "lav_conv_pack_weights")
        dev = device if device is not None else src_dev
        self.w = packed.to(dev)
        self.bias = None if bias is None else bias.detach().to(dev, torch.float32).contiguous()
        self._ws_bytes = {}
        self.scale = self.shift = None
        if bn is not None:  # eval-mode BatchNorm as y = x*scale + shift, folded in float64
            mean, var, gamma, beta = [t.detach().to("cpu", torch.float64) for t in bn]
            scale = gamma / torch.sqrt(var + bn_eps)
            self.scale = scale.to(torch.float32).to(dev)
            self.shift = (beta - mean * scale).to(torch.float32).to(dev)

    def refresh(self):
        """Re-pack the weights and re-fold the epilogue vectors from the CURRENT values of the tensors this layer was built from
        (parameters that an optimiser step or load_state_dict changed in place).  With everything resident in HBM this is one
        gather launch (lav_conv_repack through the layer's index map) plus lav_bn_fold - no host round trip, bit-identical to
        building the layer anew; anything else falls back to the host packer."""
        lib = _lib.load()
        src = self._src
        get = lambda v: v() if callable(v) else v
        w, bias, bn = get(src["weight"]), get(src["bias"]), get(src["bn"])
        dev = self.w.device
        probe = Conv.from_buffer_copy(self.desc)
        probe.h, probe.w = 64, 64
        if dev.type == "cuda" and w.device == dev and self._map is not False:
            if self._map is None:
                n = lib.lav_conv_pack_map_ints(C.byref(probe))
                m = torch.empty(max(n, 1), dtype=torch.int32)
                if n == 0 or lib.lav_conv_pack_map(C.byref(probe), m.data_ptr()) != 0:
                    self._map = False
                else:
                    self._map = m.to(dev)
        if dev.type == "cuda" and w.device == dev and self._map is not False and self._map is not None:
            wd = w.detach().to(torch.float32).contiguous()
            if self.desc.precision == _lib.CONV_F16X3:   # (the fp16 pieces are scaled by the weights' largest magnitude: measured into 512 floats of scratch)
                sc = _workspace("conv_repack", 512 * 4, dev)
                check(lib.lav_conv_repack_scratch(C.byref(probe), _ptr(wd), _ptr(self._map), _ptr(self.w), _ptr(sc), 512, _stream()), "lav_conv_repack_scratch")
            else:
                check(lib.lav_conv_repack(C.byref(probe), _ptr(wd), _ptr(self._map), _ptr(self.w), _stream()), "lav_conv_repack")
        else:
            packed = torch.empty(self.w.numel(), dtype=torch.float32)
            wh = w.detach().to("cpu", torch.float32).contiguous()
            check(lib.lav_conv_pack_weights(C.byref(probe), wh.data_ptr(), packed.data_ptr()), "lav_conv_pack_weights")
            self.w.copy_(packed)
        if self.bias is not None:
            self.bias.copy_(bias.detach())
        if bn is not None:
            if dev.type == "cuda" and all(t.device == dev for t in bn):
                mean, var, gamma, beta = [t.detach().to(torch.float32).contiguous() for t in bn]
                check(lib.lav_bn_fold(_ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), src["bn_eps"], mean.numel(), _ptr(self.scale),
                                      _ptr(self.shift), _stream()), "lav_bn_fold")
            else:
                mean, var, gamma, beta = [t.detach().to("cpu", torch.float64) for t in bn]
                scale = gamma / torch.sqrt(var + src["bn_eps"])
                self.scale.copy_(scale.to(torch.float32))
                self.shift.copy_((beta - mean * scale).to(torch.float32))

    @classmethod
    def from_module(cls, conv, bn=None, bn_slice=None, **kw):
        """Build from nn.Conv2d / nn.ConvTranspose2d (+ optional nn.BatchNorm2d, optionally a channel slice of it)."""
        import torch.nn as nn
        tr = isinstance(conv, nn.ConvTranspose2d)
        if conv.stride[0] != conv.stride[1]:
            raise RuntimeError("anisotropic stride unsupported")
        bnt, eps = None, 1e-5
        if bn is not None:
            sl = bn_slice if bn_slice is not None else slice(None)
            bnt = tuple(t[sl] for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias))
            eps = bn.eps
        return cls(conv.weight, stride=conv.stride[0], padding=tuple(conv.padding), dilation=tuple(conv.dilation),
                   transposed=tr, output_padding=conv.output_padding[0] if tr else 0, bias=conv.bias, bn=bnt, bn_eps=eps,
                   device=kw.pop("device", conv.weight.device), **kw)

    def out_hw(self, h: int, w: int):
        d = Conv.from_buffer_copy(self.desc)
        d.h, d.w = h, w
        oh, ow = C.c_int(), C.c_int()
        check(_lib.load().lav_conv_out_hw(C.byref(d), C.byref(oh), C.byref(ow)), "lav_conv_out_hw")
        return oh.value, ow.value

    def uses_amax(self, B: int, h: int, w: int) -> bool:
        """Whether this layer on a (B, ., h, w) input runs the fp16 three-product plan, i.e. has a use for its producers' maxima."""
        key = ("f16", B, h, w, _os.environ.get("LAV_CONV_SPLIT"), _os.environ.get("LAV_SPLIT_SK"))
        v = self._ws_bytes.get(key)
        if v is None:
            v = False
            if self.desc.precision == _lib.CONV_F16X3:
                d = Conv.from_buffer_copy(self.desc)
                d.batch, d.h, d.w = B, h, w
                info = (C.c_int * 9)()
                v = _lib.load().lav_conv_tile_info(C.byref(d), info) == 0 and info[0] == -1 and info[7] >= 200
            self._ws_bytes[key] = v
        return v

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 amax_in: Optional["Amax"] = None, amax_out: Optional["Amax"] = None):
        """amax_in: maxima that bound |x| (an Amax its producers filled); amax_out: an Amax this launch appends the maxima of |y| to."""
        x = _f32c(x, "x")
        B, ct, h, w = x.shape
        if ct != self.in_c_total:
            raise RuntimeError(f"conv input has {ct} channels, layer expects {self.in_c_total}")
        d = Conv.from_buffer_copy(self.desc)
        d.batch, d.h, d.w = B, h, w
        oh, ow = self.out_hw(h, w)
        if out is None:
            out = torch.empty((B, self.out_c_total, oh, ow), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (B, self.out_c_total, oh, ow) or not out.is_contiguous():
            raise RuntimeError(f"conv output buffer {tuple(out.shape)} != {(B, self.out_c_total, oh, ow)}")
        if residual is not None:
            residual = _f32c(residual, "residual")
            if residual.shape != out.shape:
                raise RuntimeError("residual shape mismatch")
        lib = _lib.load()
        key = (B, h, w, _os.environ.get("LAV_CONV_SPLIT"), _os.environ.get("LAV_SPLIT_SK"))   # lav_conv2d re-reads the plan knobs per call: the cached size follows them
        nbytes = self._ws_bytes.get(key)
        if nbytes is None:
            nbytes = self._ws_bytes[key] = lib.lav_conv_workspace_bytes(C.byref(d))
        ws = _workspace("conv", nbytes, x.device) if nbytes else None
        if amax_in is None and amax_out is None:
            check(lib.lav_conv2d(C.byref(d), _ptr(x), _ptr(self.w), _ptr(self.bias), _ptr(self.scale), _ptr(self.shift),
                                 _ptr(residual), _ptr(out), _ptr(ws), ws.numel() if ws is not None else 0, _stream()),
                  "lav_conv2d")
            return out
        a_in = amax_in.buf if amax_in is not None and amax_in.count > 0 else None
        a_out = None
        if amax_out is not None:
            ck = ("amax", B, h, w, _os.environ.get("LAV_CONV_SPLIT"), _os.environ.get("LAV_SPLIT_SK"))
            n_out = self._ws_bytes.get(ck)
            if n_out is None:
                n_out = self._ws_bytes[ck] = lib.lav_conv_amax_count(C.byref(d))
            if n_out < 1:
                raise RuntimeError("lav_conv_amax_count: " + lib.lav_last_error().decode())
            a_out = amax_out.take(n_out)
        check(lib.lav_conv2d_amax(C.byref(d), _ptr(x), _ptr(self.w), _ptr(self.bias), _ptr(self.scale), _ptr(self.shift),
                                  _ptr(residual), _ptr(out), _ptr(ws), ws.numel() if ws is not None else 0,
                                  _ptr(a_in), amax_in.count if a_in is not None else 0, _ptr(a_out), _stream()),
              "lav_conv2d_amax")
        return out


class GroupedDeconv:
    """ConvTranspose2d layers with few output channels that share geometry and read disjoint channel groups of one
    input, as ONE lav_deconv_grouped launch (memory-bound vector kernel): the tails of the detection / segmentation heads
    (4 x 64 channels -> 1 + 2 + 2 + 4), or a single layer (groups = 1: ERFNet's output layer)."""

    def __init__(self, deconvs, sigmoid_from: int = -1, softmax: bool = False, device=None):
        ct0 = deconvs[0]
        for ct in deconvs:
            if (ct.kernel_size, ct.stride, ct.padding, ct.output_padding, ct.in_channels) != \
                    (ct0.kernel_size, ct0.stride, ct0.padding, ct0.output_padding, ct0.in_channels) or ct.dilation != (1, 1) or ct.groups != 1:
                raise RuntimeError("GroupedDeconv: layers must share kernel / stride / padding / input channels")
        k, s_ = ct0.kernel_size, ct0.stride
        if k[0] != k[1] or s_[0] != s_[1] or (k[0], s_[0]) not in ((3, 2), (2, 2)) or ct0.padding[0] != ct0.padding[1]:
            raise RuntimeError(f"GroupedDeconv: kernel {k} stride {s_} unsupported")
        dev = device if device is not None else ct0.weight.device
        self.k, self.s, self.pad, self.opad = k[0], s_[0], ct0.padding[0], ct0.output_padding[0]
        self.cin_g, self.groups = ct0.in_channels, len(deconvs)
        self.outs = [ct.out_channels for ct in deconvs]
        if max(self.outs) > 8 or self.groups > 8:
            raise RuntimeError("GroupedDeconv: at most 8 groups of at most 8 output channels")
        self._deconvs, self._dev = list(deconvs), dev
        self.w = self.bias = None
        self.refresh()
        self._outs_c = (C.c_int * self.groups)(*self.outs)
        self.sigmoid_from = -2 if softmax else int(sigmoid_from)     # -2: softmax over each group's channels, in the epilogue

    def refresh(self):
        """(Re-)read the layers' parameters: concatenated where they live, moved once."""
        w = torch.cat([ct.weight.detach().to(torch.float32).reshape(-1) for ct in self._deconvs]).to(self._dev)
        b = None
        if self._deconvs[0].bias is not None:
            b = torch.cat([ct.bias.detach().to(torch.float32) for ct in self._deconvs]).to(self._dev)
        if self.w is None:
            self.w, self.bias = w, b
        else:   # in place: captured graphs hold these addresses
            self.w.copy_(w)
            if b is not None:
                self.bias.copy_(b)

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None):
        x = _f32c(x, "x")
        B, c, h, w = x.shape
        if c != self.cin_g * self.groups:
            raise RuntimeError(f"GroupedDeconv input has {c} channels, expects {self.cin_g * self.groups}")
        oh = (h - 1) * self.s - 2 * self.pad + self.k + self.opad
        ow = (w - 1) * self.s - 2 * self.pad + self.k + self.opad
        if out is None:
            out = torch.empty((B, sum(self.outs), oh, ow), dtype=torch.float32, device=x.device)
        check(_lib.load().lav_deconv_grouped(B, c, h, w, self.groups, self._outs_c, self.k, self.s, self.pad, self.opad, _ptr(x),
                                             _ptr(self.w), _ptr(self.bias), self.sigmoid_from, _ptr(out), _stream()),
              "lav_deconv_grouped")
        return out


class PointwiseUpconv:
    """ConvTranspose2d with kernel == stride (no bias) -> ReLU -> eval BatchNorm into a channel slice of a wider map, as ONE
    lav_upconv_pointwise launch: exact fp32 matrix products, no pipeline to fill (round 6: the BEV backbone's 1x1 and 4x4 / stride-4
    up-convolutions, 21.5 + 31.1 us on the implicit-GEMM kernels).  Same call surface as the ConvLayer it stands in for:
    `__call__(x, out=, amax_in=, amax_out=)`, `out_hw`, `uses_amax` (never: it reads fp32), `refresh` (re-packs on the device)."""

    @staticmethod
    def takes(ct, device=None) -> bool:
        k, s = ct.kernel_size, ct.stride
        return (isinstance(ct, torch.nn.ConvTranspose2d) and k[0] == k[1] == s[0] == s[1] and k[0] in (1, 4) and ct.bias is None
                and ct.in_channels in (64, 128) and ct.out_channels % (128 if k[0] == 1 else 32) == 0 and ct.groups == 1
                and ct.dilation == (1, 1) and ct.padding[0] == ct.padding[1] < k[0] and ct.output_padding[0] == ct.output_padding[1]
                and (k[0] > 1 or (ct.padding[0] == 0 and ct.output_padding[0] == 0))
                and _os.environ.get("LAV_UPCONV_POINTWISE", "1") != "0")

    def __init__(self, ct, bn, *, relu_pre=True, out_c_total=None, out_c_offset=0, device=None):
        if not PointwiseUpconv.takes(ct):
            raise RuntimeError("PointwiseUpconv: ConvTranspose2d with kernel == stride in {1, 4}, 64 / 128 input channels, no bias")
        self.k, self.pad, self.opad = ct.kernel_size[0], ct.padding[0], ct.output_padding[0]
        self.cin, self.cout = ct.in_channels, ct.out_channels
        self.relu_pre = bool(relu_pre)
        self.out_c_total = int(out_c_total) if out_c_total else self.cout
        self.out_c_offset = int(out_c_offset)
        self._src = (ct.weight, bn)
        self.device = torch.device(device) if device is not None else ct.weight.device
        self.w = self.scale = self.shift = None
        self.refresh()

    def refresh(self):
        """Packed weights and folded BatchNorm from the CURRENT parameter values: a permutation and five small float64 launches on the
        device the parameters live on (lav_upconv_pointwise_pack is the host form of the same permutation)."""
        w, bn = self._src
        w = w.detach().to(torch.float32)
        if self.k == 1:   # [ci][co] -> [128-cout tile][32-cout tile j][ci][32]
            p = w.reshape(self.cin, self.cout // 128, 4, 32).permute(1, 2, 0, 3)
        else:             # [ci][co][ky][kx] -> [ky][32-cout tile][kx][ci][32]
            p = w.reshape(self.cin, self.cout // 32, 32, 4, 4).permute(3, 1, 4, 0, 2)
        p = p.contiguous().reshape(-1).to(self.device)
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
        if self.w is None:
            self.w, self.scale, self.shift = p, scale.float().to(self.device).contiguous(), shift.float().to(self.device).contiguous()
        else:     # in place: HIP graphs hold these addresses
            self.w.copy_(p); self.scale.copy_(scale.float()); self.shift.copy_(shift.float())

    def out_hw(self, h: int, w: int):
        f = lambda v: (v - 1) * self.k - 2 * self.pad + self.k + self.opad
        return f(h), f(w)

    def uses_amax(self, B: int, h: int, w: int) -> bool:
        return False

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, amax_in=None, amax_out=None) -> torch.Tensor:
        lib = _lib.load()
        x = _f32c(x, "x")
        B, c, h, w = x.shape
        if c != self.cin:
            raise RuntimeError(f"PointwiseUpconv: input has {c} channels, layer expects {self.cin}")
        oh, ow = self.out_hw(h, w)
        if out is None:
            out = torch.empty((B, self.out_c_total, oh, ow), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (B, self.out_c_total, oh, ow) or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
            raise RuntimeError(f"PointwiseUpconv: out must be a contiguous float32 {(B, self.out_c_total, oh, ow)} tensor in HBM")
        a_out = None
        if amax_out is not None:
            a_out = amax_out.take(lib.lav_upconv_pointwise_parts(B, self.cin, self.cout, h, w, self.k, self.pad, self.opad))
        check(lib.lav_upconv_pointwise(B, self.cin, self.cout, h, w, self.k, self.pad, self.opad, _ptr(x), _ptr(self.w), _ptr(self.scale),
                                       _ptr(self.shift), int(self.relu_pre), self.out_c_total, self.out_c_offset, _ptr(out), _ptr(a_out), _stream()),
              "lav_upconv_pointwise")
        return out


class Conv1dPair:
    """conv(3,1) -> ReLU -> conv(1,3) (+ eval BatchNorm, + residual, ReLU) as ONE lav_conv1d_pair launch: half of ERFNet's
    non_bottleneck_1d block.  `supported(x)` tells whether the row-tile kernel takes this shape."""

    def __init__(self, conv_a, conv_b, bn=None, relu_post=True, device=None):
        lib = _lib.load()
        ch = conv_a.in_channels
        if not (conv_a.kernel_size == (3, 1) and conv_b.kernel_size == (1, 3) and conv_a.out_channels == ch == conv_b.in_channels ==
                conv_b.out_channels and conv_a.stride == (1, 1) and conv_b.stride == (1, 1)
                and conv_a.padding == (conv_a.dilation[0], 0) and conv_b.padding == (0, conv_b.dilation[1])):
            raise RuntimeError("Conv1dPair: expects conv(3,1) then conv(1,3), same channels, stride 1, 'same' padding")
        self.ch, self.da, self.db = ch, conv_a.dilation[0], conv_b.dilation[1]
        dev = device if device is not None else conv_a.weight.device
        nfl = lib.lav_conv1d_pair_packed_weight_floats(ch)
        packed = []
        for conv in (conv_a, conv_b):
            w = conv.weight.detach().to("cpu", torch.float32).reshape(ch, ch, 3).contiguous()
            out = torch.empty(nfl, dtype=torch.float32)
            check(lib.lav_conv1d_pair_pack_weights(ch, w.data_ptr(), out.data_ptr()), "lav_conv1d_pair_pack_weights")
            packed.append(out.to(dev))
        self.wa, self.wb = packed
        self.ba = conv_a.bias.detach().to(dev, torch.float32).contiguous()
        self.bb = conv_b.bias.detach().to(dev, torch.float32).contiguous()
        self.scale = self.shift = None
        if bn is not None:
            s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            self.scale = s.float().to(dev)
            self.shift = (bn.bias.detach().double() - bn.running_mean.detach().double() * s).float().to(dev)
        self.relu_post = bool(relu_post)

    def supported(self, x: torch.Tensor) -> bool:
        w = x.shape[3]
        return (w in (32, 64, 128) and self.ch % 16 == 0 and (self.ch + 31) // 32 <= 128 // w and self.db <= 16
                and _lib.load().lav_conv1d_pair_lds_bytes(self.ch, w, self.db) <= 160 * 1024)

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _f32c(x, "x")
        B, ch, h, w = x.shape
        if ch != self.ch:
            raise RuntimeError(f"Conv1dPair: input has {ch} channels, layer expects {self.ch}")
        if residual is not None:
            residual = _f32c(residual, "residual")
            if residual.shape != x.shape:
                raise RuntimeError("residual shape mismatch")
        y = torch.empty_like(x)
        check(_lib.load().lav_conv1d_pair(B, ch, h, w, self.da, self.db, _ptr(x), _ptr(self.wa), _ptr(self.ba), _ptr(self.wb),
                                          _ptr(self.bb), _ptr(self.scale), _ptr(self.shift), _ptr(residual), int(self.relu_post),
                                          _ptr(y), _stream()), "lav_conv1d_pair")
        return y


class Conv1dPairChain:
    """A run of Conv1dPair layers over one map as ONE persistent launch (lav_conv1d_pair_chain): `pairs` in execution order,
    `residual[i]` True where pair i adds the input of pair i-1 (the non_bottleneck_1d block's input).  Output buffers are kept per
    (batch, h, w) - every pair writes its own, the last one is returned (a fresh tensor per call unless `reuse_out`)."""

    def __init__(self, pairs, residual):
        if len(pairs) != len(residual) or not pairs or residual[0]:
            raise RuntimeError("Conv1dPairChain: one residual flag per pair, the first pair without")
        if any(p.ch != pairs[0].ch or p.scale is None for p in pairs):
            raise RuntimeError("Conv1dPairChain: pairs of one channel count, each with its BatchNorm affine")
        self.pairs, self.residual = list(pairs), [int(bool(r)) for r in residual]
        self._bufs = {}
        # round 6: engines built for LAV_CONV_F16X3 (the inference pipelines' default) run on two fp16 pieces per operand
        # (lav_conv1d_pair_chain_f16); LAV_ERFNET_F16=0 keeps the bf16x6 run
        self.f16 = infer_precision() == _lib.CONV_F16X3 and _os.environ.get("LAV_ERFNET_F16", "1") != "0"

    def supported(self, x: torch.Tensor) -> bool:
        lib = _lib.load()
        B, ch, h, w = x.shape
        p0 = self.pairs[0]
        lds = lib.lav_conv1d_pair_chain_lds_bytes(p0.ch, w, max(p.db for p in self.pairs))
        return (_os.environ.get("LAV_CONV_PRECISION", "bf16x6") not in ("f32", "fp32") and _os.environ.get("LAV_ERFNET_CHAIN", "1") != "0"
                and len(self.pairs) <= 16 and all(p.supported(x) for p in self.pairs)
                and B * h <= _cu_count(x.device) * (2 if (lds <= 76 * 1024 and not (ch >= 64 and (ch // 16) % 2 == 0)) else 1)
                and lds <= 152 * 1024)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        x = _f32c(x, "x")
        B, ch, h, w = x.shape
        n = len(self.pairs)
        key = (B, h, w, x.device, _stream())
        bufs = self._bufs.get(key)
        if bufs is None:
            bufs = self._bufs[key] = [torch.empty_like(x) for _ in range(n - 1)]
        out = bufs + [torch.empty_like(x)]
        # sized ONCE for the largest run the chip can hold (2 rows per CU): the buffer never grows, so the sticky time-out / launch
        # counters of already captured graphs stay the ones pair_chain_status() reads
        ws = _workspace("pair_chain", max(lib.lav_conv1d_pair_chain_workspace_bytes(B, h), lib.lav_conv1d_pair_chain_workspace_bytes(pair_chain_capacity(x.device), 1)), x.device)
        ia = lambda vals: (C.c_int * n)(*vals)
        pa = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        ps = self.pairs
        check((lib.lav_conv1d_pair_chain_f16 if self.f16 else lib.lav_conv1d_pair_chain)(B, ch, h, w, n, ia([p.da for p in ps]), ia([p.db for p in ps]), ia(self.residual),
                                        ia([int(p.relu_post) for p in ps]), _ptr(x), pa([p.wa for p in ps]), pa([p.ba for p in ps]),
                                        pa([p.wb for p in ps]), pa([p.bb for p in ps]), pa([p.scale for p in ps]), pa([p.shift for p in ps]),
                                        pa(out), _ptr(ws), ws.numel(), _stream()), "lav_conv1d_pair_chain")
        return out[-1]

    def timeouts(self, x_like: torch.Tensor) -> int:
        """Workgroups of pair-chain launches on the current stream that gave up waiting for a neighbour row (0 = all results valid)."""
        return pair_chain_status(x_like.device)[0]


def pair_chain_capacity(device) -> int:
    """Rows of progress counters in the pair-chain workspace: room for several runs' regions (pair_chain_region)."""
    return 4 * _cu_count(device)


class pair_chain_region:
    """`with ops.pair_chain_region(row_offset, clean_rows):` - the Conv1dPairChain launches enqueued inside keep their progress counters at
    that offset of the workspace and zero the first `clean_rows` counters first (0: cleaned by an earlier launch; lav_conv1d_pair_chain_region).
    ERFNet's runs use it to share one cleaning launch per forward pass."""

    def __init__(self, row_offset: int, clean_rows: int):
        self.args = (int(row_offset), int(clean_rows))

    def __enter__(self):
        check(_lib.load().lav_conv1d_pair_chain_region(*self.args), "lav_conv1d_pair_chain_region")
        return self

    def __exit__(self, *exc):
        check(_lib.load().lav_conv1d_pair_chain_region(0, -1), "lav_conv1d_pair_chain_region")
        return False


def pair_chain_status(device, stream=None):
    """(workgroups that gave up, launches) of lav_conv1d_pair_chain on `stream` (default: the current one) since its workspace
    was created.  Synchronises that stream."""
    st = stream if stream is not None else torch.cuda.current_stream()
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ws = _workspaces.get(("pair_chain", device, st.cuda_stream))
    if ws is None:
        return (0, 0)
    v = (C.c_int * 2)()
    check(_lib.load().lav_conv1d_pair_chain_status(_ptr(ws), v, st.cuda_stream), "lav_conv1d_pair_chain_status")
    return (int(v[0]), int(v[1]))


_CU_COUNT = {}


def _cu_count(device) -> int:
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _CU_COUNT:
        _CU_COUNT[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _CU_COUNT[idx]


# algorithmic work of the training-side kernels since the last reset (bench.py's training roofline reads it next to the
# library's HIP-event timers): bytes the crop gradient must move, flops of the recurrent GEMMs
train_work = {"crop_rotate_backward_bytes": 0, "crop_rotate_backward_calls": 0, "gru_seq_forward_flops": 0, "gru_seq_backward_flops": 0,
              "bn_train_fwd_bytes": 0, "bn_train_bwd_bytes": 0, "conv_wgrad_flops": 0}


class _CropRotateIndexed(torch.autograd.Function):
    """Rotated crops taken from per-sample feature maps by index, differentiable in the maps (training path)."""

    @staticmethod
    def forward(ctx, features, map_index, locs, oris, ppm, crop, ox, oy):
        lib = _lib.load()
        features = _f32c(features, "features")
        locs, oris = _f32c(locs.detach().reshape(-1, 2), "locs"), _f32c(oris.detach().reshape(-1), "oris")
        if map_index.dtype != torch.int32 or not map_index.is_cuda:
            raise RuntimeError("crop_rotate_indexed: map_index must be an int32 tensor in HBM")
        M, Cc, H, W = features.shape
        n = locs.shape[0]
        out = torch.empty((n, Cc, crop, crop), dtype=torch.float32, device=features.device)
        check(lib.lav_crop_rotate_indexed(_ptr(features), M, _ptr(map_index), Cc, H, W, _ptr(locs), _ptr(oris), n, float(ppm), int(crop),
                                          float(ox), float(oy), _ptr(out), _stream()), "lav_crop_rotate_indexed")
        ctx.save_for_backward(map_index, locs, oris)
        ctx.geom = (M, Cc, H, W, float(ppm), int(crop), float(ox), float(oy))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        map_index, locs, oris = ctx.saved_tensors
        M, Cc, H, W, ppm, crop, ox, oy = ctx.geom
        grad_out = _f32c(grad_out, "grad_out")
        grad_feat = torch.empty((M, Cc, H, W), dtype=torch.float32, device=grad_out.device)
        train_work["crop_rotate_backward_bytes"] += 4 * (grad_out.numel() + grad_feat.numel())     # read every output gradient once, write the map gradient once
        train_work["crop_rotate_backward_calls"] += 1
        check(_lib.load().lav_crop_rotate_backward(_ptr(grad_out), M, _ptr(map_index), Cc, H, W, _ptr(locs), _ptr(oris), locs.shape[0], ppm,
                                                   crop, ox, oy, _ptr(grad_feat), _stream()), "lav_crop_rotate_backward")
        return grad_feat, None, None, None, None, None, None, None


def crop_rotate_indexed(features, map_index, locs, oris, pixels_per_meter, crop, offset_x, offset_y):
    """features (M,C,H,W); crop i = rotated crop of features[map_index[i]] at (locs[i], oris[i]) -> (n,C,crop,crop)."""
    return _CropRotateIndexed.apply(features, map_index, locs, oris, pixels_per_meter, crop, offset_x, offset_y)


def det_decode(rows: torch.Tensor, actors: torch.Tensor, n_out: torch.Tensor, *, cls: int, min_score: float, ego_xy, near_px: float,
               far_px: float, min_box: float, centre_xy, skip_px: float, ppm: float, report=None):
    """Peak rows (ncls, max_det, 7) of lav_extract_peaks -> the other vehicles' ego-frame (x, y) and headings in `actors`
    ([2*max_det] + [max_det] floats) and their number in `n_out` (int32[1]), all in HBM (lav_det_decode).
    report = (host_rows float32 like rows, host_n int32[1], host_seq int32[1]), three PINNED host tensors: the launch also writes rows
    and count there and then increments host_seq (lav_det_decode_report) - the host polls the word instead of copying."""
    rows = _f32c(rows, "rows")
    ncls, max_det, _ = rows.shape
    if actors.numel() < 3 * max_det or actors.dtype != torch.float32 or n_out.dtype != torch.int32:
        raise RuntimeError("det_decode: actors must hold 3*max_det float32, n_out one int32")
    host = [None, None, None]
    if report is not None:
        h_rows, h_n, h_seq = report
        if not (h_rows.is_pinned() and h_n.is_pinned() and h_seq.is_pinned() and h_rows.is_contiguous()) or h_rows.dtype != torch.float32 \
                or h_rows.numel() < rows.numel() or h_n.dtype != torch.int32 or h_seq.dtype != torch.int32:
            raise RuntimeError("det_decode: report = (pinned float32 rows, pinned int32[1] count, pinned int32[1] sequence word)")
        host = [h_rows.data_ptr(), h_n.data_ptr(), h_seq.data_ptr()]
    check(_lib.load().lav_det_decode_report(_ptr(rows), ncls, max_det, int(cls), float(min_score), float(ego_xy[0]), float(ego_xy[1]),
                                            float(near_px), float(far_px), float(min_box), float(centre_xy[0]), float(centre_xy[1]),
                                            float(skip_px), float(ppm), _ptr(actors), _ptr(n_out), *host, _stream()), "lav_det_decode")


class batch_limit:
    """`with ops.batch_limit(d_n):` - the conv / crop / cast launches enqueued inside skip the rows >= d_n[0], a count that
    lives in HBM and is read when the kernels run (lav_batch_limit)."""

    def __init__(self, d_rows: torch.Tensor):
        if d_rows.dtype != torch.int32 or not d_rows.is_cuda:
            raise RuntimeError("batch_limit: an int32 tensor in HBM")
        self.d_rows = d_rows

    def __enter__(self):
        check(_lib.load().lav_batch_limit(_ptr(self.d_rows)), "lav_batch_limit")
        return self

    def __exit__(self, *exc):
        check(_lib.load().lav_batch_limit(None), "lav_batch_limit")
        return False


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """F.max_pool2d(x, 3, 2, 1) on liblav_amd (lav_maxpool3x3s2); honours batch_limit."""
    x = _f32c(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    check(_lib.load().lav_maxpool3x3s2(_ptr(x), B, Cc, H, W, _ptr(out), _stream()), "lav_maxpool3x3s2")
    return out


def channel_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """x (B,C,H,W) * scale[c] + shift[c] in one launch (lav_channel_affine); H*W must be a multiple of 4."""
    x = _f32c(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    check(_lib.load().lav_channel_affine(_ptr(x), B, Cc, H * W, _ptr(_f32c(scale, "scale")), _ptr(_f32c(shift, "shift")), _ptr(out), _stream()),
          "lav_channel_affine")
    return out


def nonfinite_count(tensors, counter: torch.Tensor) -> None:
    """counter (2,) int32 in HBM: [0] += number of `tensors` (float32, contiguous, <= 8) holding a NaN / Inf, [1] += 1.
    One launch, no host sync (include/lav_amd.h: lav_nonfinite_count)."""
    ts = [_f32c(t, "tensor") for t in tensors]
    n = len(ts)
    if counter.dtype != torch.int32 or counter.numel() < 2 or not counter.is_cuda:
        raise RuntimeError("nonfinite_count: counter must be an int32 tensor of two elements in HBM")
    check(_lib.load().lav_nonfinite_count(n, (C.c_void_p * n)(*[t.data_ptr() for t in ts]), (C.c_long * n)(*[t.numel() for t in ts]),
                                          _ptr(counter), _stream()), "lav_nonfinite_count")


def copy_many(pairs, block=None) -> None:
    """[(dst, src), ...] device tensors of equal shape into their destinations in at most two launches: contiguous same-dtype
    16-byte-aligned pairs through lav_copy_many (16-byte words), strided / uint8 sources of up to four dimensions into contiguous
    float32 destinations through lav_stage_many (the conversion torch's copy_ would do, for all of them at once); anything else
    takes Tensor.copy_.
    block = (dst device tensor, src HOST tensor of the same <= 256 bytes, a multiple of 4): rides in the staging launch's kernel
    arguments (lav_stage_many_block) instead of an upload of its own; copied with Tensor.copy_ when there is no such launch."""
    srcs, dsts, sizes, stage, plain = [], [], [], [], []
    for dst, src in pairs:
        nb = dst.numel() * dst.element_size()
        both = src.is_cuda and dst.is_cuda and dst.is_contiguous() and src.shape == dst.shape and nb > 0
        if (both and src.is_contiguous() and src.dtype == dst.dtype and nb % 4 == 0 and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0):
            srcs.append(src.data_ptr()); dsts.append(dst.data_ptr()); sizes.append(nb); plain.append((dst, src))
        elif both and dst.dtype == torch.float32 and src.dtype in (torch.float32, torch.uint8) and 1 <= src.dim() <= 4:
            stage.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    lib = _lib.load()
    if _os.environ.get("LAV_COPY_MERGE", "1") != "0" and stage and plain and len(stage) + len(plain) <= 8 and all(d.dtype == torch.float32 for d, _ in plain):
        # (round 6) one launch instead of two: the few contiguous float32 pairs of a frame (LiDAR rows, next waypoint) ride with the
        # strided ones - every launch on the frame's critical chain costs ~4.5 us whatever it moves
        stage = [(d.reshape(-1), s_.reshape(-1)) for d, s_ in plain] + stage
        srcs = []
    for i in range(0, len(srcs), 8):
        n = len(srcs[i:i + 8])
        check(lib.lav_copy_many(n, (C.c_void_p * n)(*srcs[i:i + 8]), (C.c_void_p * n)(*dsts[i:i + 8]), (C.c_size_t * n)(*sizes[i:i + 8]), _stream()),
              "lav_copy_many")
    blk = (None, 0, None)
    if block is not None:
        bdst, bsrc = block
        nb = bsrc.numel() * bsrc.element_size()
        if (stage and not bsrc.is_cuda and bsrc.is_contiguous() and bdst.is_cuda and bdst.is_contiguous() and nb % 4 == 0 and 0 < nb <= 256
                and nb == bdst.numel() * bdst.element_size() and bdst.data_ptr() % 4 == 0):
            blk = (bsrc.data_ptr(), nb, bdst.data_ptr())
        else:
            bdst.copy_(bsrc, non_blocking=True)
    for i in range(0, len(stage), 8):
        part = stage[i:i + 8]
        n = len(part)
        dims, strides = [], []
        for dst, src in part:
            pad = 4 - src.dim()
            dims += [1] * pad + list(src.shape)
            strides += [0] * pad + list(src.stride())
        b = blk if i == 0 else (None, 0, None)
        check(lib.lav_stage_many_block(n, (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in part]), (C.c_void_p * n)(*[d.data_ptr() for d, _ in part]),
                                       (C.c_int * (4 * n))(*dims), (C.c_long * (4 * n))(*strides),
                                       (C.c_int * n)(*[int(s_.dtype == torch.uint8) for _, s_ in part]), b[0], b[1], b[2], _stream()), "lav_stage_many")


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias, sigmoid: bool = False) -> torch.Tensor:
    """act(x @ weight.T + bias) for small layers, x (B,K) in HBM, weight (O,K) (lav_linear_act)."""
    x = _f32c(x, "x")
    B, K = x.shape
    O = weight.shape[0]
    out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    check(_lib.load().lav_linear_act(_ptr(x), B, K, _ptr(_f32c(weight, "weight")), _ptr(None if bias is None else _f32c(bias, "bias")), O,
                                     int(bool(sigmoid)), _ptr(out), _stream()), "lav_linear_act")
    return out


def attn_pool(x: torch.Tensor, u: torch.Tensor, dots_bias: torch.Tensor, w_v: torch.Tensor, b_v: torch.Tensor, heads: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B,C,h,w) in HBM -> (B,C): single-query multi-head attention pooling with folded projections (lav_attn_pool).
    out: a (B,C) buffer whose rows are C apart (e.g. one half of a (1,2C) concatenation buffer at batch 1)."""
    x = _f32c(x, "x")
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, Cc) or out.dtype != torch.float32 or not out.is_cuda or (B > 1 and out.stride(0) != Cc) or out.stride(1) != 1:
        raise RuntimeError("attn_pool: out must be a (B, C) float32 buffer in HBM with rows C apart")
    check(_lib.load().lav_attn_pool(_ptr(x), B, Cc, H * W, int(heads), _ptr(_f32c(u, "u")), _ptr(_f32c(dots_bias, "dots_bias")),
                                    _ptr(_f32c(w_v, "w_v")), _ptr(_f32c(b_v, "b_v")), _ptr(out), _stream()), "lav_attn_pool")
    return out


def pool_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, out: torch.Tensor, out_c_offset: int, relu: bool = True):
    """2x2 max pooling + per-channel affine (+ ReLU) of x (B,C,H,W) into channels [out_c_offset, +C) of out (B,Ct,H/2,W/2)."""
    x = _f32c(x, "x")
    B, Cc, H, W = x.shape
    if out.shape[0] != B or out.shape[2:] != (H // 2, W // 2) or not out.is_contiguous():
        raise RuntimeError("pool_affine: output buffer does not match")
    check(_lib.load().lav_pool_affine(_ptr(x), B, Cc, H, W, _ptr(_f32c(scale, "scale")), _ptr(_f32c(shift, "shift")), int(relu),
                                      _ptr(out), out.shape[1], int(out_c_offset), _stream()), "lav_pool_affine")
    return out


def crop_rotate(features: torch.Tensor, locs: torch.Tensor, oris: torch.Tensor, pixels_per_meter: float, crop: int,
                offset_x: float, offset_y: float) -> torch.Tensor:
    """features (1 or N, C, H, W), locs (N,2), oris (N,) in HBM -> (N, C, crop, crop)."""
    lib = _lib.load()
    features = _f32c(features, "features")
    locs = _f32c(locs.reshape(-1, 2), "locs")
    oris = _f32c(oris.reshape(-1), "oris")
    fb, Cc, H, W = features.shape
    n = locs.shape[0]
    if fb not in (1, n):
        raise RuntimeError("features batch must be 1 (shared map) or N")
    out = torch.empty((n, Cc, crop, crop), dtype=torch.float32, device=features.device)
    check(lib.lav_crop_rotate(_ptr(features), fb, Cc, H, W, _ptr(locs), _ptr(oris), n, float(pixels_per_meter), int(crop),
                              float(offset_x), float(offset_y), _ptr(out), _stream()), "lav_crop_rotate")
    return out


def merge_ticks(tick: torch.Tensor, prev: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """cat([tick, prev]) with ego-box points marked x = NaN; then prev <- tick, in place.  tick, prev (P,4)."""
    lib = _lib.load()
    tick, prev = _f32c(tick, "tick"), _f32c(prev, "prev")
    if tick.shape != prev.shape or tick.dim() != 2 or tick.shape[1] != 4:
        raise RuntimeError(f"merge_ticks: tick {tuple(tick.shape)} / prev {tuple(prev.shape)} must both be (P, 4)")
    P = tick.shape[0]
    if out is None:
        out = torch.empty((2 * P, 4), dtype=torch.float32, device=tick.device)
    check(lib.lav_merge_ticks(_ptr(tick), _ptr(prev), P, 4, _ptr(out), _stream()), "lav_merge_ticks")
    return out


def stack_sweeps(fused: torch.Tensor, ring: torch.Tensor, slot: torch.Tensor, sweeps: torch.Tensor, R: torch.Tensor,
                 t: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """History write + temporal stacking: fused (rows,8), ring (slots,rows,8), slot (1,) int64, sweeps (S,) int64,
    R (S,3,3), t (S,3) - all in HBM -> (S*rows, 8+S)."""
    lib = _lib.load()
    fused, ring, R, t = _f32c(fused, "fused"), _f32c(ring, "ring"), _f32c(R, "R"), _f32c(t.reshape(-1, 3), "t")
    S, rows, dim = sweeps.numel(), fused.shape[0], fused.shape[1]
    if ring.shape[1:] != fused.shape or slot.dtype != torch.long or sweeps.dtype != torch.long or R.shape != (S, 3, 3):
        raise RuntimeError("stack_sweeps: inconsistent arguments")
    if out is None:
        out = torch.empty((S * rows, dim + S), dtype=torch.float32, device=fused.device)
    check(lib.lav_stack_sweeps(_ptr(fused), _ptr(ring), _ptr(slot), _ptr(sweeps), _ptr(R), _ptr(t), S, rows, dim, _ptr(out),
                               _stream()), "lav_stack_sweeps")
    return out


def extract_peaks(heat: torch.Tensor, size: torch.Tensor, ori: torch.Tensor, ks: int = 7, max_det: int = 15,
                  apply_sigmoid: bool = False) -> torch.Tensor:
    """heat (ncls,H,W), size (cs,H,W), ori (co,H,W) -> (ncls, max_det, 3+cs+co) rows (score, x, y, size.., ori..)."""
    lib = _lib.load()
    heat, size, ori = _f32c(heat, "heat"), _f32c(size, "size"), _f32c(ori, "ori")
    ncls, H, W = heat.shape
    if size.shape[1:] != (H, W) or ori.shape[1:] != (H, W):
        raise RuntimeError("extract_peaks: size / ori maps must match the heat map")
    out = torch.empty((ncls, max_det, 3 + size.shape[0] + ori.shape[0]), dtype=torch.float32, device=heat.device)
    nbytes = lib.lav_extract_peaks_workspace_bytes(ncls, H, W)
    ws = _workspace("peaks", nbytes, heat.device)
    check(lib.lav_extract_peaks(_ptr(heat), ncls, H, W, ks, max_det, int(apply_sigmoid), _ptr(size), size.shape[0], _ptr(ori),
                                ori.shape[0], _ptr(out), _ptr(ws), ws.numel(), _stream()), "lav_extract_peaks")
    return out

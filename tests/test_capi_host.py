"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/lav_amd.h
declares; host-side helpers (weight packing, plan geometry, BatchNorm folding, state_dict keys) are right.
No compute call needs a GPU here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import lav_amd
from lav_amd import _lib, synth
from lav_amd._lib import Conv

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(REPO, "include", "lav_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lav_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/lav_amd.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in lav_amd/_lib.py"
    assert lib.lav_abi_version() == _lib.ABI_VERSION


def test_ops_refuse_cpu_tensors_loudly():
    ppn = lav_amd.PointPillarNet(16, [64, 64]).eval()
    with pytest.raises(RuntimeError, match="HBM|device|HIP"):
        ppn([torch.zeros(8, 11)], [8])


def conv_desc(cin, cout, k, s, p, tr=False, op=0, h=8, w=8):
    return Conv(1, cin, 0, cin, h, w, cout, k, k, s, p, p, 1, 1, int(tr), op, cout, 0, 0, 0, 0)


def tap_matrix(pk, class_off, ntaps, t, cin_pad, cout_pad):
    """[cin_pad][cout_pad] weights of tap t of one class out of the packed layout
    [cout block of 32][tap][8-channel group][channel parity][cout in block][channel pair] (include/lav_amd.h)."""
    blk = pk[class_off:class_off + ntaps * cin_pad * cout_pad].reshape(cout_pad // 32, ntaps, cin_pad // 8, 2, 32, 4)
    # channel = 8*group + 2*pair + parity, cout = 32*block + column
    return blk[:, t].transpose(1, 4, 2, 0, 3).reshape(cin_pad, cout_pad)


@pytest.mark.parametrize("cin,cout,k,s,p,tr,op", [(5, 7, 3, 1, 1, False, 0), (40, 70, 3, 2, 1, False, 0), (6, 3, 4, 2, 1, True, 0), (4, 5, 4, 4, 1, True, 2),
                                                  (3, 2, 3, 2, 1, True, 1), (2, 3, 1, 1, 0, True, 0)])
def test_conv_weight_packing_and_class_decomposition(cin, cout, k, s, p, tr, op):
    """Emulate the kernel's gather from the packed layout on the CPU and compare with torch: checks the host-side
    plan (parity classes, tap order, padded strides) without a GPU."""
    lib = _lib.load()
    H = W = 7
    d = conv_desc(cin, cout, k, s, p, tr, op, H, W)
    r = np.random.Generator(np.random.PCG64(3))
    w = torch.from_numpy(r.standard_normal((cin, cout, k, k) if tr else (cout, cin, k, k)).astype(np.float32))
    n = lib.lav_conv_packed_weight_floats(C.byref(d))
    packed = torch.zeros(n)
    assert lib.lav_conv_pack_weights(C.byref(d), w.data_ptr(), packed.data_ptr()) == 0
    oh, ow = C.c_int(), C.c_int()
    assert lib.lav_conv_out_hw(C.byref(d), C.byref(oh), C.byref(ow)) == 0
    x = torch.from_numpy(r.standard_normal((1, cin, H, W)).astype(np.float32))
    ref = torch.nn.functional.conv_transpose2d(x, w, None, s, p, op) if tr else torch.nn.functional.conv2d(x, w, None, s, p)
    assert (oh.value, ow.value) == tuple(ref.shape[2:])
    cin_pad, cout_pad = (cin + 15) // 16 * 16, (cout + 63) // 64 * 64
    pk = packed.numpy()
    out = np.zeros((cout, oh.value, ow.value), np.float64)
    xn = x[0].numpy()
    if not tr:
        taps = [(ky, kx) for ky in range(k) for kx in range(k)]
        for t, (ky, kx) in enumerate(taps):
            wt = tap_matrix(pk, 0, len(taps), t, cin_pad, cout_pad)[:cin, :cout]
            for oy in range(oh.value):
                for ox in range(ow.value):
                    iy, ix = oy * s - p + ky, ox * s - p + kx
                    if 0 <= iy < H and 0 <= ix < W:
                        out[:, oy, ox] += xn[:, iy, ix] @ wt
    else:
        off = 0
        for ry in range(s):
            for rx in range(s):
                nty = (k - ry + s - 1) // s if ry < k else 0
                ntx = (k - rx + s - 1) // s if rx < k else 0
                for dy in range(nty):
                    for dx in range(ntx):
                        wt = tap_matrix(pk, off, nty * ntx, dy * ntx + dx, cin_pad, cout_pad)[:cin, :cout]
                        for qy in range((oh.value - 1 + p) // s + 1):
                            for qx in range((ow.value - 1 + p) // s + 1):
                                oy, ox = s * qy + ry - p, s * qx + rx - p
                                iy, ix = qy - (nty - 1) + dy, qx - (ntx - 1) + dx
                                if 0 <= oy < oh.value and 0 <= ox < ow.value and 0 <= iy < H and 0 <= ix < W:
                                    out[:, oy, ox] += xn[:, iy, ix] @ wt
                off += nty * ntx * cin_pad * cout_pad
    np.testing.assert_allclose(out, ref[0].numpy(), atol=1e-4)


def test_pointnet_batchnorm_folding_matches_unfolded_math():
    from oracle import pillar as opillar
    from tests.util import pointnet_sd_numpy, state_dicts, sub_sd
    lsd, _ = state_dicts()
    ppn = lav_amd.PointPillarNet(16, [64, 64]).eval()
    ppn.load_state_dict(sub_sd(lsd, "point_pillar_net."))
    w1, b1, w2, b2 = [t.numpy().astype(np.float64) for t in ppn.point_net.folded(torch.device("cpu"))]
    f = np.random.Generator(np.random.PCG64(0)).standard_normal((50, 16)).astype(np.float32)
    ref = opillar.point_net(f, pointnet_sd_numpy())
    h = np.maximum(np.maximum(f @ w1 + b1, 0) @ w2 + b2, 0)
    np.testing.assert_allclose(h, ref, atol=2e-5, rtol=1e-5)


def test_state_dict_keys_match_reference_checkpoint_layout():
    """Key lists recorded from the reference modules by tests/golden/make_golden.py."""
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "state_dict_keys.json")))
    from tests.util import build_models
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    lm, up = build_models("cpu")
    got = dict(lidar=lm, uniplanner=up, seg=RGBSegmentationModel([4, 6, 7, 10]), bra=RGBBrakePredictionModel([4, 6, 7, 10]))
    for name, m in got.items():
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref[name], name


def test_world_size_2_gloo_replica_timing_reduction():
    """bench.py's N>1 contract on CPU: barrier + MAX-reduced wall time over ranks (gloo, world_size 2)."""
    import subprocess
    import sys
    code = r'''
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
dist.barrier(); t0 = time.perf_counter(); time.sleep(0.05 * (r + 1)); dt = time.perf_counter() - t0
t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() >= 0.1 - 1e-3, t
if r == 0: print("MAX_OK", round(t.item(), 2))
dist.destroy_process_group()
'''
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29613", "-c", code] if False else
                         [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(REPO, "tests", "_gloo_worker.py")],
                         capture_output=True, text=True, timeout=180)
    assert "MAX_OK" in out.stdout, out.stdout + out.stderr


def _conv_shapes_of(model, x_shape):
    """(conv module, input shape) for every Conv2d / ConvTranspose2d reached by a CPU forward of `model`."""
    shapes = []
    hooks = []
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: shapes.append((mod, tuple(inp[0].shape)))))
    with torch.no_grad():
        model(*[torch.zeros(s) for s in x_shape])
    for h in hooks:
        h.remove()
    return shapes


def test_every_layer_shape_of_the_frame_has_a_launch_plan():
    """Host-side plan feasibility (tile shape, staging map, LDS) for every convolution the frame runs,
    without a GPU: ERFNet on 3x(288x256), brake ResNet on 288x768 and 192x480, BEV stack, ResNet-18 on crops."""
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    lib = _lib.load()
    # train mode: the modules' torch path (eval mode is HIP only and raises on CPU tensors); the layer shapes are the same
    seg = RGBSegmentationModel([4, 6, 7, 10]).train()
    bra = RGBBrakePredictionModel([4, 6, 7, 10]).train()
    todo = _conv_shapes_of(seg, [(3, 3, 288, 256)]) + _conv_shapes_of(bra, [(1, 3, 288, 768), (1, 3, 192, 480)])
    res = lav_amd.resnet18(num_channels=384)
    res.train(False)
    # torch-only walk over the ResNet / BEV modules (their HIP forward needs a GPU): enumerate by hand
    def walk(convs, x):
        for conv in convs:
            todo.append((conv, tuple(x.shape)))
            with torch.no_grad():
                x = conv(x)
        return x
    bb = lav_amd.ConvBackbone(64)
    x1 = walk([m for m in bb.conv1 if isinstance(m, torch.nn.Conv2d)], torch.zeros(1, 64, 320, 320))
    x2 = walk([m for m in bb.conv2 if isinstance(m, torch.nn.Conv2d)], x1)
    x3 = walk([m for m in bb.conv3 if isinstance(m, torch.nn.Conv2d)], x2)
    for up, x in ((bb.upconv1[0], x1), (bb.upconv2[0], x2), (bb.upconv3[0], x3)):
        todo.append((up, tuple(x.shape)))
    head = lav_amd.Head(384, 3)
    todo += [(head.net[0], (1, 384, 160, 160)), (head.net[3], (1, 64, 160, 160))]
    todo.append((torch.nn.Conv2d(384, 256, 3, 1, 1), (1, 384, 160, 160)))           # the fused 4-head convolution
    for nb in (1, 16):
        x = walk([res.conv1], torch.zeros(nb, 384, 96, 96))
        x = torch.nn.functional.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            for blk in getattr(res, f"layer{i}"):
                if blk.downsample is not None:
                    todo.append((blk.downsample[0], tuple(x.shape)))
                x = walk([blk.conv1, blk.conv2], x)
    assert len(todo) > 150
    info = (C.c_int * 9)()
    for conv, shp in todo:
        tr = isinstance(conv, torch.nn.ConvTranspose2d)
        cin, cout = (conv.in_channels, conv.out_channels)
        d = Conv(shp[0], cin, 0, cin, shp[2], shp[3], cout, conv.kernel_size[0], conv.kernel_size[1], conv.stride[0],
                 conv.padding[0], conv.padding[1], conv.dilation[0], conv.dilation[1], int(tr),
                 conv.output_padding[0] if tr else 0, cout, 0, 0, 0, 0)
        rc = lib.lav_conv_tile_info(C.byref(d), info)
        assert rc == 0, f"{conv} on {shp}: {lib.lav_last_error().decode()}"
        assert info[3] * info[4] <= 256 * 12 and info[5] <= 160 * 1024


def test_vectorised_detection_decode_equals_the_loops():
    """InferModel.det_decode_fast (numpy masks; used on the frame's critical chain) == det_decode +
    UniPlanner.others_from_detections (the row loops that restate model_inference.py:95-144)."""
    import types
    from lav_amd.model_inference import InferModel
    from lav_amd.uniplanner import UniPlanner
    up = types.SimpleNamespace(pixels_per_meter=4, offsets=lambda: (0.0, 0.75))
    up.others_from_detections = lambda det, H, W: UniPlanner.others_from_detections(up, det, H, W)
    im = types.SimpleNamespace(pixels_per_meter=4, uniplanner=up, _bev_hw=(320, 320))
    rng = np.random.default_rng(0)
    for trial in range(50):
        rows = np.zeros((2, 15, 7), np.float32)
        rows[..., 0] = rng.uniform(-0.2, 1.0, (2, 15))
        rows[..., 1] = rng.integers(100, 220, (2, 15)); rows[..., 2] = rng.integers(150, 319, (2, 15))
        rows[..., 3:5] = rng.uniform(0, 3, (2, 15, 2)); rows[..., 5:7] = rng.normal(size=(2, 15, 2))
        rows[1, 0, 1:3] = (160, 280); rows[1, 1, 1:3] = (161, 282); rows[1, 2, 1:3] = (163, 281)    # ego pixel neighbourhood
        rows[0, 3, 0] = 0.2                                                                          # score threshold is strict
        dets = InferModel.det_decode(im, rows.tolist())
        locs, oris = up.others_from_detections(dets[1], 320, 320)
        fd, fl, fo = InferModel.det_decode_fast(im, rows)
        assert fd == dets
        np.testing.assert_allclose(fl, np.asarray(locs, np.float64).reshape(-1, 2), rtol=0, atol=0)
        np.testing.assert_allclose(fo, np.asarray(oris, np.float64), rtol=0, atol=1e-15)


def test_direct_plan_is_chosen_where_it_was_measured_faster():
    """Host-side plan choice (no GPU): small ResNet maps and the mid-size BEV layers take the direct kernel
    (info[0] == 0, info[1] waves, info[2] cout blocks, info[6] split-K); 7x7 stems, the deep 2x2-tiled head convolution
    and ragged-channel layers stay on the tiled kernel; every direct plan keeps a whole number of 8-channel
    groups per wave and a reduction scratch within the LDS."""
    lib = _lib.load()
    info = (C.c_int * 9)()

    def plan(B, cin, cout, k, s, p, H, W, tr=False, precision=_lib.CONV_F32):
        d = Conv(B, cin, 0, cin, H, W, cout, k, k, s, p, p, 1, 1, int(tr), 0, cout, 0, 0, 0, 0, 0, 0.0, precision)
        assert lib.lav_conv_tile_info(C.byref(d), info) == 0, lib.lav_last_error().decode()
        return list(info)

    direct = [(1, 64, 64, 3, 1, 1, 24, 24), (7, 64, 64, 3, 1, 1, 24, 24), (1, 128, 128, 3, 1, 1, 12, 12), (7, 256, 256, 3, 1, 1, 6, 6),
              (1, 512, 512, 3, 1, 1, 3, 3), (15, 512, 512, 3, 1, 1, 3, 3), (1, 256, 512, 1, 2, 0, 6, 6), (1, 64, 64, 3, 1, 1, 160, 160),
              (1, 128, 128, 3, 1, 1, 80, 80), (1, 64, 128, 3, 2, 1, 160, 160), (3, 16, 48, 3, 2, 1, 144, 128)]
    for shp in direct:
        i = plan(*shp)
        cin = shp[1]
        assert i[0] == 0, f"{shp}: expected the direct kernel, got tile {i[0]}x{i[1]}"
        waves, mc, ks = i[1], i[2], i[6]
        assert waves in (1, 2, 4, 8, 16) and mc in (1, 2) and cin % (8 * waves * ks) == 0
        assert i[5] == waves * mc * 32 * 33 * 4 <= 160 * 1024
        assert mc == 1 or shp[2] >= 64
    tiled = [(1, 384, 64, 7, 2, 3, 96, 96), (7, 384, 64, 7, 2, 3, 96, 96), (1, 384, 256, 3, 1, 1, 160, 160), (1, 13, 48, 3, 1, 1, 20, 33)]
    for shp in tiled:
        assert plan(*shp)[0] >= 1, f"{shp}: expected a tiled plan"
    # the camera stems (3 input channels, stride 2) run on packed fp32 FMAs (round 5, conv_smallcin.hpp): plan {-2, tile 32 x 8, workgroups}
    for shp, th, wgs in (((1, 3, 64, 7, 2, 3, 288, 768), 8, 216), ((1, 3, 64, 7, 2, 3, 192, 480), 8, 96), ((3, 3, 13, 3, 2, 1, 288, 256), 8, 216)):
        i = plan(*shp)
        assert i[:4] == [-2, 32, th, wgs], f"{shp}: expected the small-cin kernel, got {i}"
    # transposed convolutions run on the direct kernel too (one launch, output-parity classes in grid.z)
    i = plan(1, 128, 128, 4, 2, 1, 80, 80, tr=True)
    assert i[0] == 0 and i[2] == 2
    assert plan(1, 128, 128, 4, 4, 0, 40, 40, tr=True)[:3] == [0, 4, 1]


def test_split_plan_takes_the_matrix_bound_layers_only():
    """Precision bf16x6 (the default): the split kernel (info[0] == -1; info[1..4] = MP, MC, pixel waves, tile width; info[5] LDS
    bytes; info[6] split-K) takes the layers that are bound by the matrix pipes - the fused 384->256 head convolution, the
    64-channel BEV layers at 160x160, the 4x4 up-convolution - and leaves the small ResNet maps to the exact fp32 direct kernel;
    with precision f32 it never runs."""
    lib = _lib.load()
    info = (C.c_int * 9)()

    def plan(B, cin, cout, k, s, p, H, W, tr=False, precision=_lib.CONV_BF16X6):
        d = Conv(B, cin, 0, cin, H, W, cout, k, k, s, p, p, 1, 1, int(tr), 0, cout, 0, 0, 0, 0, 0, 0.0, precision)
        assert lib.lav_conv_tile_info(C.byref(d), info) == 0, lib.lav_last_error().decode()
        return list(info)

    head = plan(1, 384, 256, 3, 1, 1, 160, 160)
    assert head[0] == -1 and (head[1], head[2]) == (2, 2) and head[5] <= 160 * 1024 and head[4] % 32 == 0
    assert plan(1, 64, 64, 3, 1, 1, 160, 160)[0] == -1
    assert plan(1, 128, 128, 4, 2, 1, 80, 80, tr=True)[0] == -1
    for shp in [(1, 64, 64, 3, 1, 1, 24, 24), (1, 128, 128, 3, 1, 1, 12, 12), (7, 256, 256, 3, 1, 1, 6, 6), (1, 512, 512, 3, 1, 1, 3, 3)]:
        assert plan(*shp)[0] == 0, f"{shp}: small maps stay on the direct kernel"
    assert plan(1, 384, 256, 3, 1, 1, 160, 160, precision=_lib.CONV_F32)[0] >= 1
    # the packed weights of a split-capable layer carry both layouts
    d32 = Conv(1, 64, 0, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1, 1, 0, 0, 64, 0, 0, 0, 0, 0, 0.0, _lib.CONV_F32)
    d16 = Conv(1, 64, 0, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1, 1, 0, 0, 64, 0, 0, 0, 0, 0, 0.0, _lib.CONV_BF16X6)
    n32, n16 = lib.lav_conv_packed_weight_floats(C.byref(d32)), lib.lav_conv_packed_weight_floats(C.byref(d16))
    assert n32 == 64 * 64 * 9 and n16 == n32 + 64 * 64 * 9 * 3 // 2


def test_crop_stems_are_planned_in_tap_pair_mode():
    """The 7x7 stride-2 stems of the crop embedders (384 input channels, 96x96 crops) run on the split kernel's tap-pair mode
    (round 4: info[7] = tap group + 100): 16x8-pixel tiles of 1x2 wave tiles, 150 KB of LDS, split-K so that the launch fills the
    chip at 1 / 4 / 7 crops, at the others branch's capacity of 15 and at a training batch; the brake net's 3-channel stems
    (cin % 16 != 0) and 3x3 layers never do, and LAV_CONV_PRECISION / precision f32 keeps the stems on the fp32 tiled kernel."""
    lib = _lib.load()
    info = (C.c_int * 9)()

    def plan(B, cin, cout, k, s, p, H, W, precision=_lib.CONV_BF16X6):
        d = Conv(B, cin, 0, cin, H, W, cout, k, k, s, p, p, 1, 1, 0, 0, cout, 0, 0, 0, 0, 0, 0.0, precision)
        assert lib.lav_conv_tile_info(C.byref(d), info) == 0, lib.lav_last_error().decode()
        return list(info)

    for B in (1, 4, 7, 15, 32):
        i = plan(B, 384, 64, 7, 2, 3, 96, 96)
        assert i[0] == -1 and i[7] >= 100, f"batch {B}: {i}"
        assert (i[1], i[2], i[3]) == (1, 2, 4) and (i[4], i[8]) == (16, 8) and i[5] <= 160 * 1024 and i[7] % 100 == 4
        wgs = 18 * B * i[6]
        assert B >= 15 or 200 <= wgs <= 256, f"batch {B}: split-K {i[6]} gives {wgs} workgroups"
    assert plan(1, 3, 64, 7, 2, 3, 288, 768)[7] < 100          # brake stem: 3 input channels
    assert plan(1, 384, 256, 3, 1, 1, 160, 160)[7] < 100       # 3x3 layers keep 16-channel chunks
    assert plan(7, 384, 64, 7, 2, 3, 96, 96, precision=_lib.CONV_F32)[0] >= 1


def test_grouped_deconv_rejects_unsupported_geometry_before_any_launch():
    lib = _lib.load()
    outs = (C.c_int * 2)(3, 9)
    dummy = C.c_void_p(64)   # never dereferenced: argument checks come first
    assert lib.lav_deconv_grouped(1, 64, 8, 8, 2, outs, 3, 2, 1, 1, dummy, dummy, None, -1, dummy, None) != 0   # 9 couts in a group
    assert b"output channels" in lib.lav_last_error()
    outs = (C.c_int * 2)(3, 4)
    assert lib.lav_deconv_grouped(1, 64, 8, 8, 2, outs, 4, 2, 1, 0, dummy, dummy, None, -1, dummy, None) != 0   # 4x4 kernel
    assert lib.lav_deconv_grouped(1, 65, 8, 8, 2, outs, 3, 2, 1, 1, dummy, dummy, None, -1, dummy, None) != 0   # 65 % 2


@pytest.mark.parametrize("case", [(64, 128, 3, 2, False), (384, 256, 3, 1, False), (128, 64, 3, 2, True), (3, 64, 7, 2, False), (130, 37, 1, 1, False)])
def test_pack_map_reproduces_the_host_packing_bit_for_bit(case):
    """lav_conv_pack_map (the index map lav_conv_repack gathers with on the device) applied to a random weight on the host
    - gather, then the three-piece bf16 split restated in numpy - gives exactly lav_conv_pack_weights' buffer."""
    import numpy as np
    cin, cout, k, s, tr = case
    lib = _lib.load()
    rng = np.random.default_rng(11)
    w = rng.standard_normal((cin, cout, k, k) if tr else (cout, cin, k, k)).astype(np.float32)
    for prec in (_lib.CONV_F32, _lib.CONV_BF16X6):
        d = _lib.Conv(1, cin, 0, cin, 64, 64, cout, k, k, s, k // 2, k // 2, 1, 1, int(tr), 1 if tr else 0, cout, 0, 0, 0, 0, 0, 0.0, prec)
        nfl = lib.lav_conv_packed_weight_floats(C.byref(d))
        packed = np.zeros(nfl, np.float32)
        assert lib.lav_conv_pack_weights(C.byref(d), w.ctypes.data, packed.ctypes.data) == 0
        nm = lib.lav_conv_pack_map_ints(C.byref(d))
        m = np.zeros(nm, np.int32)
        assert lib.lav_conv_pack_map(C.byref(d), m.ctypes.data) == 0
        flat = np.concatenate([w.reshape(-1), np.zeros(1, np.float32)])          # index -1 -> 0
        if prec == _lib.CONV_F32:
            assert nm == nfl
            assert np.array_equal(flat[m].view(np.uint32), packed.view(np.uint32))
            continue
        # nm = nf + ntrip and nfl * 4 = nf * 4 + ntrip * 6
        ntrip = (nfl * 4 - nm * 4) // 2
        nf = nm - ntrip
        assert nf * 4 + ntrip * 6 == nfl * 4
        assert np.array_equal(flat[m[:nf]].view(np.uint32), packed[:nf].view(np.uint32))
        v = flat[m[nf:]]
        def piece(x):
            u = (x.view(np.uint32).astype(np.uint64) + 0x8000) & 0xffff0000
            return u.astype(np.uint32)
        p0 = piece(v); r1 = v - p0.view(np.float32)
        p1 = piece(r1); r2 = r1 - p1.view(np.float32)
        p2 = piece(r2)
        want = packed[nf:].view(np.uint16).reshape(-1, 3, 512)
        got = np.stack([(p >> 16).astype(np.uint16).reshape(-1, 512) for p in (p0, p1, p2)], axis=1)
        assert np.array_equal(got, want)


def test_split_kernel_is_never_planned_for_classes_without_taps():
    """A transposed convolution whose kernel is smaller than its stride has output-parity classes with no tap; the split-operand
    kernel streams weights unconditionally and must not be chosen for them (any batch / size)."""
    lib = _lib.load()
    for cin, cout in ((64, 128), (128, 64), (512, 256)):
        for B, H in ((1, 12), (57, 12), (16, 48), (4, 160)):
            d = Conv(B, cin, 0, cin, H, H, cout, 1, 1, 2, 0, 0, 1, 1, 1, 1, cout, 0, 0, 0, 0, 0, 0.0, _lib.CONV_BF16X6)
            info = (C.c_int * 9)()
            assert lib.lav_conv_tile_info(C.byref(d), info) == 0
            assert info[0] != -1, (cin, cout, B, H, list(info))


def test_library_holds_no_ds_write2_b64(tmp_path):
    """hipcc pairs adjacent 64-bit LDS stores into ds_write2_b64 and may overwrite the store's data registers with the very next
    instructions (its hazard recognizer only pads stores whose FIRST data operand is wider than 64 bits); on gfx950 that store
    then writes the new register contents whenever another kernel's waves keep the SIMD's matrix pipe busy - round 4's
    wrong-but-finite plans (profiles/r04_plan_stress.txt, lav_amd/csrc/common.hpp: lds_store_fence).  No such instruction may be
    left in the built library."""
    import glob
    import shutil
    import subprocess
    from lav_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is missing")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert objs, "no gfx950 code object found in the library"
    bad = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        bad += asm.count("ds_write2_b64") + asm.count("ds_write2st64_b64")
    assert bad == 0, f"{bad} ds_write2_b64 instructions in liblav_amd.so: separate the stores with lav::lds_store_fence()"


def test_plan_kernel_has_no_lds_instruction_and_no_barrier(tmp_path):
    """Round 5: the frame's persistent plan kernel (k_plan_wave) must not be able to become a victim of the co-residency effect of
    DESIGN 4.4c (LDS-dependent results going wrong beside matrix + LDS heavy neighbours): it holds no LDS, issues no DS instruction
    (its cross-lane sums are v_permlane swaps and DPP adds, not ds_bpermute) and no barrier.  Disassembles the built library."""
    import glob
    import re
    import shutil
    import subprocess
    from lav_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is missing")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    found = 0
    for o in glob.glob(str(tmp_path / "lib.so.*gfx950")):
        asm = subprocess.run([objdump, "-d", o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(_Z[^>]*k_plan_wave[^>]*)>:\n(.*?)(?=^\S|\Z)", asm, re.S | re.M):
            name, body = m.group(1), m.group(2)
            found += 1
            ops_ = re.findall(r"^\s+(\w+)", body, re.M)
            assert len(ops_) > 500, (name, len(ops_))
            bad = sorted({op for op in ops_ if op.startswith("ds_") or op.startswith("s_barrier")})
            assert not bad, f"{name}: {bad}"
            assert any(op.startswith("v_permlane32_swap") for op in ops_) and any("dpp" in op for op in ops_), name
    assert found == 4, f"{found} k_plan_wave instantiations found (expected <1,8>, <1,0>, <6,8>, <6,0>)"


def test_library_holds_no_packed_fp32_with_op_sel(tmp_path):
    """Round 5, the root cause of round 4's finite-but-wrong results (profiles/r05_coresidency.md): on gfx950 a packed
    fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose op_sel bit is set for a source - the LOW result takes the
    HIGH register of the 64-bit source pair - returns wrong values in lanes 48-63 while waves of a bf16-matrix + LDS heavy kernel
    share its SIMD (tools/lds_hazard.py pattern 55: one such v_pk_mul_f32 in a loop, evaluated twice on the same inputs, disagrees
    with itself 535 936 times in 1.26e10 beside ERFNet's 16-channel persistent run and never alone).  hipcc's SLP vectoriser emits
    that form freely; rounds 2-4's plan kernel held exactly one (the waypoint sum) and lav_crop_rotate three.  The library is built
    with -fno-slp-vectorize and writes its packed instructions by hand with op_sel = 0; none with an op_sel bit may be left."""
    import glob
    import re
    import shutil
    import subprocess
    from lav_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is missing")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert objs, "no gfx950 code object found in the library"
    bad = []
    for o in objs:
        asm = subprocess.run([objdump, "-d", o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for line in asm.splitlines():
            if re.search(r"\bv_pk_\w+_f32\b", line) and re.search(r"op_sel:\[[01,]*1", line):
                bad.append(line.split("//")[0].strip())
    assert not bad, f"{len(bad)} packed fp32 instructions with an op_sel bit in liblav_amd.so, e.g. {bad[:3]}"


def test_f16x3_plan_and_packing():
    """LAV_CONV_F16X3 (round 5, host side): the head convolution's plan carries the fp16 flag and a workspace for the absmax launch's
    per-workgroup maxima; other layers of that precision plan as bf16x6; the packed buffer holds, behind the fp32 and bf16 sections, two
    fp16 pieces per weight of w / s_w with s_w the power of two that puts the largest |w| into [16384, 32768): h0 + h1 reproduces every
    weight to 2^-21 of its magnitude + 2^-24 s_w (= 2^-39 of the largest weight: fp16's subnormal quantum)."""
    import ctypes as C
    import numpy as np
    from lav_amd import _lib
    from lav_amd._lib import Conv
    lib = _lib.load()
    info = (C.c_int * 9)()
    head = lambda prec: Conv(1, 384, 0, 384, 160, 160, 256, 3, 3, 1, 1, 1, 1, 1, 0, 0, 256, 0, 0, 0, 0, 0, 0.0, prec)
    d2, d3 = head(_lib.CONV_BF16X6), head(_lib.CONV_F16X3)
    assert lib.lav_conv_tile_info(C.byref(d3), info) == 0 and info[0] == -1 and info[7] == 202, list(info)
    assert lib.lav_conv_tile_info(C.byref(d2), info) == 0 and info[7] == 2
    assert lib.lav_conv_workspace_bytes(C.byref(d3)) == 2048 and lib.lav_conv_workspace_bytes(C.byref(d2)) == 0
    # round 6: every split plan takes the mode - a 64-channel BEV layer, a split-K layer (its slabs + the maxima behind them), a
    # tap-pair stem, the classes of an up-convolution; layers on the direct / tiled fp32 kernels ignore it
    mk = lambda B, cin, cout, k, s_, p_, H, W, prec, tr=0, op=0: Conv(B, cin, 0, cin, H, W, cout, k, k, s_, p_, p_, 1, 1, tr, op, cout, 0, 0, 0, 0, 0, 0.0, prec)
    for args in ((1, 64, 64, 3, 1, 1, 160, 160), (1, 512, 512, 3, 1, 1, 9, 24), (7, 384, 64, 7, 2, 3, 96, 96), (1, 128, 128, 4, 2, 1, 80, 80, 1, 0)):
        d16, d6 = mk(*args[:8], _lib.CONV_F16X3, *args[8:]), mk(*args[:8], _lib.CONV_BF16X6, *args[8:])
        info6 = (C.c_int * 9)()
        assert lib.lav_conv_tile_info(C.byref(d16), info) == 0 and lib.lav_conv_tile_info(C.byref(d6), info6) == 0
        assert info[0] == -1 and info[7] == info6[7] + 200 and list(info[:7]) == list(info6[:7]), (args, list(info), list(info6))
        assert lib.lav_conv_workspace_bytes(C.byref(d16)) == lib.lav_conv_workspace_bytes(C.byref(d6)) + 2048
        assert lib.lav_conv_packed_weight_floats(C.byref(d16)) > lib.lav_conv_packed_weight_floats(C.byref(d6))
        assert lib.lav_conv_amax_count(C.byref(d16)) == lib.lav_conv_amax_count(C.byref(d6)) >= 1
    tiny = mk(1, 256, 256, 3, 1, 1, 6, 6, _lib.CONV_F16X3)
    assert lib.lav_conv_tile_info(C.byref(tiny), info) == 0 and info[0] == 0, "a 6x6 map stays on the direct fp32 kernel"
    assert lib.lav_conv_amax_count(C.byref(tiny)) >= 1
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((256, 384, 3, 3)) * np.exp(rng.uniform(-5, 2, (256, 384, 3, 3))) / 60).astype(np.float32)
    n2, n3 = lib.lav_conv_packed_weight_floats(C.byref(d2)), lib.lav_conv_packed_weight_floats(C.byref(d3))
    nb16 = 9 * 8 * 24 * 2 * 1024
    assert n3 == n2 + nb16 // 4 + 4
    p2, p3 = np.zeros(n2, np.float32), np.zeros(n3, np.float32)
    assert lib.lav_conv_pack_weights(C.byref(d2), w.ctypes.data, p2.ctypes.data) == 0
    assert lib.lav_conv_pack_weights(C.byref(d3), w.ctypes.data, p3.ctypes.data) == 0
    assert np.array_equal(p2.view(np.uint32), p3[:n2].view(np.uint32)), "the fp32 and bf16 sections are the bf16x6 layer's"
    sec = p3[n2:].view(np.uint8)
    sw = float(sec[nb16:nb16 + 4].view(np.float32)[0])
    assert sw == 2.0 ** round(np.log2(sw)) and 16384 <= np.abs(w).max() / sw < 32768
    h = sec[:nb16].view(np.float16).astype(np.float64).reshape(8, 9, 24, 2, 64, 8)    # cout block, tap, chunk, piece, lane, channel
    rec = (h[:, :, :, 0] + h[:, :, :, 1]) * sw
    blk, tap, ch, lane, el = np.meshgrid(np.arange(8), np.arange(9), np.arange(24), np.arange(64), np.arange(8), indexing="ij")
    ref = w[blk * 32 + (lane & 31), ch * 16 + 8 * (lane >> 5) + el, tap // 3, tap % 3].astype(np.float64)
    # h0 carries 11 bits of w / s_w, h1 11 more of what is left - down to fp16's subnormal quantum 2^-24 (in units of s_w)
    assert (np.abs(rec - ref) <= np.abs(ref) * 2.0 ** -21 + sw * 2.0 ** -24).all()
    assert lib.lav_conv_repack(C.byref(d3), None, None, None, None) != 0        # (null arguments are refused)
    # the index map of the fp16 section (what lav_conv_repack_scratch gathers with): one entry per fp16 pair, in the section's order
    nm2, nm3 = lib.lav_conv_pack_map_ints(C.byref(d2)), lib.lav_conv_pack_map_ints(C.byref(d3))
    assert nm3 == nm2 + nb16 // 4
    m = np.zeros(nm3, np.int32)
    assert lib.lav_conv_pack_map(C.byref(d3), m.ctypes.data) == 0
    m16 = m[nm2:].reshape(8, 9, 24, 64, 8)
    assert np.array_equal(w.reshape(-1)[m16].astype(np.float64), ref)


def test_the_test_session_keeps_miopen_databases_to_itself():
    """tests/conftest.py: a GPU test session must not leave its (deterministic-mode) solver choices in MIOpen's default user database -
    the next process on the box inherits them (round 5: train_full at 634 instead of 128 ms per step, profiles/r05_miopen_db_poisoning.txt)."""
    import os
    import tempfile
    d = os.environ.get("MIOPEN_USER_DB_PATH")
    assert d and os.path.isdir(d) and os.path.realpath(d) != os.path.realpath(os.path.expanduser("~/.config/miopen"))
    if os.path.realpath(d).startswith(os.path.realpath(tempfile.gettempdir())) or "lav_amd" in os.path.realpath(d):
        return
    pytest.skip("MIOPEN_USER_DB_PATH was set by the caller: its isolation is the caller's business")

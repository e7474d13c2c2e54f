"""lav_conv2d (fp32 MFMA implicit GEMM, through the C ABI) against torch CPU convolutions in float32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lav_amd.ops import ConvLayer
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rnd(shape, seed, scale=1.0):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((r.standard_normal(shape) * scale).astype(np.float32))


CASES = [
    # name, cin, cout, k, stride, pad, dil, transposed, out_pad, H, W
    ("3x3 s1 64->64", 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 40, 56),
    ("3x3 s2 64->128", 64, 128, (3, 3), 2, (1, 1), (1, 1), False, 0, 80, 80),
    ("3x3 s1 128->128 40x40", 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 40, 40),
    ("3x3 s2 odd size", 64, 64, (3, 3), 2, (1, 1), (1, 1), False, 0, 37, 51),
    ("1x1 convT 64->128", 64, 128, (1, 1), 1, (0, 0), (1, 1), True, 0, 48, 48),
    ("4x4 convT s2 p1", 128, 128, (4, 4), 2, (1, 1), (1, 1), True, 0, 24, 20),
    ("4x4 convT s4 p1 op2", 128, 128, (4, 4), 4, (1, 1), (1, 1), True, 2, 12, 10),
    ("3x3 convT s2 p1 op1 64->3", 64, 3, (3, 3), 2, (1, 1), (1, 1), True, 1, 32, 40),
    ("7x7 s2 p3 384->64", 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("1x1 s2 64->128", 64, 128, (1, 1), 2, (0, 0), (1, 1), False, 0, 24, 24),
    ("3x1 dil 4", 16, 16, (3, 1), 1, (4, 0), (4, 1), False, 0, 36, 32),
    ("1x3 dil 2 cin 13", 13, 48, (1, 3), 1, (0, 2), (1, 2), False, 0, 20, 33),
    ("3x3 cin 3", 3, 13, (3, 3), 2, (1, 1), (1, 1), False, 0, 64, 48),
    ("7x7 s2 wide image (row-blocked tiles)", 3, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 768),
    ("3x3 s1 wide 16ch", 16, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 12, 1000),
    ("3x3 512->512 on 3x3 (split-K 16)", 512, 512, (3, 3), 1, (1, 1), (1, 1), False, 0, 3, 3),
    ("3x3 s2 256->512 on 6x6 (split-K)", 256, 512, (3, 3), 2, (1, 1), (1, 1), False, 0, 6, 6),
    ("1x1 s2 256->512 downsample", 256, 512, (1, 1), 2, (0, 0), (1, 1), False, 0, 6, 6),
    ("7x7 s2 p3 384->64 batch crop", 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_matches_torch(case):
    name, cin, cout, k, s, p, d, tr, op, H, W = case
    B = 2
    x = rnd((B, cin, H, W), 1)
    fan = cin * k[0] * k[1]
    w = rnd((cin, cout, *k) if tr else (cout, cin, *k), 2, scale=1.0 / np.sqrt(fan))
    if tr:
        ref = F.conv_transpose2d(x, w, None, s, p, op)
    else:
        ref = F.conv2d(x, w, None, s, p, d)
    layer = ConvLayer(w, stride=s, padding=p, dilation=d, transposed=tr, output_padding=op, device=DEV)
    y = layer(x.to(DEV)).cpu()
    assert_close(y.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what=name)


def test_conv_epilogues_and_channel_windows():
    B, cin, cout, H, W = 1, 64, 64, 24, 40
    x = rnd((B, cin + 32, H, W), 3)
    w = rnd((cout, cin, 3, 3), 4, scale=0.05)
    bias = rnd((cout,), 5)
    bn = (rnd((cout,), 6, 0.1), rnd((cout,), 7).abs() + 0.5, rnd((cout,), 8).abs() + 0.5, rnd((cout,), 9, 0.1))
    res = rnd((B, cout + 16, H, W), 10)
    xin = x[:, 16:16 + cin]
    # BEV block: conv -> relu -> bn
    ref = F.batch_norm(F.relu(F.conv2d(xin, w, None, 1, 1)), bn[0], bn[1], bn[2], bn[3], False, 0., 1e-3)
    layer = ConvLayer(w, padding=1, bn=bn, bn_eps=1e-3, relu_pre=True, in_c_total=cin + 32, in_c_offset=16,
                      out_c_total=cout + 16, out_c_offset=8, device=DEV)
    out = torch.full((B, cout + 16, H, W), 7.0, device=DEV)
    layer(x.to(DEV), out=out)
    out = out.cpu()
    assert_close(out[:, 8:8 + cout].numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what="relu->bn window")
    assert (out[:, :8] == 7).all() and (out[:, 8 + cout:] == 7).all(), "channels outside the window were touched"
    # ResNet block tail: conv -> bn -> + identity -> relu ; plus bias and sigmoid
    ref2 = torch.sigmoid(F.relu(F.batch_norm(F.conv2d(xin, w, bias, 1, 1), bn[0], bn[1], bn[2], bn[3], False, 0., 1e-5) + res[:, :cout]))
    layer2 = ConvLayer(w, padding=1, bias=bias, bn=bn, bn_eps=1e-5, relu_post=True, sigmoid=True, in_c_total=cin + 32,
                       in_c_offset=16, device=DEV)
    y2 = layer2(x.to(DEV), residual=res[:, :cout].contiguous().to(DEV)).cpu()
    assert_close(y2.numpy(), ref2.numpy(), atol=1e-5, what="bias+bn+residual+relu+sigmoid")


def test_conv_transpose_detecting():
    """A = identity-like weights with an asymmetric input: catches swapped MFMA operands / row-col mixups."""
    cin = cout = 64
    w = torch.zeros((cout, cin, 1, 1))
    for c in range(cout):
        w[c, (c * 7 + 3) % cin, 0, 0] = 1.0 + c
    x = torch.arange(cin * 8 * 40, dtype=torch.float32).reshape(1, cin, 8, 40) * 1e-3
    ref = F.conv2d(x, w)
    y = ConvLayer(w, device=DEV)(x.to(DEV)).cpu()
    assert_close(y.numpy(), ref.numpy(), atol=1e-4, rtol=1e-6, what="permutation conv")


def test_split_k_epilogue_matches_unsplit():
    """bias + BN + residual + ReLU through the split-K reduce kernel (deep layer, tiny image)."""
    B, cin, cout, H, W = 1, 256, 256, 6, 6
    x = rnd((B, cin, H, W), 21)
    w = rnd((cout, cin, 3, 3), 22, scale=0.02)
    bias = rnd((cout,), 23)
    bn = (rnd((cout,), 24, 0.1), rnd((cout,), 25).abs() + 0.5, rnd((cout,), 26).abs() + 0.5, rnd((cout,), 27, 0.1))
    res = rnd((B, cout, H, W), 28)
    ref = F.relu(F.batch_norm(F.conv2d(x, w, bias, 1, 1), bn[0], bn[1], bn[2], bn[3], False, 0., 1e-5) + res)
    layer = ConvLayer(w, padding=1, bias=bias, bn=bn, relu_post=True, device=DEV)
    y = layer(x.to(DEV), residual=res.to(DEV)).cpu()
    assert_close(y.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what="split-K epilogue")


@pytest.mark.parametrize("cin,k,cout,B,H,W", [(3, 7, 64, 1, 288, 768), (3, 7, 64, 2, 50, 70), (3, 3, 13, 3, 288, 256), (3, 3, 13, 2, 37, 45), (3, 7, 40, 1, 20, 24),
                                              (3, 3, 16, 1, 16, 64)])
def test_small_cin_vector_kernel(cin, k, cout, B, H, W):
    """The camera stems (3 input channels, stride 2: brake net 7x7 -> 64, ERFNet 3x3 -> 13) on k_conv_smallcin (packed fp32 FMAs, round 5):
    an fp32 FMA chain per output - held to 2e-6 of sum |w||x| against float64 - ragged tiles, a folded-normalisation pad value, every
    epilogue piece, channel windows on both sides, and the layers' real sizes; the plan says which kernel ran."""
    import ctypes as C
    from lav_amd import _lib
    from lav_amd._lib import Conv
    lib = _lib.load()
    info = (C.c_int * 9)()
    d = Conv(B, cin, 0, cin, H, W, cout, k, k, 2, k // 2, k // 2, 1, 1, 0, 0, cout, 0, 0, 0, 0)
    assert lib.lav_conv_tile_info(C.byref(d), info) == 0 and info[0] == -2, f"expected the small-cin kernel, plan {list(info)}"
    x = rnd((B, cin + 2, H, W), 51)
    w = rnd((cout, cin, k, k), 52, scale=1.0 / np.sqrt(cin * k * k))
    bias = rnd((cout,), 53)
    bn = (rnd((cout,), 54, 0.1), rnd((cout,), 55).abs() + 0.5, rnd((cout,), 56).abs() + 0.5, rnd((cout,), 57, 0.1))
    xin = x[:, 1:1 + cin]
    ref64 = F.conv2d(xin.double(), w.double(), None, 2, k // 2)
    mag = F.conv2d(xin.double().abs(), w.double().abs(), None, 2, k // 2)
    y = ConvLayer(w, stride=2, padding=k // 2, in_c_total=cin + 2, in_c_offset=1, device=DEV)(x.to(DEV)).cpu()
    err = ((y.double() - ref64).abs() / mag.clamp_min(1e-30)).max().item()
    assert err < 2e-6, f"max |y - ref| / sum|w||x| = {err:.3e}"
    # conv -> + bias -> relu -> bn into a channel window, out-of-image taps reading a pad value (ERFNet's folded input normalisation)
    pv = 0.37
    xp = F.pad(xin, (k // 2,) * 4, value=pv)
    ref = F.batch_norm(F.relu(F.conv2d(xp, w, bias, 2, 0)), bn[0], bn[1], bn[2], bn[3], False, 0., 1e-3)
    layer = ConvLayer(w, stride=2, padding=k // 2, bias=bias, bn=bn, bn_eps=1e-3, relu_pre=True, in_c_total=cin + 2, in_c_offset=1, out_c_total=cout + 5,
                      out_c_offset=2, pad_value=pv, device=DEV)
    out = torch.full((B, cout + 5, ref.shape[2], ref.shape[3]), 7.0, device=DEV)
    layer(x.to(DEV), out=out)
    out = out.cpu()
    assert_close(out[:, 2:2 + cout].numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what="small-cin epilogue window")
    assert (out[:, :2] == 7).all() and (out[:, 2 + cout:] == 7).all(), "channels outside the window were touched"


def test_f16x3_head_convolution():
    """LAV_CONV_F16X3 (round 5): the head convolution's plan on two fp16 pieces per operand and three products, the activation scale taken
    from the tensor's largest finite magnitude by a first launch.  Against float64 on inputs that span five decades: the error bar of the
    bf16x6 kernel (2e-6 of sum |w||x|); the epilogue; an all-zero input; an Inf that must stay local; same bits on a second launch."""
    import ctypes as C
    from lav_amd import _lib
    from lav_amd._lib import Conv
    lib = _lib.load()
    info = (C.c_int * 9)()
    d = Conv(1, 384, 0, 384, 160, 160, 256, 3, 3, 1, 1, 1, 1, 1, 0, 0, 256, 0, 0, 0, 0, 0, 0.0, _lib.CONV_F16X3)
    assert lib.lav_conv_tile_info(C.byref(d), info) == 0 and info[0] == -1 and info[7] >= 200, f"expected the fp16 three-product plan, got {list(info)}"
    B, cin, cout, H, W = 1, 384, 256, 160, 160
    g = np.random.Generator(np.random.PCG64(61))
    x = torch.from_numpy((g.standard_normal((B, cin, H, W)) * np.exp(g.uniform(-8.0, 3.0, (B, cin, H, W)))).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((cout, cin, 3, 3)) * np.exp(g.uniform(-4.0, 1.0, (cout, cin, 3, 3))) / np.sqrt(cin * 9)).astype(np.float32))
    bias = rnd((cout,), 63)
    bn = (rnd((cout,), 64, 0.1), rnd((cout,), 65).abs() + 0.5, rnd((cout,), 66).abs() + 0.5, rnd((cout,), 67, 0.1))
    xd, wd = x.to(DEV).double(), w.to(DEV).double()
    conv64 = F.conv2d(xd, wd, None, 1, 1)
    mag = F.conv2d(xd.abs(), wd.abs(), None, 1, 1)
    plain = ConvLayer(w, padding=1, precision=_lib.CONV_F16X3, device=DEV)
    y = plain(x.to(DEV))
    err = ((y.double() - conv64).abs() / mag).max().item()
    assert err < 2e-6, f"max |y - ref| / sum|w||x| = {err:.3e}"
    y6 = ConvLayer(w, padding=1, device=DEV)(x.to(DEV))
    err6 = ((y6.double() - conv64).abs() / mag).max().item()
    print(f"head convolution, five decades of input magnitude: f16x3 {err:.2e}, bf16x6 {err6:.2e} of sum|w||x|")
    assert torch.equal(y, plain(x.to(DEV))), "bit-reproducible"
    ref = F.batch_norm(F.relu(conv64 + bias.to(DEV).double()[None, :, None, None]), bn[0].to(DEV).double(), bn[1].to(DEV).double(), bn[2].to(DEV).double(),
                       bn[3].to(DEV).double(), False, 0., 1e-3).float()
    layer = ConvLayer(w, padding=1, bias=bias, bn=bn, bn_eps=1e-3, relu_pre=True, out_c_total=cout + 8, out_c_offset=8, precision=_lib.CONV_F16X3, device=DEV)
    out = torch.full((B, cout + 8, H, W), 7.0, device=DEV)
    layer(x.to(DEV), out=out)
    assert (out[:, :8] == 7).all()
    # (the inputs span five decades: the bar stays relative to sum |w||x|, carried through the BatchNorm scale)
    gain = (bn[2].abs() / torch.sqrt(bn[1] + 1e-3)).to(DEV).double()[None, :, None, None]
    excess = ((out[:, 8:].double() - ref.double()).abs() - (2e-6 * mag * gain + 1e-5 * ref.double().abs() + 1e-6)).max().item()
    assert excess <= 0, f"f16x3 epilogue: beyond 2e-6 of sum|w||x| (through the BatchNorm gain) by {excess:.3e}"
    z = layer(torch.zeros((B, cin, H, W), device=DEV))
    zr = F.batch_norm(F.relu(bias.to(DEV)[None, :, None, None].expand(B, cout, H, W)), bn[0].to(DEV), bn[1].to(DEV), bn[2].to(DEV), bn[3].to(DEV), False, 0., 1e-3)
    assert_close(z[:, 8:].cpu().numpy(), zr.cpu().numpy(), atol=1e-6, what="all-zero input")
    xi = x.clone(); xi[0, 5, 80, 80] = float("inf")
    yi = plain(xi.to(DEV))
    assert not torch.isfinite(yi[0, :, 79:82, 79:82]).all(), "an Inf input must reach the outputs that read it"
    far = torch.ones((H, W), dtype=torch.bool); far[78:83, 78:83] = False
    assert torch.equal(yi[0][:, far], y[0][:, far]), "an Inf input must not change the scale: every other output keeps its bits"


def test_stream_k_head_convolution(monkeypatch):
    """LAV_SPLIT_SK=1 (opt-in: no faster on this power-bound chip, profiles/r05_clock_power.txt): the head convolution's plan (384 ->
    256, 3x3, one 160 x 160 image: 400 tiles on 256 CUs) runs as a stream-K launch - persistent
    workgroups share the K work evenly, a cut tile's two parts go through slabs and k_conv_sk_fixup (conv_split.hpp).  Against float64:
    the error bar of the split kernel (2e-6 of sum |w||x|); every epilogue piece through both the direct and the fix-up path; the
    untouched channels of a wider output stay untouched; two launches give the same bits (fixed cuts, fixed order)."""
    import ctypes as C
    from lav_amd import _lib
    from lav_amd._lib import Conv
    lib = _lib.load()
    info = (C.c_int * 9)()
    d = Conv(1, 384, 0, 384, 160, 160, 256, 3, 3, 1, 1, 1, 1, 1, 0, 0, 256, 0, 0, 0, 0)
    assert lib.lav_conv_tile_info(C.byref(d), info) == 0 and info[6] == 1, "whole tiles by default"
    monkeypatch.setenv("LAV_SPLIT_SK", "1")
    assert lib.lav_conv_tile_info(C.byref(d), info) == 0
    assert info[0] == -1 and info[6] == -256, f"the head convolution is expected on the stream-K launch, plan {list(info)}"
    B, cin, cout, H, W = 1, 384, 256, 160, 160
    x = rnd((B, cin, H, W), 41)
    w = rnd((cout, cin, 3, 3), 42, scale=1.0 / np.sqrt(cin * 9))
    bias = rnd((cout,), 43)
    bn = (rnd((cout,), 44, 0.1), rnd((cout,), 45).abs() + 0.5, rnd((cout,), 46).abs() + 0.5, rnd((cout,), 47, 0.1))
    res = rnd((B, cout, H, W), 48)
    xd, wd = x.to(DEV).double(), w.to(DEV).double()
    conv64 = F.conv2d(xd, wd, None, 1, 1)
    mag = F.conv2d(xd.abs(), wd.abs(), None, 1, 1)
    # plain convolution: the arithmetic
    y = ConvLayer(w, padding=1, device=DEV)(x.to(DEV))
    err = ((y.double() - conv64).abs() / mag).max().item()
    assert err < 2e-6, f"max |y - ref| / sum|w||x| = {err:.3e}"
    # epilogue through both paths, into a channel window
    ref = F.relu(F.batch_norm(conv64 + bias.to(DEV).double()[None, :, None, None], bn[0].to(DEV).double(), bn[1].to(DEV).double(), bn[2].to(DEV).double(),
                              bn[3].to(DEV).double(), False, 0., 1e-5) + res.to(DEV).double()).float()
    layer = ConvLayer(w, padding=1, bias=bias, bn=bn, relu_post=True, device=DEV)
    outs = []
    for _ in range(2):
        outs.append(layer(x.to(DEV), residual=res.to(DEV)).clone())
    assert torch.equal(outs[0], outs[1]), "stream-K results must be bit-reproducible"
    assert_close(outs[0].cpu().numpy(), ref.cpu().numpy(), atol=3e-5, rtol=1e-5, what="stream-K epilogue")
    sig = ConvLayer(w, padding=1, bias=bias, sigmoid=True, out_c_total=cout + 24, out_c_offset=8, device=DEV)
    out = torch.full((B, cout + 24, H, W), 7.0, device=DEV)
    sig(x.to(DEV), out=out)
    assert (out[:, :8] == 7).all() and (out[:, 8 + cout:] == 7).all(), "channels outside the window were touched"
    assert_close(out[:, 8:8 + cout].cpu().numpy(), torch.sigmoid(conv64 + bias.to(DEV).double()[None, :, None, None]).float().cpu().numpy(), atol=1e-5, what="stream-K sigmoid window")


def test_crop_rotate_matches_torch_grid_sample():
    from lav_amd import ops
    from lav_amd.planner_common import crop_feature_torch
    feat = rnd((1, 48, 160, 160), 31)
    locs = torch.tensor([[0.0, 0.0], [-5.0, -20.0], [7.5, -32.5], [30.0, 10.0], [-12.0, 3.0]])
    oris = torch.tensor([0.0, 0.3, -1.2, 2.9, -3.1])
    ref = crop_feature_torch(feat.expand(5, -1, -1, -1), locs, oris, 2.0, 96, 0.0, 0.75)
    out = ops.crop_rotate(feat.to(DEV), locs.to(DEV), oris.to(DEV), 2.0, 96, 0.0, 0.75).cpu()
    assert_close(out.numpy(), ref.numpy(), atol=2e-4, what="rotated crop (shared map)")
    fb = rnd((5, 8, 40, 56), 32)
    ref = crop_feature_torch(fb, locs, oris, 1.0, 24, 0.1, 0.5)
    out = ops.crop_rotate(fb.to(DEV), locs.to(DEV), oris.to(DEV), 1.0, 24, 0.1, 0.5).cpu()
    assert_close(out.numpy(), ref.numpy(), atol=2e-4, what="rotated crop (per-sample maps)")


def test_crop_rotate_staged_kernel_is_bit_identical_to_the_gathering_one(monkeypatch):
    """The LDS-staged forward (16 x 16 output tiles, the touched box of the map read row by row) and the kernel that gathers its
    four corners from L2: same corners, same weights, same order of the products - same bits.  Frame-sized map (the ego crop and
    crops half off the map), a ragged channel block, a crop size that is not a multiple of the tile."""
    from lav_amd import ops
    locs = torch.tensor([[0.0, 0.0], [-5.0, -20.0], [7.5, -32.5], [38.0, 38.0], [-39.0, 3.0], [55.0, -60.0]]).to(DEV)
    oris = torch.tensor([0.0, 0.3, -1.2, 2.9, -3.1, 0.78]).to(DEV)
    for shape, crop in (((1, 40, 160, 160), 96), ((6, 8, 80, 80), 41)):
        feat = rnd(shape, 33).to(DEV)
        staged = ops.crop_rotate(feat, locs, oris, 2.0, crop, 0.0, 0.75)
        monkeypatch.setenv("LAV_CROP_FWD_GENERAL", "1")
        gathered = ops.crop_rotate(feat, locs, oris, 2.0, crop, 0.0, 0.75)
        monkeypatch.delenv("LAV_CROP_FWD_GENERAL")
        assert torch.equal(staged, gathered), f"{shape} crop {crop}: max |diff| {(staged - gathered).abs().max().item()}"
        assert float(staged.abs().max()) > 0


def test_workspace_growth_does_not_invalidate_captured_graphs():
    """A split-K layer captured in a HIP graph keeps working after a later, larger layer made the per-stream workspace
    grow: the old buffer (whose address is baked into the graph's kernel nodes) is retired, not freed."""
    from lav_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    small = ops.ConvLayer(torch.randn((256, 256, 3, 3), generator=g) * 0.02, padding=(1, 1), device=dev)
    x = torch.randn((1, 256, 6, 6), generator=g).to(dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        ref = small(x).clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            y = small(x)
        big = ops.ConvLayer(torch.randn((512, 512, 3, 3), generator=g) * 0.02, padding=(1, 1), device=dev)
        junk = [big(torch.randn((8, 512, 6, 6), generator=g).to(dev)) for _ in range(3)]      # grows the "conv" workspace on this stream
        hog = [torch.randn((1 << 20,), device=dev) for _ in range(8)]                            # would land in a freed buffer
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and len(junk) == 3 and len(hog) == 8


@pytest.mark.parametrize("ch,h,w,d,batch,with_res", [(128, 36, 32, 1, 3, True), (128, 36, 32, 2, 3, False), (128, 36, 32, 16, 2, True),
                                                        (64, 72, 64, 1, 3, True), (16, 144, 128, 1, 2, True), (128, 5, 32, 8, 1, True),
                                                        (32, 7, 128, 4, 1, False), (48, 9, 64, 3, 2, True)])
def test_conv1d_pair_matches_torch(ch, h, w, d, batch, with_res):
    """lav_conv1d_pair (conv3x1 -> ReLU -> conv1x3 -> BN -> +x -> ReLU in one launch) vs torch CPU fp32
    (lav/models/erfnet.py:45-62), incl. dilation 16 on a 36-row image (taps fall outside: zero padding) and odd heights."""
    from lav_amd import ops
    g = torch.Generator().manual_seed(ch * 1000 + w + d)
    ca = torch.nn.Conv2d(ch, ch, (3, 1), padding=(d, 0), dilation=(d, 1))
    cb = torch.nn.Conv2d(ch, ch, (1, 3), padding=(0, d), dilation=(1, d))
    bn = torch.nn.BatchNorm2d(ch, eps=1e-3).eval()
    with torch.no_grad():
        for p in list(ca.parameters()) + list(cb.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / (3 * ch) ** 0.5 if p.dim() > 1 else 0.1))
        bn.weight.copy_(torch.rand(ch, generator=g) + 0.5); bn.bias.copy_(torch.randn(ch, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(ch, generator=g) * 0.1); bn.running_var.copy_(torch.rand(ch, generator=g) + 0.5)
        x = torch.randn((batch, ch, h, w), generator=g)
        ref = bn(cb(torch.relu(ca(x))))
        ref = torch.relu(ref + x) if with_res else torch.relu(ref)
    dev = torch.device("cuda")
    pair = ops.Conv1dPair(ca, cb, bn, device=dev)
    xd = x.to(dev)
    assert pair.supported(xd)
    y = pair(xd, residual=xd if with_res else None)
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), f"max err {err}"


def test_conv1d_pair_rejects_unsupported_rows():
    from lav_amd import ops
    ca, cb = torch.nn.Conv2d(64, 64, (3, 1), padding=(1, 0)), torch.nn.Conv2d(64, 64, (1, 3), padding=(0, 1))
    pair = ops.Conv1dPair(ca, cb, None, device=torch.device("cuda"))
    assert not pair.supported(torch.zeros((1, 64, 8, 48))) and not pair.supported(torch.zeros((1, 64, 8, 128)))
    with pytest.raises(RuntimeError, match="row width"):
        pair(torch.zeros((1, 64, 8, 48), device="cuda"))


def test_pool_affine_and_folded_input_normalisation():
    """lav_pool_affine + a convolution with pad_value: ERFNet's DownsamplerBlock on RAW input with (x/255 - .5)*2 folded in
    == the torch block on the normalised input (lav/models/erfnet.py:9-23, team_code_v2/models/rgb.py:44-46)."""
    from lav_amd.erfnet import DownsamplerBlock
    g = torch.Generator().manual_seed(7)
    blk = DownsamplerBlock(3, 16).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        blk.bn.running_mean.copy_(torch.randn(16, generator=g) * 0.1); blk.bn.running_var.copy_(torch.rand(16, generator=g) + 0.5)
        raw = torch.randint(0, 256, (2, 3, 40, 36), generator=g).float()
        ref = blk((raw / 255. - .5) * 2)
    run = blk.engine(torch.device("cuda"), input_affine=(2.0 / 255.0, -1.0))
    out = run(raw.cuda()).cpu()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))
    plain = blk.engine(torch.device("cuda"))(((raw / 255. - .5) * 2).cuda()).cpu()      # no folding: same block, same answer
    np.testing.assert_allclose(plain.numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


DIRECT_CASES = [
    # name, batch, cin, cout, k, stride, pad, dil, H, W
    ("layer1 block 64->64 24x24 x7", 7, 64, 64, 3, 1, 1, 1, 24, 24),
    ("layer2 down 64->128 s2 x3", 3, 64, 128, 3, 2, 1, 1, 24, 24),
    ("layer2 1x1 s2 x1", 1, 64, 128, 1, 2, 0, 1, 24, 24),
    ("layer3 block 256->256 6x6 x1", 1, 256, 256, 3, 1, 1, 1, 6, 6),
    ("layer4 block 512->512 3x3 x7", 7, 512, 512, 3, 1, 1, 1, 3, 3),
    ("layer4 block 512->512 3x3 x1", 1, 512, 512, 3, 1, 1, 1, 3, 3),
    ("ragged: 96->40 5x7 map dil 2 x5", 5, 96, 40, 3, 1, 2, 2, 5, 7),
]


@pytest.mark.parametrize("case", DIRECT_CASES, ids=[c[0] for c in DIRECT_CASES])
def test_direct_path_small_layers(case):
    """Small maps take the direct kernel (pixels of the whole batch in one GEMM dimension, operands straight from L2):
    full epilogue (bias, BN, residual, ReLU), channel windows on both sides and a non-zero pad value."""
    import ctypes as C
    from lav_amd import _lib
    name, B, cin, cout, k, s, p, d, H, W = case
    x = rnd((B, cin + 16, H, W), 21)
    w = rnd((cout, cin, k, k), 22, scale=1.0 / np.sqrt(cin * k * k))
    bias = rnd((cout,), 23)
    bn = (rnd((cout,), 24, 0.1), rnd((cout,), 25).abs() + 0.5, rnd((cout,), 26).abs() + 0.5, rnd((cout,), 27, 0.1))
    pad_value = 0.25
    layer = ConvLayer(w, stride=s, padding=p, dilation=d, bias=bias, bn=bn, relu_post=True, in_c_total=cin + 16, in_c_offset=8,
                      out_c_total=cout + 8, out_c_offset=8, pad_value=pad_value, device=DEV)
    desc = _lib.Conv.from_buffer_copy(layer.desc)
    desc.batch, desc.h, desc.w = B, H, W
    info = (C.c_int * 9)()
    assert _lib.load().lav_conv_tile_info(C.byref(desc), info) == 0
    assert info[0] == 0, f"{name}: expected the direct plan, got tile {info[0]}x{info[1]}"
    xin = F.pad(x[:, 8:8 + cin], (p, p, p, p), value=pad_value)
    pre = F.batch_norm(F.conv2d(xin, w, bias, s, 0, d), bn[0], bn[1], bn[2], bn[3], False, 0., 1e-5)
    res = rnd((B, cout + 8, *pre.shape[2:]), 28)
    ref = F.relu(pre + res[:, 8:])
    out = torch.full((B, cout + 8, *pre.shape[2:]), -3.0, device=DEV)
    layer(x.to(DEV), out=out, residual=res.to(DEV))
    out = out.cpu()
    assert_close(out[:, 8:].numpy(), ref.numpy(), atol=3e-5, rtol=1e-5, what=name)
    assert (out[:, :8] == -3).all(), "channels outside the output window were touched"


@pytest.mark.parametrize("outs,cin_g,k,pad,opad,B,H,W,sig", [((1, 2, 2, 4), 64, 3, 1, 1, 1, 40, 56, True), ((5,), 16, 2, 0, 0, 3, 36, 32, False),
                                                           ((3, 8), 32, 3, 1, 1, 2, 17, 9, False), ((2,), 48, 3, 1, 0, 1, 8, 300, True)])
def test_grouped_deconv_matches_torch(outs, cin_g, k, pad, opad, B, H, W, sig):
    """lav_deconv_grouped (heads' tails, ERFNet output layer) against torch's ConvTranspose2d per group."""
    from lav_amd.ops import GroupedDeconv
    torch.manual_seed(5)
    cts = [torch.nn.ConvTranspose2d(cin_g, o, k, stride=2, padding=pad, output_padding=opad) for o in outs]
    x = rnd((B, cin_g * len(outs), H, W), 31)
    with torch.no_grad():
        ref = torch.cat([ct(x[:, i * cin_g:(i + 1) * cin_g]) for i, ct in enumerate(cts)], dim=1)
        sf = sum(outs[:-1]) if sig else -1
        if sig:
            ref[:, sf:] = torch.sigmoid(ref[:, sf:])
    y = GroupedDeconv(cts, sigmoid_from=sf, device=DEV)(x.to(DEV)).cpu()
    assert y.shape == ref.shape
    assert_close(y.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what=f"grouped deconv {outs}")


@pytest.mark.parametrize("outs,cin_g,opad,B,H,W", [((2, 2, 2, 3), 64, 1, 1, 160, 160), ((1, 2, 2, 4), 40, 1, 2, 37, 44), ((2,), 48, 0, 1, 8, 300), ((3, 3), 16, 1, 3, 5, 4)])
def test_staged_grouped_deconv_is_bit_identical_to_the_gathering_kernel(outs, cin_g, opad, B, H, W, monkeypatch):
    """Round 5: k_deconv_tile (3x3 stride 2: tiles of 8 x 32 input-grid positions staged through LDS 16 channels at a time) against
    k_deconv_grouped (every thread gathers its four pixels per channel from L2): same products in the same order, so the same bits -
    the heads' own shape, ragged channel chunks / tiles / batch, output_padding 0, a map narrower than a tile."""
    from lav_amd.ops import GroupedDeconv
    torch.manual_seed(7)
    cts = [torch.nn.ConvTranspose2d(cin_g, o, 3, stride=2, padding=1, output_padding=opad) for o in outs]
    x = rnd((B, cin_g * len(outs), H, W), 33).to(DEV)
    layer = GroupedDeconv(cts, sigmoid_from=sum(outs[:-1]), device=DEV)
    staged = layer(x).clone()
    monkeypatch.setenv("LAV_DECONV_IMPL", "gather")
    gathered = layer(x).clone()
    monkeypatch.delenv("LAV_DECONV_IMPL")
    assert torch.equal(staged, gathered), f"max |diff| {float((staged - gathered).abs().max())}"
    with torch.no_grad():
        ref = torch.cat([ct(x.cpu()[:, i * cin_g:(i + 1) * cin_g]) for i, ct in enumerate(cts)], dim=1)
        ref[:, sum(outs[:-1]):] = torch.sigmoid(ref[:, sum(outs[:-1]):])
    assert_close(staged.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what=f"staged grouped deconv {outs}")


@pytest.mark.parametrize("ch,h,w,dils", [(128, 36, 32, (2, 4, 8, 16)), (64, 72, 64, (1, 1, 1)), (64, 20, 64, (1, 3)), (16, 144, 128, (1, 1))])
def test_pair_chain_is_bit_identical_to_the_pairs_launched_one_by_one(ch, h, w, dils):
    """lav_conv1d_pair_chain (one persistent launch for a run of non_bottleneck_1d blocks, rows handed between workgroups through
    write-through stores and per-row counters) against the same pairs as separate lav_conv1d_pair launches: same arithmetic in
    the same order, so the results must agree bit for bit - on every row, whatever the dilations; no workgroup may have timed
    out.  Repeated launches (stale counters, stale buffers of the previous launch) must give the same bits again."""
    import torch.nn as nn
    from lav_amd.ops import Conv1dPair, Conv1dPairChain
    torch.manual_seed(ch + h)
    B = 3
    pairs, res = [], []
    for d in dils:
        for dd in (1, d):   # a non_bottleneck_1d block: an undilated pair, then the dilated one with the residual
            ca = nn.Conv2d(ch, ch, (3, 1), padding=(dd, 0), dilation=(dd, 1))
            cb = nn.Conv2d(ch, ch, (1, 3), padding=(0, dd), dilation=(1, dd))
            bn = nn.BatchNorm2d(ch, eps=1e-3)
            with torch.no_grad():
                bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            pairs.append(Conv1dPair(ca, cb, bn, device=DEV))
            res.append(len(pairs) % 2 == 0)
    chain = Conv1dPairChain(pairs, res)
    x = torch.randn((B, ch, h, w), device=DEV)
    assert chain.supported(x)
    want = x
    for i in range(0, len(pairs), 2):
        want = pairs[i + 1](pairs[i](want), residual=want)
    got = chain(x)
    torch.cuda.synchronize()
    assert chain.timeouts(x) == 0
    assert torch.equal(got, want), f"max |diff| {(got - want).abs().max().item():.3e}"
    for _ in range(3):
        again = chain(x)
    torch.cuda.synchronize()
    assert chain.timeouts(x) == 0 and torch.equal(again, want)
    # ... and beside a stream that keeps every CU busy with the tap-pair 7x7 stem kernel (150 KB of LDS per workgroup, waves that
    # share SIMDs with other kernels' - the load that broke round 4's quarter-poll plan kernel, tools/plan_stress.py)
    from lav_amd import _lib
    hog = ConvLayer(torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5, stride=2, padding=(3, 3), relu_post=True, precision=_lib.CONV_BF16X6, device=DEV)
    hog_x = torch.randn(15, 384, 96, 96, device=DEV)
    s_hog, s_chain = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s_hog):
        hog(hog_x)
    torch.cuda.synchronize()
    outs = []
    for _ in range(30):
        with torch.cuda.stream(s_hog):
            hog(hog_x)
        with torch.cuda.stream(s_chain):
            outs.append(chain(x).clone())
    torch.cuda.synchronize()
    with torch.cuda.stream(s_chain):
        assert chain.timeouts(x) == 0
    assert all(torch.equal(o, want) for o in outs), "the persistent run differs from the single launches beside the stem kernel"


@pytest.mark.parametrize("ch,h,w,dils", [(128, 36, 32, (2, 4, 8, 16)), (64, 72, 64, (1, 1, 1)), (64, 20, 64, (1, 3)), (16, 144, 128, (1, 1))])
def test_pair_chain_on_fp16_pieces_keeps_the_fp32_dot_products_error(ch, h, w, dils):
    """lav_conv1d_pair_chain_f16 (round 6: the persistent run on two scaled fp16 pieces per operand and three products, the activation
    scale derived per workgroup and pair from its own row's maximum and the neighbours' maxima that travel with the hand-off, the
    intermediate row's from a bound) against float64: no further from it than a small multiple of the bf16x6 run (which holds the error
    of an fp32 dot product), on inputs whose ROWS differ by six decades (a scale taken from the wrong rows overflows fp16 or drops the
    small rows' bits), with an all-zero image in the batch, repeatedly, and deterministic from launch to launch."""
    import torch.nn as nn
    from lav_amd import _lib, ops
    from lav_amd.ops import Conv1dPair, Conv1dPairChain
    torch.manual_seed(ch + h + 1)
    B = 3
    mods, pairs, res = [], [], []
    for d in dils:
        for dd in (1, d):
            ca = nn.Conv2d(ch, ch, (3, 1), padding=(dd, 0), dilation=(dd, 1))
            cb = nn.Conv2d(ch, ch, (1, 3), padding=(0, dd), dilation=(1, dd))
            bn = nn.BatchNorm2d(ch, eps=1e-3).eval()
            with torch.no_grad():
                bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            mods.append((ca, cb, bn))
            pairs.append(Conv1dPair(ca, cb, bn, device=DEV))
            res.append(len(pairs) % 2 == 0)
    with ops.precision(_lib.CONV_F16X3):
        chain16 = Conv1dPairChain(pairs, res)
    with ops.precision(_lib.CONV_BF16X6):
        chain_bf = Conv1dPairChain(pairs, res)
    assert chain16.f16 and not chain_bf.f16

    def ref64(x):
        y = x.double().cpu()
        with torch.no_grad():
            for i in range(0, len(mods), 2):
                blk = y
                for k in (i, i + 1):
                    ca, cb, bn = (m.double() for m in mods[k])
                    y = bn(cb(torch.relu(ca(y))))
                    if k == i:
                        y = torch.relu(y)
                y = torch.relu(y + blk)
        for trip in mods:
            for m in trip:
                m.float()
        return y

    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, ch, h, w), generator=g)
    cases = {"plain": x.clone()}
    rows = x.clone()
    rows *= (10.0 ** torch.linspace(-3, 3, h))[None, None, :, None]     # six decades from the top row to the bottom one
    rows[1] = 0.0                                                       # an all-zero image: maxima of zero
    cases["rows over six decades"] = rows
    cases["tiny"] = x * 1e-20
    for name, xin in cases.items():
        xd = xin.to(DEV)
        assert chain16.supported(xd)
        want = ref64(xin)
        got16 = chain16(xd)
        got_bf = chain_bf(xd)
        torch.cuda.synchronize()
        assert chain16.timeouts(xd) == 0
        assert bool(torch.isfinite(got16).all()), name
        # per image row (the rows' magnitudes differ): error against the row's own largest value
        top = want.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-300)
        e16 = ((got16.double().cpu() - want).abs() / top).max().item()
        ebf = ((got_bf.double().cpu() - want).abs() / top).max().item()
        assert e16 <= max(4 * ebf, 4e-6), f"{name}: fp16-piece run {e16:.3e} of the row maximum from float64, bf16x6 run {ebf:.3e}"
        for _ in range(3):
            again = chain16(xd)
        torch.cuda.synchronize()
        assert torch.equal(again, got16), name


@pytest.mark.parametrize("f16", [False, True])
def test_pair_chain_timeout_is_counted_and_never_hangs(monkeypatch, f16):
    """A workgroup of the persistent run that waits too long for a neighbour row (forced: a spin limit of 0) gives up instead of
    hanging the device, and the sticky counter of the workspace says that the result is void; the next launch is valid again."""
    import torch.nn as nn
    from lav_amd import ops
    from lav_amd.ops import Conv1dPair, Conv1dPairChain
    torch.manual_seed(5)
    ch, h, w = 128, 36, 32
    pairs = []
    for d in (1, 2, 1, 4, 1, 8):
        pairs.append(Conv1dPair(nn.Conv2d(ch, ch, (3, 1), padding=(d, 0), dilation=(d, 1)), nn.Conv2d(ch, ch, (1, 3), padding=(0, d), dilation=(1, d)),
                                nn.BatchNorm2d(ch, eps=1e-3).eval(), device=DEV))
    from lav_amd import _lib
    with ops.precision(_lib.CONV_F16X3 if f16 else _lib.CONV_BF16X6):
        chain = Conv1dPairChain(pairs, [i % 2 == 1 for i in range(len(pairs))])
    assert chain.f16 == f16
    x = torch.randn((3, ch, h, w), device=DEV)
    good = chain(x)
    torch.cuda.synchronize()
    t0, n0 = ops.pair_chain_status(DEV)
    monkeypatch.setenv("LAV_CHAIN_SPIN_LIMIT", "0")
    void = chain(x)
    torch.cuda.synchronize()          # (returns: nobody waits forever)
    monkeypatch.delenv("LAV_CHAIN_SPIN_LIMIT")
    t1, n1 = ops.pair_chain_status(DEV)
    assert n1 == n0 + 1 and t1 > t0, "108 rows never finish a pair at the same instant: some workgroup must have given up"
    # round 5: a row that gives up (or sees that another one did) stops computing and voids its row of the result - the
    # output holds rows that completed every pair (bit-identical to the good run) and NaN rows, nothing computed from stale neighbours
    nan_rows = torch.isnan(void).all(dim=3).all(dim=1)                     # (batch, h)
    same_rows = (void == good).all(dim=3).all(dim=1)
    assert bool((nan_rows | same_rows).all()), "every row is either complete and right, or void"
    assert bool(nan_rows.any()), "an aborted persistent run must not look like a result"
    back = chain(x)
    torch.cuda.synchronize()
    t2, n2 = ops.pair_chain_status(DEV)
    assert (t2, n2) == (t1, n1 + 1) and torch.equal(back, good)


TAP_PAIR_CASES = [
    # name, B, cin, cout, k, stride, pad, H, W, LAV_SPLIT_FORCE (MP,MC,WPX,tw,tg,ks,ring,tp)
    ("7x7 s2 stem 1x2 tile, ragged tiles", 3, 32, 64, 7, 2, 3, 50, 46, "1,2,4,16,4,0,0,1"),
    ("7x7 s2 stem 2x2 tile", 2, 64, 64, 7, 2, 3, 96, 96, "2,2,4,16,1,0,0,1"),
    ("7x7 s2 stem, split-K 3", 1, 96, 64, 7, 2, 3, 96, 96, "1,2,4,16,4,3,0,1"),
    ("5x5 s1 (13 pairs, odd tap out), ragged couts", 2, 48, 40, 5, 1, 2, 20, 36, "1,2,4,16,4,0,0,1"),
    ("6x6 s2 (even tap count), split-K 2", 2, 32, 64, 6, 2, 2, 40, 40, "1,2,4,16,4,2,0,1"),
    ("7x7 s2 stem at a training-size batch of 20 crops", 20, 32, 64, 7, 2, 3, 48, 48, "1,2,4,16,4,0,0,1"),
]


@pytest.mark.parametrize("case", TAP_PAIR_CASES, ids=[c[0] for c in TAP_PAIR_CASES])
def test_split_kernel_tap_pair_mode(case, monkeypatch):
    """k_conv_split<..., TP>: 8-channel chunks, two taps per matrix instruction (the 7x7 stride-2 crop stems).  The plan is
    pinned and checked (tap-pair mode reports tap group + 100), the result held to the fp32 reference with bias, eval-mode
    BatchNorm, residual and ReLU in the epilogue, and to the exact-fp32 kernel of this library."""
    import ctypes
    from lav_amd import _lib
    name, B, cin, cout, k, s, p, H, W, force = case
    monkeypatch.setenv("LAV_CONV_SPLIT", "2")
    monkeypatch.setenv("LAV_SPLIT_FORCE", force)
    x = rnd((B, cin, H, W), 31)
    w = rnd((cout, cin, k, k), 32, scale=1.0 / np.sqrt(cin * k * k))
    bias = rnd((cout,), 33)
    bn = (rnd((cout,), 34, 0.1), rnd((cout,), 35).abs() + 0.5, rnd((cout,), 36).abs() + 0.5, rnd((cout,), 37, 0.1))
    conv = F.conv2d(x, w, bias, s, p)
    res = rnd(tuple(conv.shape), 38)
    ref = F.relu(F.batch_norm(conv, bn[0], bn[1], bn[2], bn[3], False, 0., 1e-5) + res)
    kw = dict(stride=s, padding=(p, p), bias=bias, bn=bn, relu_post=True, device=DEV)
    layer = ConvLayer(w, precision=_lib.CONV_BF16X6, **kw)
    d = _lib.Conv.from_buffer_copy(layer.desc); d.batch, d.h, d.w = B, H, W
    info = (ctypes.c_int * 9)()
    assert _lib.load().lav_conv_tile_info(ctypes.byref(d), info) == 0
    assert info[0] == -1 and info[7] >= 100, f"expected a tap-pair split plan, got {list(info)}"
    y = layer(x.to(DEV), residual=res.to(DEV)).cpu()
    assert_close(y.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5, what=name)
    monkeypatch.delenv("LAV_SPLIT_FORCE")
    exact = ConvLayer(w, precision=_lib.CONV_F32, **kw)(x.to(DEV), residual=res.to(DEV)).cpu()
    assert_close(y.numpy(), exact.numpy(), atol=3e-5, rtol=1e-5, what=name + " vs fp32 kernel")   # (two fp32 summation orders: 2.1e-5 in 2 of 737 280 outputs of the 20-crop case)


def test_tap_pair_stem_respects_the_batch_limit(monkeypatch):
    """The others branch runs its stem at capacity 15 with a device-resident row limit: rows >= N keep their old contents,
    rows < N are bit-identical to a batch-N launch."""
    import ctypes
    from lav_amd import _lib, ops
    monkeypatch.setenv("LAV_CONV_SPLIT", "2")
    monkeypatch.setenv("LAV_SPLIT_FORCE", "1,2,4,16,4,2,0,1")   # split-K 2: the reduce launch honours the limit as well
    B, N, cin = 5, 2, 32
    x = rnd((B, cin, 48, 48), 41)
    w = rnd((64, cin, 7, 7), 42, scale=1.0 / np.sqrt(cin * 49))
    layer = ConvLayer(w, stride=2, padding=(3, 3), relu_post=True, precision=_lib.CONV_BF16X6, device=DEV)
    d = _lib.Conv.from_buffer_copy(layer.desc); d.batch, d.h, d.w = B, 48, 48
    info = (ctypes.c_int * 9)()
    assert _lib.load().lav_conv_tile_info(ctypes.byref(d), info) == 0
    assert info[0] == -1 and info[7] >= 100, f"expected a tap-pair split plan, got {list(info)}"
    full = layer(x.to(DEV)).clone()
    out = torch.full_like(full, 7.0)
    n_dev = torch.tensor([N], dtype=torch.int32, device=DEV)
    with ops.batch_limit(n_dev):
        layer(x.to(DEV), out=out)
    torch.cuda.synchronize()
    assert torch.equal(out[:N], full[:N]) and (out[N:] == 7).all()


F16_CASES = [
    # name, B, cin, cout, k, stride, pad, transposed, out_pad, H, W, LAV_SPLIT_FORCE ("" = the plan search)
    ("BEV 64->64 s1 160x160", 1, 64, 64, 3, 1, 1, False, 0, 160, 160, ""),
    ("BEV 64->128 s2", 1, 64, 128, 3, 2, 1, False, 0, 160, 160, ""),
    ("BEV 128->128 80x80", 1, 128, 128, 3, 1, 1, False, 0, 80, 80, ""),
    ("brake 512->512 9x24 (split-K)", 1, 512, 512, 3, 1, 1, False, 0, 9, 24, ""),
    ("up-convolution 4x4 s2 (four parity classes)", 1, 128, 128, 4, 2, 1, True, 0, 80, 80, ""),
    ("up-convolution 4x4 s4 op2 (16 classes, one tap each)", 1, 128, 128, 4, 4, 1, True, 2, 40, 40, ""),
    ("ragged channels 48->40, 2x2/w4 tile", 2, 48, 40, 3, 1, 1, False, 0, 37, 51, "2,2,4,-1,0,0,0,0"),
    ("1x2/w2 tile, two taps per barrier, split-K 2", 1, 96, 64, 3, 1, 1, False, 0, 40, 56, "1,2,2,-1,2,2,0,0"),
    ("7x7 s2 crop stem, tap pairs, 3 crops", 3, 384, 64, 7, 2, 3, False, 0, 96, 96, ""),
    ("7x7 s2 stem 2x2 tile, tap pairs", 2, 64, 64, 7, 2, 3, False, 0, 96, 96, "2,2,4,16,1,0,0,1"),
    ("5x5 s1 tap pairs with the odd tap out, split-K 2", 2, 48, 40, 5, 1, 2, False, 0, 20, 36, "1,2,4,16,4,2,0,1"),
]


@pytest.mark.parametrize("case", F16_CASES, ids=[c[0] for c in F16_CASES])
def test_f16x3_on_every_split_plan(case, monkeypatch):
    """LAV_CONV_F16X3, round 6: every plan of the split kernel (strides, tiles, tap groups, split-K, tap pairs, the parity classes
    of a transposed convolution) on two fp16 pieces and three products.  Inputs span five decades of magnitude; against float64
    the error stays below 2e-6 of sum |w||x| - the bar the bf16x6 kernel is held to - through bias, ReLU and BatchNorm; the
    maxima handed in by a producer give the same bits as the measuring launch; the maxima the launch leaves are those of its output."""
    import ctypes
    from lav_amd import _lib, ops
    name, B, cin, cout, k, s_, p_, tr, op, H, W, force = case
    if force:
        monkeypatch.setenv("LAV_CONV_SPLIT", "2")
        monkeypatch.setenv("LAV_SPLIT_FORCE", force)
    g = np.random.Generator(np.random.PCG64(101))
    x = torch.from_numpy((g.standard_normal((B, cin, H, W)) * np.exp(g.uniform(-8.0, 3.0, (B, cin, H, W)))).astype(np.float32))
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.from_numpy((g.standard_normal(wshape) * np.exp(g.uniform(-4.0, 1.0, wshape)) / np.sqrt(cin * k * k)).astype(np.float32))
    bias = rnd((cout,), 103)
    bn = (rnd((cout,), 104, 0.1), rnd((cout,), 105).abs() + 0.5, rnd((cout,), 106).abs() + 0.5, rnd((cout,), 107, 0.1))
    xd, wd = x.to(DEV).double(), w.to(DEV).double()
    conv = (lambda a, b: F.conv_transpose2d(a, b, None, s_, p_, op)) if tr else (lambda a, b: F.conv2d(a, b, None, s_, p_))
    conv64, mag = conv(xd, wd), conv(xd.abs(), wd.abs())
    kw = dict(stride=s_, padding=(p_, p_), transposed=tr, output_padding=op, device=DEV)
    plain = ConvLayer(w, precision=_lib.CONV_F16X3, **kw)
    d = _lib.Conv.from_buffer_copy(plain.desc); d.batch, d.h, d.w = B, H, W
    info = (ctypes.c_int * 9)()
    assert _lib.load().lav_conv_tile_info(ctypes.byref(d), info) == 0
    assert info[0] == -1 and info[7] >= 200, f"expected an fp16 split plan, got {list(info)}"
    assert plain.uses_amax(B, H, W)
    y = plain(x.to(DEV))
    err = ((y.double() - conv64).abs() / mag.clamp_min(1e-30)).max().item()
    assert err < 2e-6, f"{name}: max |y - ref| / sum|w||x| = {err:.3e}"
    assert torch.equal(y, plain(x.to(DEV))), "bit-reproducible"
    # a producer's maxima (here: any parts whose maximum is max |x|) instead of the measuring launch: the same scale, the same bits
    am_in = ops.Amax(DEV)
    parts = am_in.take(7)
    parts.zero_(); parts[3] = x.abs().max().item(); parts[5] = 1e-3
    am_out = ops.Amax(DEV)
    y2 = plain(x.to(DEV), amax_in=am_in, amax_out=am_out)
    assert torch.equal(y, y2), "handed-in maxima must give the measuring launch's bits"
    assert am_out.count >= 1 and am_out.buf[:am_out.count].max().item() == y.abs().max().item(), "amax_out: the largest |y|"
    assert (am_out.buf[am_out.count:] == 0).all()
    # epilogue: bias -> ReLU -> BatchNorm affine, into a channel window
    layer = ConvLayer(w, bias=bias, bn=bn, bn_eps=1e-3, relu_pre=True, out_c_total=cout + 8, out_c_offset=8, precision=_lib.CONV_F16X3, **kw)
    out = torch.full((B, cout + 8) + tuple(y.shape[2:]), 7.0, device=DEV)
    am_out.reset()
    layer(x.to(DEV), out=out, amax_out=am_out)
    assert (out[:, :8] == 7).all()
    ref = F.batch_norm(F.relu(conv64 + bias.to(DEV).double()[None, :, None, None]), bn[0].to(DEV).double(), bn[1].to(DEV).double(),
                       bn[2].to(DEV).double(), bn[3].to(DEV).double(), False, 0., 1e-3)
    gain = (bn[2].abs() / torch.sqrt(bn[1] + 1e-3)).to(DEV).double()[None, :, None, None]
    excess = ((out[:, 8:].double() - ref).abs() - (2e-6 * mag * gain + 1e-5 * ref.abs() + 1e-6)).max().item()
    assert excess <= 0, f"{name}: epilogue beyond 2e-6 of sum|w||x| (through the BatchNorm gain) by {excess:.3e}"
    assert am_out.buf[:am_out.count].max().item() == out[:, 8:].abs().max().item()


def test_amax_hand_off_from_every_kernel_family(monkeypatch):
    """lav_conv2d_amax: whichever kernel runs the producing layer - split (whole K / split-K), direct (whole K / split-K), tiled
    fp32, small-cin vector kernel - the maxima it leaves are those of its output, rows beyond lav_batch_limit leave zeros, and a
    LAV_CONV_F16X3 consumer fed from them computes what it computes from a measurement of its own."""
    from lav_amd import _lib, ops
    shapes = [  # cin, cout, k, stride, H, W, B, precision
        (64, 64, 3, 1, 160, 160, 1, _lib.CONV_BF16X6),     # split, whole K
        (512, 512, 3, 1, 9, 24, 1, _lib.CONV_BF16X6),      # split, split-K
        (128, 128, 3, 1, 40, 40, 1, _lib.CONV_BF16X6),     # direct
        (256, 256, 3, 1, 6, 6, 4, _lib.CONV_BF16X6),       # direct, split-K
        (64, 64, 3, 1, 72, 192, 1, _lib.CONV_F32),         # tiled fp32
        (3, 64, 7, 2, 96, 160, 1, 0),                      # small-cin vector kernel
    ]
    for i, (cin, cout, k, s_, H, W, B, prec) in enumerate(shapes):
        x = rnd((B, cin, H, W), 200 + i).to(DEV)
        w = rnd((cout, cin, k, k), 300 + i, scale=1.0 / np.sqrt(cin * k * k))
        layer = ConvLayer(w, stride=s_, padding=(k // 2, k // 2), relu_post=True, precision=prec, device=DEV)
        am = ops.Amax(DEV)
        y = layer(x, amax_out=am)
        assert am.count >= 1
        assert am.buf[:am.count].max().item() == y.abs().max().item(), f"shape {i}: maxima of |y|"
        nxt = ConvLayer(rnd((32, cout, 3, 3), 400 + i, scale=0.05), padding=(1, 1), precision=_lib.CONV_F16X3, device=DEV)
        if nxt.uses_amax(B, y.shape[2], y.shape[3]):
            assert torch.equal(nxt(y, amax_in=am), nxt(y)), f"shape {i}: consumer on handed-in maxima"
    # batch limit: the dead rows of a capacity-sized launch contribute nothing
    x = rnd((6, 64, 24, 24), 501).to(DEV)
    x[4:] *= 1e4
    layer = ConvLayer(rnd((64, 64, 3, 3), 502, scale=0.05), padding=(1, 1), relu_post=True, device=DEV)
    n_dev = torch.tensor([3], dtype=torch.int32, device=DEV)
    am = ops.Amax(DEV)
    out = torch.zeros((6, 64, 24, 24), device=DEV)
    with ops.batch_limit(n_dev):
        layer(x, out=out, amax_out=am)
    assert am.buf[:am.count].max().item() == out[:3].abs().max().item()


@pytest.mark.parametrize("cin,cout,k,pad,opad,h,w,B", [(64, 128, 1, 0, 0, 160, 160, 1), (128, 128, 4, 1, 2, 40, 40, 1), (64, 96, 4, 0, 0, 9, 13, 2),
                                                       (128, 256, 1, 0, 0, 7, 50, 3), (128, 32, 4, 3, 1, 5, 5, 1), (64, 64, 4, 2, 3, 11, 6, 2)])
def test_pointwise_upconv_matches_torch(cin, cout, k, pad, opad, h, w, B):
    """lav_upconv_pointwise (round 6: transposed convolutions with kernel == stride as k*k independent exact-fp32 matrix products - the
    BEV backbone's 1x1 and 4x4 / stride-4 up-convolutions, team_code_v2/models/lidar.py:114-131) against ConvTranspose2d -> ReLU ->
    eval BatchNorm in float64: the error of an fp32 dot product; the channel slice of a wider map is written and nothing else; the
    parts it leaves bound what it wrote (and equal its largest magnitude); the host packer is the device permutation; refresh()
    follows in-place parameter changes."""
    import torch.nn as nn
    from lav_amd import _lib, ops
    from lav_amd.ops import Amax, PointwiseUpconv
    import ctypes as C
    torch.manual_seed(cin + cout + k + h)
    ct = nn.ConvTranspose2d(cin, cout, k, k, pad, opad, bias=False).to(DEV)
    bn = nn.BatchNorm2d(cout, eps=1e-3).to(DEV).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    assert PointwiseUpconv.takes(ct)
    total, off = cout + 48, 16
    layer = PointwiseUpconv(ct, bn, relu_pre=True, out_c_total=total, out_c_offset=off, device=DEV)
    x = torch.randn((B, cin, h, w), device=DEV)

    def ref():
        with torch.no_grad():
            y = F.conv_transpose2d(x.double().cpu(), ct.weight.double().cpu(), None, k, pad, opad)
            return F.batch_norm(torch.relu(y), bn.running_mean.double().cpu(), bn.running_var.double().cpu(), bn.weight.double().cpu(),
                                bn.bias.double().cpu(), False, 0.0, bn.eps)
    want = ref()
    oh, ow = layer.out_hw(h, w)
    assert tuple(want.shape) == (B, cout, oh, ow)
    out = torch.full((B, total, oh, ow), -7.0, device=DEV)
    am = Amax(DEV)
    layer(x, out=out, amax_out=am)
    torch.cuda.synchronize()
    got = out[:, off:off + cout].double().cpu()
    # error of an fp32 dot product: |sum w x| accumulated in fp32 over cin terms
    with torch.no_grad():
        mag = F.conv_transpose2d(x.double().abs().cpu(), ct.weight.double().abs().cpu(), None, k, pad, opad) * (bn.weight.double().cpu() / torch.sqrt(bn.running_var.double().cpu() + bn.eps)).abs()[None, :, None, None]
    err = (got - want).abs()
    assert bool((err <= 2e-6 * mag + 1e-6 * want.abs() + 1e-7).all()), f"max err {err.max().item():.3e}"
    assert bool((out[:, :off] == -7.0).all()) and bool((out[:, off + cout:] == -7.0).all()), "wrote outside its channel slice"
    assert am.count == _lib.load().lav_upconv_pointwise_parts(B, cin, cout, h, w, k, pad, opad) > 0
    assert float(am.buf[:am.count].max()) == float(out[:, off:off + cout].abs().max())
    # host packer == the device permutation
    lib = _lib.load()
    n = lib.lav_upconv_pointwise_packed_floats(cin, cout, k)
    assert n == layer.w.numel()
    hw = ct.weight.detach().cpu().contiguous()
    packed = torch.empty(n, dtype=torch.float32)
    assert lib.lav_upconv_pointwise_pack(cin, cout, k, hw.data_ptr(), packed.data_ptr()) == 0
    assert torch.equal(packed, layer.w.cpu())
    # in-place parameter change -> refresh() -> the same buffers, new values
    ptrs = (layer.w.data_ptr(), layer.scale.data_ptr())
    with torch.no_grad():
        ct.weight.mul_(0.5); bn.bias.add_(0.25)
    layer.refresh()
    assert ptrs == (layer.w.data_ptr(), layer.scale.data_ptr())
    layer(x, out=out)
    torch.cuda.synchronize()
    want2 = ref()
    assert (out[:, off:off + cout].double().cpu() - want2).abs().max().item() < 1e-4 * max(1.0, want2.abs().max().item())


def test_backbone_takes_the_pointwise_upconvs_only_for_the_fp16_engines():
    """ConvBackbone's engine: at LAV_CONV_F16X3 (the inference pipelines) upconv1 / upconv3 run on lav_upconv_pointwise, upconv2 (4x4 stride 2:
    four taps per output pixel) and every other precision stay on lav_conv2d; both engines agree to the error of an fp32 dot product and
    the feature map's bound covers the pointwise layers' slices."""
    from lav_amd import _lib, ops
    from lav_amd.ops import PointwiseUpconv
    from tests.util import build_models
    lm, _ = build_models(DEV)
    bb = lm.backbone
    x = torch.randn((1, 64, 320, 320), device=DEV).relu_()
    with ops.precision(_lib.CONV_F16X3):
        e16 = bb._engine(x.device)
        y16 = bb(x)
    kinds = [type(u).__name__ for u in e16["ups"]]
    assert kinds == ["PointwiseUpconv", "ConvLayer", "PointwiseUpconv"], kinds
    am = ops.amax_of(y16)
    assert am is not None and float(am.buf[:am.count].max()) >= float(y16.abs().max())
    with ops.precision(_lib.CONV_BF16X6):
        ebf = bb._engine(x.device)
        ybf = bb(x)
    assert all(type(u).__name__ == "ConvLayer" for u in ebf["ups"])
    torch.cuda.synchronize()
    scale = float(ybf.abs().max())
    assert float((y16 - ybf).abs().max()) < 3e-5 * scale

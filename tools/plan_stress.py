"""Does the persistent plan kernel (lav_gru_plan) ever return a finite but WRONG plan while other streams hold the chip?

Alternates two different inputs (so that a value left over from the previous launch would show), compares every launch with the
step-per-launch path (lav_gru_plan_steps) of the same input, with and without a hog stream (the others branch's 7x7 stem at
capacity 15: 150 KB of LDS per workgroup, one workgroup per CU), eager and as a replayed HIP graph.

    [LAV_PLAN_POLL=all] python tools/plan_stress.py [launches]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib, ops  # noqa: E402
from lav_amd.ops import ConvLayer  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
H, T, NC = 512, 20, 6
g = lambda *s, sc=1.0: (torch.randn(*s) * sc).to(dev)
w_ih, w_hh, b_ih, b_hh = g(3 * H, 4, sc=0.3), g(3 * H, H, sc=H ** -0.5), g(3 * H, sc=0.1), g(3 * H, sc=0.1)
mlp_w, mlp_b = g(2, H, sc=0.05), g(2, sc=0.1)
inputs = [(g(1, H, sc=0.5), g(1, 2, sc=3.0), g(1, NC, T, 2, sc=2.0)) for _ in range(2)]


def plan(i, impl="auto"):
    e, n, c = inputs[i & 1]
    return ops.gru_plan(e, n, c, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, 5, 3, 4.0, 192.0, impl=impl)


want = [plan(0, "steps").clone(), plan(1, "steps").clone()]
torch.cuda.synchronize()
print("persistent vs steps, quiet chip:", [float((plan(i) - want[i]).abs().max()) for i in range(2)])

HOG = os.environ.get("PLAN_STRESS_HOG", "stem")   # stem: the 7x7 crop stem at capacity 15 | head: 384->256 3x3 @160x160 | fill: 400 MB fills
#                                                    synth0/1/2: tools/probes/lds_hog.hip (stem-shaped workgroups: no LDS access / LDS traffic / sleeping)
if HOG.startswith("synth"):
    import ctypes
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes")
    so = os.path.join(here, "liblds_hog.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(here, "lds_hog.hip"), "-o", so])
    hoglib = ctypes.CDLL(so)
    hoglib.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    hog_x = torch.zeros(16, device=dev)
    mode = int(HOG[5:])

    def hog(x):
        rc = hoglib.hog_launch(810, 153600, mode, 2000 if mode == 2 else 4000, x.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
elif HOG in ("erfnet", "brake"):   # the frame's own camera networks as neighbours: ERFNet (persistent pair runs, direct / split convolutions, deconvolutions) or the brake net (fp32 tiled stems, direct + split ResNet layers, attention pooling)
    from lav_amd import synth
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    if HOG == "erfnet":
        net = RGBSegmentationModel([4, 6, 7, 10]); net.load_state_dict(synth.seeded_state_dict(net, prefix="seg."))
        net = net.eval().to(dev)
        hog_x = torch.rand(3, 3, 288, 256, device=dev) * 255
        hog = lambda x: net(x)
    else:
        net = RGBBrakePredictionModel([4, 6, 7, 10]); net.load_state_dict(synth.seeded_state_dict(net, prefix="bra."))
        net = net.eval().to(dev)
        hog_x = (torch.rand(1, 3, 288, 768, device=dev) * 255, torch.rand(1, 3, 192, 480, device=dev) * 255)
        hog = lambda x: net(*x)
elif HOG in ("tiled", "erfup", "erfdown", "pool"):   # single ERFNet layers: first convolution (fp32 tiled kernel, LDS DMA), up- / down-convolution (direct kernel)
    if HOG == "tiled":
        hog = ConvLayer(torch.randn(13, 3, 3, 3) * 0.2, stride=2, padding=(1, 1), relu_post=True, device=dev)
        hog_x = torch.randn(3, 3, 288, 256, device=dev)
    elif HOG == "erfup":
        hog = ConvLayer(torch.randn(128, 64, 3, 3) * 0.03, stride=2, padding=(1, 1), transposed=True, output_padding=1, relu_post=True, device=dev)
        hog_x = torch.randn(3, 128, 36, 32, device=dev)
    elif HOG == "erfdown":
        hog = ConvLayer(torch.randn(64, 64, 3, 3) * 0.04, stride=2, padding=(1, 1), relu_post=True, device=dev)
        hog_x = torch.randn(3, 64, 72, 64, device=dev)
    else:
        hog_x = torch.randn(3, 16, 144, 128, device=dev)
        hog = lambda x: ops.maxpool3x3s2(x)
    _hog1 = hog
    hog = lambda x: [_hog1(x) for _ in range(8)]
elif HOG == "head":
    hog_w = torch.randn(256, 384, 3, 3) / (384 * 9) ** 0.5
    hog = ConvLayer(hog_w, stride=1, padding=(1, 1), relu_post=True, precision=_lib.CONV_BF16X6, device=dev)
    hog_x = torch.randn(1, 384, 160, 160, device=dev)
elif HOG == "fill":
    hog_x = torch.empty(100 * 1024 * 1024, device=dev)
    hog = lambda x: x.fill_(1.0)
else:
    hog_w = torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5
    hog = ConvLayer(hog_w, stride=2, padding=(3, 3), relu_post=True, precision=_lib.CONV_BF16X6, device=dev)
    hog_x = torch.randn(15, 384, 96, 96, device=dev)
s_hog, s_plan = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s_hog):
    hog(hog_x)
torch.cuda.synchronize()


def run(label, with_hog, graph):
    outs = []
    graphs = None
    if graph:
        graphs = []
        for i in range(2):
            with torch.cuda.stream(s_plan):
                plan(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s_plan):
                o = plan(i)
            graphs.append((gr, o))
        torch.cuda.synchronize()
    for i in range(N):
        if with_hog:
            with torch.cuda.stream(s_hog):
                hog(hog_x)
        with torch.cuda.stream(s_plan):
            if graph:
                graphs[i & 1][0].replay()
                outs.append(graphs[i & 1][1].clone())
            else:
                outs.append(plan(i))
    torch.cuda.synchronize()
    dev_ = [float((o - want[i & 1]).abs().max()) for i, o in enumerate(outs)]
    bad = [(i, round(d, 6)) for i, d in enumerate(dev_) if not d < 1e-5]
    if bad:   # where the first wrong launch goes wrong: max |diff| per iteration, first (iteration, step) above 1e-6
        i0 = bad[0][0]
        d = (outs[i0] - want[i0 & 1]).abs()[0, :, 0].max(-1).values.cpu()      # (iters, T)
        first = [(int(it), int(t)) for it in range(d.shape[0]) for t in range(d.shape[1]) if d[it, t] > 1e-6][:1]
        print("   launch", i0, "per-iteration max", [round(float(v), 6) for v in d.max(1).values], "first step off", first,
              "row of that iteration", [round(float(v), 5) for v in d[first[0][0]]] if first else None)
    diag = ops.gru_plan_diag(1, H, NC, 3, dev, stream=s_plan)
    print(f"{label:28s} launches {N}  max |diff| {max(dev_):.3e}  wrong {len(bad)}  first {bad[:6]}  aborted {diag['aborted_launches']}/{diag['launches']}", flush=True)


for graph in (False, True):
    for with_hog in (False, True):
        run(f"{'graph' if graph else 'eager'} {'+ hog stream' if with_hog else 'alone'}", with_hog, graph)

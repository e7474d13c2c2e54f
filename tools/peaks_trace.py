import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from lav_amd import ops
dev = torch.device("cuda:0")
pipe, sds, (lm, up, seg, bra) = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
for i in range(25):
    loc, ori = bench.pose(i)
    pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)
with torch.no_grad():
    heat, size, ori, _ = lm.heads(pipe.b_features)
    for _ in range(21):
        ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True)
    torch.cuda.synchronize()

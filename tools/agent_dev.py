"""Per-tick deviations of the HIP-graph agent from the reference agent's recorded run (tests/golden/agent_fast.npz): steer,
EKF pose handed to the stacking, stacked xyz, ego plan / cast waypoints.  Diagnosis companion of
tests/test_gpu_agent.py::test_agent_vs_reference_agent_golden.    [LAV_SPLIT_TP=0] python tools/agent_dev.py"""
import os
import pathlib
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import synth  # noqa: E402
from tests.test_gpu_agent import _make  # noqa: E402
from tests.util import Golden  # noqa: E402

g = Golden()["agent_fast"]
a, sc = _make(pathlib.Path(tempfile.mkdtemp()), hip_graphs=True)
ticks, npts = int(g["ticks"][0]), int(g["n_points"][0])
rows = []
for i in range(ticks):
    ctl = a.run_step(synth.agent_inputs(i, sc, n_points=npts), i * 0.05)
    if os.environ.get("AGENT_DEV_SYNC"):
        import torch
        torch.cuda.synchronize()
    want = g["controls"][i]
    if i == 0:
        continue
    out = a.last_outputs
    pose = a.pipeline.poses[-1]
    dpose = np.abs(np.r_[pose[0], pose[1]] - g["poses"][i]).max()
    dxyz = -1.0
    if f"t{i}/stacked" in g:
        pts = out["lidar_points"].cpu().numpy()
        pts = pts[~np.isnan(pts[:, 0])]
        ref = g[f"t{i}/stacked"]
        dxyz = float(np.abs(pts[:, :3] - ref[:, :3]).max()) if len(pts) == len(ref) else float("nan")
    dplan = float(np.abs(out["ego_plan_locs"].cpu().numpy() - g[f"t{i}/ego_plan"]).max())
    dcast = float(np.abs(out["ego_cast_locs"].cpu().numpy() - g[f"t{i}/ego_cast"]).max())
    rows.append((i, abs(ctl.steer - want[0]), abs(ctl.throttle - want[1]), dpose, dxyz, dplan, dcast))
print("tick  |steer|    |throttle|  pose       stacked xyz  plan       cast")
for r in rows:
    print(f"{r[0]:4d}  {r[1]:.2e}  {r[2]:.2e}    {r[3]:.2e}   {r[4]:.2e}     {r[5]:.2e}   {r[6]:.2e}")
m = np.array(rows)[:, 1:].max(0)
print("max   " + "  ".join(f"{v:.2e}" for v in m))
h = a.pipeline.health()
print("health", {k: h[k] for k in ("plan_launches", "plan_aborts", "plans_recomputed", "nonfinite_outputs", "decode_mismatches", "pair_chain_timeouts")})

"""Where do un-forced time-outs of k_plan_persistent come from?  

Drives the HIP-graph agent over the ticks of tests/test_gpu_agent.py and prints, per tick: host wall time of the step, NaN
count of the plan, the kernel's diagnosis words (lav_gru_plan_diag), whether graphs were captured on this tick.

    python tools/plan_timeout_probe.py [--ticks 24] [--precapture] [--skip brake,others]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lav_amd import ops, synth  # noqa: E402
from lav_amd.agent import RoadOption  # noqa: E402
from lav_amd.lav_agent import LAVAgent  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=24)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--precapture", action="store_true")
    ap.add_argument("--points", type=int, default=8192)
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    p = os.path.join(d, "cfg.yaml")
    open(p, "w").write(yaml.safe_dump(dict(synthetic_weights=True, points_per_tick=a.points, precapture=a.precapture, hip_graphs=True)))
    agent = LAVAgent(p)
    sc = synth.agent_scenario()
    agent.set_global_plan([({"lat": la, "lon": lo, "z": 0.0}, RoadOption(int(c))) for la, lo, c in zip(sc["lat"], sc["lon"], sc["cmds"])])
    pipe = agent.pipeline
    orig_recover = pipe.recover_plan
    events = []

    def recover(out, cmd_value):
        diag = ops.gru_plan_diag(1, 512, 6, int(cmd_value), pipe.device, stream=pipe.s_ego)
        events.append(diag)
        return orig_recover(out, cmd_value)

    pipe.recover_plan = recover
    aborts = 0
    for i in range(0, a.ticks * a.stride, a.stride):
        n_graphs = len(pipe.graphs)
        t0 = time.time()
        ctl = agent.run_step(synth.agent_inputs(i, sc, n_points=a.points), i * 0.05)
        torch.cuda.synchronize()
        dt = (time.time() - t0) * 1e3
        cmd = 3
        line = f"tick {i:3d}  {dt:8.1f} ms  graphs {n_graphs}->{len(pipe.graphs)}"
        if events:
            line += "  ABORT " + str(events[-1])
            aborts += 1
            events.clear()
        else:
            dg = ops.gru_plan_diag(1, 512, 6, cmd, pipe.device, stream=pipe.s_ego)
            line += f"  ok entered {dg['entered']} completed {dg['completed']} status {dg['status']}"
        print(line, flush=True)
    print("plan_aborts", pipe.plan_aborts, "seen", aborts)


if __name__ == "__main__":
    main()

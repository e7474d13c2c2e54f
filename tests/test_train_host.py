"""Training path on the CPU (no GPU): the privileged-planner step against the REFERENCE trainer's loss terms
(tests/golden/make_golden.py:gold_train ran lav/lav_privileged_v2.py `train_bev` itself), the loss functions, and
data-parallel training over gloo with world_size 2."""
import os
import subprocess
import sys

import numpy as np
import torch

from lav_amd.train import LAV, DetLoss, TrainConfig, build_seg_mask, synthetic_bev_batch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_bev_two_steps_match_reference_trainer(golden):
    ref = golden["train"]["bev_terms"]
    lav = LAV(TrainConfig(), "cpu", what="bev")
    batch = synthetic_bev_batch(2, seed=11, num_objs=3)
    keys = ("plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
    for step in range(2):
        torch.manual_seed(100 + step)            # the jitter draws of BEVPlanner.forward come from the global CPU generator
        info = lav.train_bev(*batch, other_weight=0.5)
        # step 0: same weights, same jitter -> float noise only.  step 1 follows one Adam update, whose first step is
        # lr*sign(g) for every weight: rounding-level gradient differences flip the sign where g ~ 0, so the two runs
        # agree to a fraction of a percent, not to float precision
        np.testing.assert_allclose([info[k] for k in keys], ref[step], rtol=2e-4 if step == 0 else 2e-2, atol=1e-5, err_msg=f"step {step}")
    assert abs(ref[1] - ref[0]).max() > 1e-3          # the second step sees updated weights


def test_det_loss_and_seg_mask_shapes_and_values():
    g = torch.Generator().manual_seed(0)
    hm = torch.rand((2, 2, 32, 32), generator=g) ** 4
    pred = torch.randn((2, 2, 32, 32), generator=g)
    sz, psz = torch.rand((2, 2, 32, 32), generator=g), torch.randn((2, 2, 32, 32), generator=g)
    det, box, ori = DetLoss()(pred, hm, psz, sz, psz * 0.5, sz)
    # independent restatement of lav/models/loss.py:18-27
    w = hm.max(dim=1, keepdim=True)[0]
    p = torch.sigmoid(pred * (1 - 2 * hm))
    bce = torch.nn.BCEWithLogitsLoss(reduction="none")(pred, hm)
    assert torch.allclose(det, (bce * p).mean() / p.mean())
    assert torch.allclose(box, (w * torch.nn.SmoothL1Loss(reduction="none")(psz, sz)).mean() / w.mean())
    m = build_seg_mask(w=320, h=320, cx=160, cy=280)
    assert m.shape == (320, 320) and float(m[280, 160]) == 1.0 and float(m[0, 0]) < float(m[200, 160])


def test_world_size_2_gloo_data_parallel_train_step():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29617", os.path.join(REPO, "tests", "_gloo_train_worker.py")],
                         capture_output=True, text=True, timeout=600)
    assert "DDP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]

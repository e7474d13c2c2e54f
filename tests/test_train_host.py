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


def test_world_size_2_gloo_train_lidar_student_with_unused_parameters():
    """train_lidar through _Student + DistributedDataParallel, one rank without any vehicle (parameters that
    take no gradient on one rank must not stall the all-reduce)."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29631", os.path.join(REPO, "tests", "_gloo_train_lidar_worker.py")],
                         capture_output=True, text=True, timeout=900)
    assert "DDP_LIDAR_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_train_driver_epochs_scheduler_checkpoints(tmp_path):
    """train_bev_v2.py's loop (lav/train_bev_v2.py:18-34): epochs, StepLR stepped once per epoch, a checkpoint per
    --num-per-save epochs under the reference's file names, --config-path, resuming from a checkpoint."""
    import json
    import yaml
    cfgp = tmp_path / "config_v2.yaml"
    cfgp.write_text(yaml.safe_dump(dict(num_plan=20, num_cmds=6, cmd_weight=0.1, branch_weights=[5, 5, 5, 1, 1, 1], camera_x=1.5)))
    base = [sys.executable, os.path.join(REPO, "train_bev_v2.py"), "--synthetic", "--device", "cpu", "--config-path", str(cfgp),
            "--batch-size", "1", "--steps-per-epoch", "1", "--num-per-log", "1", "--save-dir", str(tmp_path / "ck")]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run(base + ["--num-epoch", "3", "--num-per-save", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    summary = json.loads(out.stdout.strip().splitlines()[-1])
    assert summary["epochs"] == 3 and summary["steps"] == 3
    assert summary["scheduler_epochs"] == 3 and summary["lr"] == 3e-4     # StepLR(step_size=32) stepped once per epoch
    assert sorted(os.listdir(tmp_path / "ck")) == ["bev_2.th"]
    sd = torch.load(tmp_path / "ck" / "bev_2.th")
    ref = LAV(TrainConfig(), "cpu", what="bev").bev_planner.state_dict()
    assert list(sd) == list(ref) and any(not torch.equal(sd[k], ref[k]) for k in sd)
    out2 = subprocess.run(base + ["--num-epoch", "1", "--bev", str(tmp_path / "ck" / "bev_2.th")], capture_output=True, text=True,
                          timeout=900, env=env)
    assert out2.returncode == 0, out2.stderr[-3000:]


def test_train_lidar_step_runs_on_the_host_through_the_oracle_stand_ins():
    """bench.py's cpu_baseline leg of the train_full line: the train_lidar step on torch CPU ops, with the oracle's restatements
    standing in for the two pieces that exist on HIP only (train-mode PointPillar front end, the frozen teacher's inference)."""
    import torch
    from lav_amd.point_pillar import PointPillarNet
    from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch
    from oracle import train_cpu
    restore = PointPillarNet.forward_train
    PointPillarNet.forward_train = lambda self, lidars, num_points: train_cpu.pillar_forward_train(self, lidars, num_points)
    try:
        torch.manual_seed(0)
        lav = LAV(TrainConfig(log_inference=False), torch.device("cpu"), what="lidar")
        batch = synthetic_lidar_batch(1, seed=41, max_points=4000, num_objs=2)
        with train_cpu.teacher_on_cpu(lav.bev_planner):
            before = [p.detach().clone() for p in lav.lidar_model.backbone.parameters()]
            info = lav.train_lidar(*batch)
        assert all(v == v and abs(v) < 1e6 for v in info.values() if isinstance(v, float)), info
        assert any(not torch.equal(a, b) for a, b in zip(before, lav.lidar_model.backbone.parameters()))
        assert "crop_feature" not in lav.bev_planner.__dict__          # the stand-ins are gone again
    finally:
        PointPillarNet.forward_train = restore


def test_train_mode_forwards_on_the_host_are_the_modules_own_torch_ops():
    """lav_amd/train/hipnn.py on a CPU tensor: ConvBackbone.forward_train and ResNet.forward_train (which route BatchNorm + ReLU
    through bn_act / conv_relu_bn) are exactly the nn.Sequential / BasicBlock arithmetic of the reference modules
    (team_code_v2/models/lidar.py:110-143, lav/models/resnet.py:59-80,165-172), running statistics included."""
    import copy
    import torch
    import torch.nn.functional as F
    from lav_amd.lidar import ConvBackbone
    from lav_amd.resnet import resnet18
    torch.manual_seed(2)
    bb = ConvBackbone(num_feature=8).train()
    ref = copy.deepcopy(bb)
    x = torch.randn(2, 8, 32, 32)
    got = bb(x)
    f1 = ref.conv1(x); f2 = ref.conv2(f1); f3 = ref.conv3(f2)
    want = torch.cat([ref.upconv1(f1), ref.upconv2(f2), ref.upconv3(f3)], dim=1)
    assert torch.equal(got, want)
    assert all(torch.equal(a, b) for a, b in zip(bb.buffers(), ref.buffers()))
    rn = resnet18(num_channels=5).train()
    rr = copy.deepcopy(rn)
    x = torch.randn(3, 5, 48, 48)
    got = rn(x)
    y = rr.maxpool(F.relu(rr.bn1(rr.conv1(x))))
    for i in range(1, 5):
        for blk in getattr(rr, f"layer{i}"):
            idt = y if blk.downsample is None else blk.downsample(y)
            y = F.relu(blk.bn2(blk.conv2(F.relu(blk.bn1(blk.conv1(y))))) + idt)
    assert torch.equal(got, y)
    assert all(torch.equal(a, b) for a, b in zip(rn.buffers(), rr.buffers()))


def test_decoder_caches_follow_load_state_dict_and_mode_toggles():
    """the stacked cast / plan GRU weights and the cached crop offsets are copies of the parameters; a
    load_state_dict or a train() -> optimiser step -> eval() round trip must invalidate them (the conv engines re-pack on the
    device; these caches are rebuilt)."""
    from lav_amd.bev_planner import BEVPlanner
    from lav_amd.uniplanner import UniPlanner
    cpu = torch.device("cpu")
    bev = BEVPlanner(num_cmds=6, num_plan=20, num_plan_iter=5, x_offset=0, y_offset=0.75)
    uni = UniPlanner(BEVPlanner(num_cmds=6, num_plan=20, num_plan_iter=5), num_cmds=6, num_plan=20, num_plan_iter=5, x_offset=0, y_offset=0.75)
    for m in (bev, uni):
        m.eval()
        before = m._dec(cpu)["cast"]["w_ih"].clone()
        assert m.offsets() == (0.0, 0.75)
        sd = {k: v + 1 for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        after = m._dec(cpu)["cast"]["w_ih"].clone()
        assert torch.equal(after, before + 1), type(m).__name__
        assert m.offsets() == (1.0, 1.75)
        m.train()
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1)
        m.eval()
        assert torch.equal(m._dec(cpu)["cast"]["w_ih"][0], m._cast_modules()[0][0].weight_ih_l0), type(m).__name__
        assert not torch.equal(m._dec(cpu)["cast"]["w_ih"], after)
        assert m.offsets() == (2.0, 2.75)

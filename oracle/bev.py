"""Oracle: BEV backbone, detection/segmentation heads, peak extraction, rotated crop,
ResNet-18 embedder, GRU cast / plan decoders - float32 on CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.  These are floating-point layers, so
the restatement is written with torch CPU functional ops (conv2d, conv_transpose2d,
grid_sample) plus explicit GRU recurrences, driven by a plain state_dict; no
nn.Module of the product or the reference is involved.  Follows
  team_code_v2/models/lidar.py:48-161        (ConvBackbone, Head)
  team_code_v2/model_inference.py:95-251     (det_inference, uniplanner_infer,
                                              extract_peak, crop_feature, transform_points)
  lav/models/resnet.py:39-82,148-250         (BasicBlock, ResNet forward up to layer4)
  team_code_v2/models/uniplanner.py:36-53,255-308 (embedder, cast, plan, cast_cmd_pred)
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


# --------------------------------------------------------------------------- backbone
def _conv_relu_bn(x, sd, p, i, stride):
    """lidar.py:57-60 pattern: Conv2d(3x3, pad 1, no bias) -> ReLU -> BatchNorm2d(eps 1e-3)."""
    x = F.conv2d(x, sd[f"{p}.{i}.weight"], None, stride, 1)
    return _bn(F.relu(x), sd, f"{p}.{i + 2}", 1e-3)


def conv_backbone(x, sd, p="backbone"):
    """ConvBackbone.forward (lidar.py:133-143): (B,64,320,320) -> (B,384,160,160)."""
    def stage(x, name, n):
        for j in range(n):
            x = _conv_relu_bn(x, sd, f"{p}.{name}", 3 * j, 2 if j == 0 else 1)
        return x
    x1 = stage(x, "conv1", 4)
    x2 = stage(x1, "conv2", 6)
    x3 = stage(x2, "conv3", 6)
    u1 = _bn(F.relu(F.conv_transpose2d(x1, sd[f"{p}.upconv1.0.weight"], None, 1, 0)), sd, f"{p}.upconv1.2", 1e-3)
    u2 = _bn(F.relu(F.conv_transpose2d(x2, sd[f"{p}.upconv2.0.weight"], None, 2, 1)), sd, f"{p}.upconv2.2", 1e-3)
    u3 = _bn(F.relu(F.conv_transpose2d(x3, sd[f"{p}.upconv3.0.weight"], None, 4, 1, 2)), sd, f"{p}.upconv3.2", 1e-3)
    return torch.cat([u1, u2, u3], dim=1)


def head(x, sd, p, sigmoid=False):
    """Head.forward (lidar.py:147-161): Conv3x3 -> ReLU -> BN -> ConvTranspose2d(3, s2, p1, op1) (+sigmoid)."""
    x = F.conv2d(x, sd[f"{p}.net.0.weight"], None, 1, 1)
    x = _bn(F.relu(x), sd, f"{p}.net.2", 1e-3)
    x = F.conv_transpose2d(x, sd[f"{p}.net.3.weight"], sd[f"{p}.net.3.bias"], 2, 1, 1)
    return torch.sigmoid(x) if sigmoid else x


def lidar_heads(feat, sd):
    return (head(feat, sd, "center_head"), head(feat, sd, "box_head"), head(feat, sd, "ori_head"),
            head(feat, sd, "seg_head", sigmoid=True))


# --------------------------------------------------------------------------- detection
def extract_peak(heatmap, max_pool_ks=7, max_det=15):
    """model_inference.py:189-202: 7x7 max-pool NMS then top-15 of the flattened map."""
    mx = F.max_pool2d(heatmap[None, None], kernel_size=max_pool_ks, padding=max_pool_ks // 2, stride=1)[0, 0]
    possible = heatmap - (mx > heatmap).float() * 1e5
    k = min(max_det, possible.numel())
    return torch.topk(possible.reshape(-1), k)


def det_inference(heatmaps, sizemaps, orimaps, ppm=4, min_score=0.2):
    """model_inference.py:95-121.  heatmaps already sigmoid-ed (2,H,W)."""
    dets = []
    for i, c in enumerate(heatmaps):
        det = []
        score, loc = extract_peak(c)
        for s, l in zip(score.tolist(), loc.tolist()):
            if not s > min_score:
                continue
            x, y = int(l) % c.size(1), int(l) // c.size(1)
            w, h = float(sizemaps[0, y, x]), float(sizemaps[1, y, x])
            cos, sin = float(orimaps[0, y, x]), float(orimaps[1, y, x])
            if i == 1 and max(w, h) < 0.1 * ppm:
                continue
            dist = float(np.linalg.norm([x - 160, y - 280]))
            if dist <= 2 or dist >= 30 * ppm:
                continue
            det.append((x, y, w, h, cos, sin))
        dets.append(det)
    return dets


# --------------------------------------------------------------------------- crop
def crop_feature(features, rel_locs, rel_oris, pixels_per_meter, crop_size, offset_x, offset_y):
    """model_inference.py:204-238: affine theta -> affine_grid -> bilinear grid_sample
    (zeros padding, align_corners=True)."""
    B, C, H, W = features.shape
    rel_locs = rel_locs.view(-1, 2) * pixels_per_meter / torch.tensor([H / 2, W / 2]).type_as(rel_locs)
    cos, sin = torch.cos(rel_oris), torch.sin(rel_oris)
    rx, ry = rel_locs[..., 0], rel_locs[..., 1]
    k = crop_size / H
    rot_x = -k * offset_x * cos + k * offset_y * sin + offset_x
    rot_y = -k * offset_x * sin - k * offset_y * cos + offset_y
    theta = torch.stack([torch.stack([k * cos, k * -sin, rot_x + rx], dim=-1),
                         torch.stack([k * sin, k * cos, rot_y + ry], dim=-1)], dim=-2)
    grid = F.affine_grid(theta, (B, C, crop_size, crop_size), align_corners=True)
    return F.grid_sample(features, grid, align_corners=True)


def transform_points(locs, oris):
    """model_inference.py:240-251."""
    cos, sin = torch.cos(oris), torch.sin(oris)
    R = torch.stack([torch.stack([cos, sin], dim=-1), torch.stack([-sin, cos], dim=-1)], dim=-2)
    return locs @ R


# --------------------------------------------------------------------------- ResNet-18 embedder
def _basic_block(x, sd, p, stride):
    """resnet.py:39-82."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1", 1e-5))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2", 1e-5)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd, p + ".downsample.1", 1e-5)
    return F.relu(out + x)


def resnet18_embed(x, sd, p="lidar_conv_emb.0"):
    """uniplanner.py:36-40 + resnet.py:235-247: ResNet-18 up to layer4, global average pool, flatten."""
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 3), sd, p + ".bn1", 1e-5))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _basic_block(x, sd, f"{p}.layer{li}.0", stride)
        x = _basic_block(x, sd, f"{p}.layer{li}.1", 1)
    return x.mean(dim=(2, 3))


# --------------------------------------------------------------------------- GRU decoders
def gru_sequence(u, h0, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU (1 layer, batch_first) restated.  u (B,T,I), h0 (B,H) -> (B,T,H).
    r = s(Wir x + bir + Whr h + bhr); z likewise; n = tanh(Win x + bin + r*(Whn h + bhn));
    h' = (1-z)*n + z*h.   Gate order in the stacked weights: r, z, n."""
    H = h0.shape[1]
    h = h0
    outs = []
    for t in range(u.shape[1]):
        gi = u[:, t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs.append(h)
    return torch.stack(outs, dim=1)


def _gru_w(sd, p):
    return sd[p + ".weight_ih_l0"], sd[p + ".weight_hh_l0"], sd[p + ".bias_ih_l0"], sd[p + ".bias_hh_l0"]


def cast(embd, sd, num_cmds=6, num_plan=20):
    """UniPlanner.cast (uniplanner.py:288-308).  Both modes use the *_ego weights (:296-300)."""
    B = embd.shape[0]
    u = embd[:, None, :].expand(B, num_plan, embd.shape[1])
    locs = []
    for i in range(num_cmds):
        out = gru_sequence(u, torch.zeros(B, 64), *_gru_w(sd, f"cast_grus_ego.{i}"))
        wp = out @ sd[f"cast_mlps_ego.{i}.weight"].T + sd[f"cast_mlps_ego.{i}.bias"]
        locs.append(torch.cumsum(wp, dim=1))
    return torch.stack(locs, dim=1)


def plan(embd, nxp, cast_locs, sd, pixels_per_meter=4, crop_size=192, num_cmds=6, num_plan=20, num_plan_iter=5):
    """UniPlanner.plan / _plan (uniplanner.py:255-286) -> (B, iters, cmds, T, 2)."""
    B = embd.shape[0]
    u0 = nxp * pixels_per_meter / crop_size * 2 - 1
    w = _gru_w(sd, "plan_gru")
    plan_loc = cast_locs
    res = []
    for _ in range(num_plan_iter):
        locs = []
        for i in range(num_cmds):
            u = torch.cat([u0[:, None, :].expand(B, num_plan, 2), plan_loc[:, i]], dim=2)
            out = gru_sequence(u, embd, *w)
            wp = out @ sd["plan_mlp.weight"].T + sd["plan_mlp.bias"]
            locs.append(torch.cumsum(wp, dim=1))
        plan_loc = torch.stack(locs, dim=1) + plan_loc
        res.append(plan_loc)
    return torch.stack(res, dim=1)


def cast_cmd_pred(embd, sd):
    """uniplanner.py:50-53."""
    return torch.sigmoid(embd @ sd["cast_cmd_pred.0.weight"].T + sd["cast_cmd_pred.0.bias"])


# --------------------------------------------------------------------------- InferModel.forward tail
def uniplanner_infer(features, det, cmd_value, nxp, sd, ppm=4, crop_size=96, offset_x=0.0, offset_y=0.75,
                     num_cmds=6, num_plan=20):
    """InferModel.uniplanner_infer (model_inference.py:123-187).  features (384,160,160)."""
    H, W = features.size(1) * 2, features.size(2) * 2
    center_x = float(W / 2 + offset_x * W / 2)
    center_y = float(H / 2 + offset_y * H / 2)
    locs, oris = [], []
    for X, Y, h, w, cos, sin in det:
        if np.linalg.norm([X - center_x, Y - center_y]) <= 4:
            continue
        locs.append([(X - center_x) / ppm, (Y - center_y) / ppm])
        oris.append(float(np.arctan2(sin, cos)))
    locs = torch.tensor(locs, dtype=torch.float32).reshape(-1, 2)
    oris = torch.tensor(oris, dtype=torch.float32)
    N = len(locs)
    if N > 0:
        crops = crop_feature(features.expand(N, *features.size()), locs, oris, ppm / 2, crop_size, offset_x, offset_y)
        other_embd = resnet18_embed(crops, sd)
        other_cast = cast(other_embd, sd, num_cmds, num_plan)
        other_cmds = cast_cmd_pred(other_embd, sd)
        other_cast = transform_points(other_cast, oris[:, None].repeat(1, num_cmds))
        other_cast = other_cast + locs.view(N, 1, 1, 2)
    else:
        other_cast = torch.zeros((0, num_cmds, num_plan, 2))
        other_cmds = torch.zeros((0, num_cmds))
    ego_crop = crop_feature(features[None], torch.zeros(1, 2), torch.zeros(1), ppm / 2, crop_size, offset_x, offset_y)
    ego_embd = resnet18_embed(ego_crop, sd)
    ego_cast = cast(ego_embd, sd, num_cmds, num_plan)
    ego_plan = plan(ego_embd, nxp[None], ego_cast, sd, ppm, crop_size * 2, num_cmds, num_plan)[0, -1, cmd_value]
    return ego_embd, ego_plan, ego_cast[0, cmd_value], other_cast, other_cmds


def move_lidar_points(lidar_xyz: torch.Tensor, dloc, ori0, ori1):
    """team_code_v2/lav_agent_fast.py:547-565 (dloc, ori in float64 numpy on the host, points float32)."""
    dloc = np.asarray(dloc) @ np.array([[math.cos(ori0), -math.sin(ori0)], [math.sin(ori0), math.cos(ori0)]])
    ori = ori1 - ori0
    R = torch.tensor([[math.cos(ori), math.sin(ori), 0], [-math.sin(ori), math.cos(ori), 0], [0, 0, 1]],
                     dtype=torch.float32)
    out = lidar_xyz @ R
    out[:, 0] += dloc[0]
    out[:, 1] += dloc[1]
    return out

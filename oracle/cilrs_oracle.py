"""CPU oracle for the CILRS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the shipped package (cilrs_mi355) never does.

What it restates (all citations are into /root/reference):

* ``CILRSOracle``         model/autonomous_drive.py:361-399 == notebook/notebook.ipynb:440-477
* ``resnet34_trunk``      torchvision.models.resnet34 (torchvision>=0.14, requirements.txt:2 --
                          NOT vendored in the reference and NOT installed here).  Restated from its
                          published architecture: 7x7/s2/p3 stem conv (no bias) + BN + ReLU +
                          3x3/s2/p1 max-pool, BasicBlock stacks [3,4,6,3] with widths
                          64/128/256/512, 1x1/s2 conv + BN down-sample on the first block of
                          layers 2-4, adaptive avg-pool.  The wiring is pinned by the three
                          known answers the reference holds: 22,421,453 parameters
                          (notebook/notebook.ipynb:52), the strict state_dict key contract
                          (model/autonomous_drive.py:497) and the 256.9 MB checkpoint
                          (notebook/notebook.ipynb:306).
* ``loss_l1`` (Config B)  notebook/notebook.ipynb:504-527
* ``loss_mse`` (Config A) README.md:104-105, configs/train_config.json:30-32 (documented only)
* ``train_step``          notebook/notebook.ipynb:549-555 (forward, loss, zero_grad, backward,
                          [clip], Adam step)
* ``preprocess_frame`` / ``predict_controls``  model/autonomous_drive.py:897-920 (without the
                          cv2.resize, cv2 being absent: frames are fed at 88x200 already)

Arithmetic is torch fp32 on the CPU -- the same library the reference's path runs on, so the
oracle IS "the reference PyTorch CPU path" for every op below the torchvision boundary.

Pinning: oracle/make_golden.py execs the reference's own ``CILRS`` / ``CILRSLoss`` /
``train_one_epoch`` source (ast-extracted as text) against ``resnet34_trunk`` as the stand-in for
the absent torchvision constructor, checks this restatement against it bit-for-bit, and writes
tests/golden/*.  The trunk's internals are therefore "pinned by known answers only" (see above).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

IMG_MEAN = (0.485, 0.456, 0.406)      # autonomous_drive.py:481
IMG_STD = (0.229, 0.224, 0.225)       # autonomous_drive.py:482
IMG_W, IMG_H = 200, 88                # autonomous_drive.py:483-484
SPEED_NORM = 90.0                     # autonomous_drive.py:485
N_PARAMS = 22_421_453                 # notebook/notebook.ipynb:52


# --------------------------------------------------------------------------------------
# ResNet-34 trunk (restatement of the un-vendored torchvision constructor)
# --------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet34Trunk(nn.Module):
    """Attribute names follow torchvision so the reference's nn.Sequential re-wrapping
    (autonomous_drive.py:366-370) yields the key names of SURVEY.md 8a/A1."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)          # dropped by the reference (:366-370)
        for m in self.modules():                # torchvision's default init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)


def resnet34_trunk(*_a, **_k):
    """Stand-in for ``torchvision.models.resnet34(pretrained=False)`` (no weights fetch)."""
    return ResNet34Trunk()


# --------------------------------------------------------------------------------------
# CILRS (autonomous_drive.py:361-399)
# --------------------------------------------------------------------------------------
class CILRSOracle(nn.Module):
    def __init__(self, num_commands=4, dropout=0.0):
        super().__init__()
        self.num_commands = num_commands
        r = resnet34_trunk()
        self.visual_encoder = nn.Sequential(
            r.conv1, r.bn1, r.relu, r.maxpool, r.layer1, r.layer2, r.layer3, r.layer4,
            r.avgpool, nn.Flatten())
        self.speed_encoder = nn.Sequential(
            nn.Linear(1, 128), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(128, 128), nn.ReLU(inplace=True))
        self.control_branches = nn.ModuleList([
            nn.Sequential(
                nn.Linear(640, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 3))
            for _ in range(num_commands)])
        self.speed_predictor = nn.Sequential(
            nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(256, 256), nn.ReLU(inplace=True),
            nn.Linear(256, 1))

    def forward(self, image, speed, command):
        visual = self.visual_encoder(image)
        speed_feat = self.speed_encoder(speed.unsqueeze(1))
        combined = torch.cat([visual, speed_feat], dim=1)
        pred_speed = self.speed_predictor(visual).squeeze(1)
        b = image.size(0)
        all_out = torch.stack([br(combined) for br in self.control_branches], dim=0)
        idx = command.unsqueeze(0).unsqueeze(2).expand(1, b, 3)
        controls = all_out.gather(0, idx).squeeze(0)
        return controls, pred_speed


# Dropout sites in the order the HIP engine numbers them (include/cilrs_hip.h, cilrs_dropout):
# 0 speed_encoder.2; 1 + 2k control_branches.k.2; 2 + 2k control_branches.k.5; 9 speed_predictor.2
DROPOUT_SITES = {"speed_encoder": (0,), "speed_predictor": (9,),
                 **{f"control_branches.{k}": (1 + 2 * k, 2 + 2 * k) for k in range(4)}}


def _seq_with_masks(seq, x, masks, sites):
    """nn.Sequential forward with every nn.Dropout replaced by a multiplication with the given
    mask (values 0 or 1/(1-p)): the functional form of training-mode dropout
    (autonomous_drive.py:371-387) under a KNOWN mask instead of torch's CPU RNG stream."""
    it = iter(sites)
    for layer in seq:
        x = x * masks[next(it)] if isinstance(layer, nn.Dropout) else layer(x)
    return x


def forward_with_dropout_masks(model: CILRSOracle, image, speed, command, masks):
    """CILRSOracle.forward (autonomous_drive.py:389-399) in train mode with the dropout masks
    supplied by the caller: masks[site] has the shape of the activation that site drops."""
    visual = model.visual_encoder(image)
    speed_feat = _seq_with_masks(model.speed_encoder, speed.unsqueeze(1), masks,
                                 DROPOUT_SITES["speed_encoder"])
    combined = torch.cat([visual, speed_feat], dim=1)
    pred_speed = _seq_with_masks(model.speed_predictor, visual, masks,
                                 DROPOUT_SITES["speed_predictor"]).squeeze(1)
    b = image.size(0)
    all_out = torch.stack([_seq_with_masks(br, combined, masks,
                                           DROPOUT_SITES[f"control_branches.{k}"])
                           for k, br in enumerate(model.control_branches)], dim=0)
    idx = command.unsqueeze(0).unsqueeze(2).expand(1, b, 3)
    controls = all_out.gather(0, idx).squeeze(0)
    return controls, pred_speed


# --------------------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------------------
LOSS_KEYS = ("total", "control", "steer", "throttle", "brake", "speed")


def loss_l1(pred_controls, target_controls, pred_speed, target_speed,
            steer_w=5.0, throttle_w=1.0, brake_w=1.0, speed_w=0.5):
    """Config B -- notebook/notebook.ipynb:514-527."""
    l1 = nn.functional.l1_loss
    steer = l1(pred_controls[:, 0], target_controls[:, 0])
    throttle = l1(pred_controls[:, 1], target_controls[:, 1])
    brake = l1(pred_controls[:, 2], target_controls[:, 2])
    control = steer_w * steer + throttle_w * throttle + brake_w * brake
    speed = nn.functional.mse_loss(pred_speed, target_speed)
    total = control + speed_w * speed
    return total, dict(total=total.item(), control=control.item(), steer=steer.item(),
                       throttle=throttle.item(), brake=brake.item(), speed=speed.item())


def loss_mse(pred_controls, target_controls, pred_speed, target_speed, speed_w=0.05):
    """Config A -- documented only (README.md:104-105, configs/train_config.json:30-32):
    nn.MSELoss() over the [B,3] controls + speed_loss_weight * nn.MSELoss() on speed.
    The per-channel entries are the per-channel MSEs (control == their mean)."""
    mse = nn.functional.mse_loss
    control = mse(pred_controls, target_controls)
    speed = mse(pred_speed, target_speed)
    total = control + speed_w * speed
    with torch.no_grad():
        per = ((pred_controls - target_controls) ** 2).mean(dim=0)
    return total, dict(total=total.item(), control=control.item(), steer=per[0].item(),
                       throttle=per[1].item(), brake=per[2].item(), speed=speed.item())


@dataclass
class TrainConfig:
    """A = BASELINE.json / configs/train_config.json; B = the executed notebook (nb:489-502)."""
    name: str = "A"
    lr: float = 2e-4
    weight_decay: float = 1e-4
    loss: str = "mse"                      # "mse" | "l1"
    loss_weights: tuple = (1.0, 1.0, 1.0, 0.05)   # steer, throttle, brake, speed
    grad_clip: float = 0.0
    dropout: float = 0.0
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8


CONFIG_A = TrainConfig()
CONFIG_B = TrainConfig(name="B", lr=1e-4, loss="l1", loss_weights=(5.0, 1.0, 1.0, 0.5),
                       grad_clip=1.0)


def compute_loss(cfg: TrainConfig, pc, tc, ps, ts):
    if cfg.loss == "l1":
        w = cfg.loss_weights
        return loss_l1(pc, tc, ps, ts, w[0], w[1], w[2], w[3])
    return loss_mse(pc, tc, ps, ts, cfg.loss_weights[3])


def make_optimizer(model, cfg: TrainConfig):
    """notebook/notebook.ipynb:533-534: Adam with coupled L2 weight decay over ALL parameters."""
    return torch.optim.Adam(model.parameters(), lr=cfg.lr, betas=cfg.betas, eps=cfg.eps,
                            weight_decay=cfg.weight_decay)


def train_step(model, optimizer, cfg: TrainConfig, imgs, speeds, cmds, tgts):
    """One iteration of train_one_epoch's loop body (notebook/notebook.ipynb:549-555).
    ``speeds`` is both an input and the speed head's regression target (nb:550)."""
    model.train()
    pred_ctrl, pred_spd = model(imgs, speeds, cmds)
    loss, ld = compute_loss(cfg, pred_ctrl, tgts, pred_spd, speeds)
    optimizer.zero_grad()
    loss.backward()
    gnorm = None
    if cfg.grad_clip > 0:
        gnorm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_clip))
    optimizer.step()
    return ld, gnorm


@torch.no_grad()
def validate_batches(model, cfg, batches):
    """notebook/notebook.ipynb:563-585: mean of batch means + per-command mean |steer error|."""
    model.eval()
    losses = {k: 0.0 for k in LOSS_KEYS}
    cmd_err = {i: [] for i in range(4)}
    n = 0
    for imgs, speeds, cmds, tgts in batches:
        pc, ps = model(imgs, speeds, cmds)
        _, ld = compute_loss(cfg, pc, tgts, ps, speeds)
        for k, v in ld.items():
            losses[k] += v
        n += 1
        serr = (pc[:, 0] - tgts[:, 0]).abs()
        for ci in range(4):
            m = cmds == ci
            if m.any():
                cmd_err[ci].extend(serr[m].numpy().tolist())
    names = {0: "FOLLOW", 1: "LEFT", 2: "RIGHT", 3: "STRAIGHT"}
    cmd_avg = {names[i]: (float(np.mean(e)) if e else float("nan")) for i, e in cmd_err.items()}
    return {k: v / max(n, 1) for k, v in losses.items()}, cmd_avg


# --------------------------------------------------------------------------------------
# Inference adapter (autonomous_drive.py:897-920), cv2.resize excluded (cv2 absent)
# --------------------------------------------------------------------------------------
def preprocess_frame(rgb_u8_hwc: np.ndarray) -> torch.Tensor:
    """uint8 RGB [88,200,3] -> f32 [1,3,88,200]: /255, HWC->CHW, (x-mean)/std (:898-902)."""
    assert rgb_u8_hwc.shape == (IMG_H, IMG_W, 3) and rgb_u8_hwc.dtype == np.uint8
    img = torch.from_numpy(rgb_u8_hwc.astype(np.float32) / 255.0).permute(2, 0, 1)
    mean = torch.tensor(IMG_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMG_STD, dtype=torch.float32).view(3, 1, 1)
    return ((img - mean) / std).unsqueeze(0)


def resize_bilinear_u8(frame: np.ndarray, width: int = IMG_W, height: int = IMG_H) -> np.ndarray:
    """cv2.resize(frame, (width, height)) for uint8 HWC frames with the default INTER_LINEAR
    (autonomous_drive.py:898).  cv2 is not installed in the build image, so this restates
    OpenCV's published 8-bit algorithm -- PARITY UNPINNED against cv2 itself:
      coefficient f = float((d + 0.5) * scale - 0.5), s = floor(f), f -= s; horizontally a tap
      left of 0 / right of the last column drops its fraction; weights round(f * 2048) and
      round((1 - f) * 2048) (nearest-even) as int16; horizontal pass in int32; vertical pass
      ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 with rows clipped."""
    assert frame.dtype == np.uint8 and frame.ndim == 3
    sh, sw = frame.shape[:2]
    if (sh, sw) == (height, width):
        return frame.copy()

    def coefs(dsize, ssize, zero_edge):
        scale = 1.0 / (np.float64(dsize) / np.float64(ssize))
        f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = f - s0.astype(np.float32)
        if zero_edge:
            lo, hi = s0 < 0, s0 >= ssize - 1
            f = np.where(lo | hi, np.float32(0), f)
            s0 = np.where(lo, 0, np.where(hi, ssize - 1, s0))
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s0, c0, c1

    sx, a0, a1 = coefs(width, sw, True)
    sy, b0, b1 = coefs(height, sh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    src = frame.astype(np.int64)
    hrow = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]   # [sh, W, C]
    h0, h1 = hrow[y0], hrow[y1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_camera(frame_u8: np.ndarray) -> torch.Tensor:
    """Whole preprocess_image (:897-902): camera frame (any size, 3 or 4 bytes per pixel, first
    three kept like :870) -> f32 [1,3,88,200]."""
    return preprocess_frame(resize_bilinear_u8(np.ascontiguousarray(frame_u8[:, :, :3])))


@torch.no_grad()
def predict_controls(model, rgb_u8_hwc, speed_kmh, command_idx):
    """(:908-920) -> (steer, throttle, brake, speed_kmh)."""
    model.eval()
    img = preprocess_frame(rgb_u8_hwc)
    spd = torch.tensor([min(speed_kmh / SPEED_NORM, 1.0)], dtype=torch.float32)
    cmd = torch.tensor([command_idx], dtype=torch.long)
    pc, ps = model(img, spd, cmd)
    return (pc[0, 0].item(), pc[0, 1].item(), pc[0, 2].item(), ps[0].item() * SPEED_NORM)


# --------------------------------------------------------------------------------------
# Portable counter-based initialiser + synthetic batches (weights are 89.7 MB: never committed)
# --------------------------------------------------------------------------------------
def _hash_u01(seed: int, stream: int, n: int) -> np.ndarray:
    """splitmix64 of (seed, stream, index) -> float64 uniform in [0,1). Pure integer arithmetic,
    identical on every platform."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64)
             + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
             + np.uint64((stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF))
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def portable_state_dict(template_sd, seed: int = 0):
    """Fill a CILRS state_dict (250 entries) deterministically from (seed, entry index, element
    index).  Conv/linear weights ~ U(+-sqrt(3/fan_in)), BN gamma ~ U(0.7,1.3) (U(0.2,0.5) for
    each block's bn2 so eval-mode activations stay O(1)), BN beta and biases ~ U(-0.2,0.2), running_mean ~ U(-0.2,0.2), running_var ~ U(0.5,1.5),
    num_batches_tracked = 0.  Logical (OIHW / [out,in]) element order."""
    out = {}
    for idx, (name, t) in enumerate(template_sd.items()):
        shape = tuple(t.shape)
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        u = _hash_u01(seed, idx, n)
        if name.endswith("running_var"):
            v = 0.5 + u
        elif name.endswith("running_mean"):
            v = (u - 0.5) * 0.4
        elif len(shape) == 4 or (len(shape) == 2):
            fan_in = int(np.prod(shape[1:]))
            a = math.sqrt(3.0 / fan_in)                       # var(w) = 1/fan_in
            v = (2.0 * u - 1.0) * a
        elif name.endswith("bn2.weight") or name.endswith("bn3.weight"):
            v = 0.2 + 0.3 * u       # residual-branch BN gamma: small, keeps eval activations O(1)
        elif name.endswith(".weight"):                        # other BN gammas
            v = 0.7 + 0.6 * u
        else:                                                 # BN beta, linear bias
            v = (u - 0.5) * 0.4
        out[name] = torch.from_numpy(v.astype(np.float32)).reshape(shape).clone()
    return out


def synthetic_batch(batch: int, seed: int = 1, h: int = IMG_H, w: int = IMG_W):
    """SURVEY.md 8d config 2 inputs: images = ImageNet-normalised U{0..255} uint8, speed ~ U[0,1),
    command ~ U{0..3}, targets steer ~ U[-1,1], throttle/brake ~ U[0,1]."""
    u8 = np.floor(_hash_u01(seed, 1000, batch * h * w * 3) * 256.0).astype(np.uint8)
    u8 = u8.reshape(batch, h, w, 3)
    img = torch.from_numpy(u8.astype(np.float32) / 255.0).permute(0, 3, 1, 2)
    mean = torch.tensor(IMG_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMG_STD, dtype=torch.float32).view(1, 3, 1, 1)
    img = ((img - mean) / std).contiguous()
    speed = torch.from_numpy(_hash_u01(seed, 1001, batch).astype(np.float32))
    cmd = torch.from_numpy(np.floor(_hash_u01(seed, 1002, batch) * 4.0).astype(np.int64))
    t = _hash_u01(seed, 1003, batch * 3).reshape(batch, 3)
    t[:, 0] = 2.0 * t[:, 0] - 1.0
    tgt = torch.from_numpy(t.astype(np.float32))
    return img, speed, cmd, tgt, u8


def build_oracle(seed: int = 0, dropout: float = 0.0) -> CILRSOracle:
    m = CILRSOracle(4, dropout)
    m.load_state_dict(portable_state_dict(m.state_dict(), seed), strict=True)
    return m

"""CPU oracle of the ResNet-50 variant (BASELINE.json configs[3]) -- TEST INFRASTRUCTURE ONLY.

The reference has NO such model: model/autonomous_drive.py:365 builds resnet34 only, and
BASELINE.json asks for a "ResNet-50 backbone variant, 400x176 input, bf16 MFMA path" on top of it.
This file is therefore the DEFINITION the HIP engine's variant 1 is checked against, not a
restatement of reference code: parity for this variant is "parity unpinned" by the reference.

What it follows:
* trunk: torchvision.models.resnet50's published architecture (v1.5: Bottleneck [3,4,6,3],
  1x1 -> 3x3 (stride) -> 1x1 x4, 1x1 down-sample + BN on the first block of every layer, 7x7/s2
  stem + BN + ReLU + 3x3/s2 max-pool, adaptive avg-pool), re-wrapped into ``visual_encoder`` the
  way the reference re-wraps resnet34 (autonomous_drive.py:366-370) so the state_dict keys follow
  the same scheme;
* heads and forward: exactly the reference's (autonomous_drive.py:371-399) with the feature width
  512 replaced by 2048 (first Linear of each branch 2176 wide, of the speed predictor 2048 wide).

Known answer used to pin the wiring: torchvision's ResNet-50 has 25,557,032 parameters of which
the dropped fc layer holds 2,049,000 -> 23,508,032 in the trunk.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import cilrs_oracle as O

TRUNK_PARAMS = 25_557_032 - (2048 * 1000 + 1000)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


def _layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                             nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down)]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class CILRSResNet50Oracle(nn.Module):
    def __init__(self, num_commands=4, dropout=0.0):
        super().__init__()
        self.visual_encoder = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1),
            _layer(64, 64, 3, 1), _layer(256, 128, 4, 2), _layer(512, 256, 6, 2),
            _layer(1024, 512, 3, 2), nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())
        self.speed_encoder = nn.Sequential(
            nn.Linear(1, 128), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(128, 128), nn.ReLU(inplace=True))
        self.control_branches = nn.ModuleList([
            nn.Sequential(
                nn.Linear(2048 + 128, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 3))
            for _ in range(num_commands)])
        self.speed_predictor = nn.Sequential(
            nn.Linear(2048, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(256, 256), nn.ReLU(inplace=True),
            nn.Linear(256, 1))

    forward = O.CILRSOracle.forward           # autonomous_drive.py:389-399, unchanged


def build_oracle50(seed: int = 0) -> CILRSResNet50Oracle:
    m = CILRSResNet50Oracle()
    m.load_state_dict(O.portable_state_dict(m.state_dict(), seed), strict=True)
    return m

"""``CILRS`` -- drop-in for the reference's nn.Module of the same name.

Mirrors model/autonomous_drive.py:361-399 (== notebook/notebook.ipynb:440-477):

* ctor ``CILRS(num_commands=4, dropout=p)``;
* ``forward(image f32[B,3,H,W], speed f32[B], command i64[B]) -> (controls f32[B,3],
  pred_speed f32[B])`` -- a 2-tuple, exactly what ``predict_controls`` (:915-917) and
  ``train_one_epoch`` (nb:549) unpack;
* the strict ``state_dict`` contract (:497): 250 entries, same names / logical shapes / dtypes.

The module tree holds ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` objects purely
as parameter containers; none of their ``forward`` methods ever runs.  All arithmetic happens in
libcilrs_hip.so (hand-written gfx950 kernels) through :mod:`cilrs_mi355.engine`; on a device
without the HIP engine ``forward`` raises -- there is no eager/CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .engine import Engine


class _NoEagerForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("cilrs_mi355 modules are parameter containers; call CILRS.forward, "
                           "which runs the HIP engine (no eager fallback)")


class BasicBlock(_NoEagerForward):
    """Parameter container with torchvision BasicBlock's attribute names."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_NoEagerForward):
    """Parameter container with torchvision Bottleneck's attribute names (ResNet-50 variant)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)      # stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


def _make_bottleneck_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                             nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down)]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def _make_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                             nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, down)]
    layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class CILRS(nn.Module):
    VARIANT = 0          # engine architecture variant (include/cilrs_hip.h)
    FEATURES = 512

    def _trunk(self):
        # ResNet-34 trunk, re-wrapped exactly as autonomous_drive.py:366-370 does, so indices
        # 0,1,4,5,6,7 carry the parameters and 2,3,8,9 carry none.
        return nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1),
            _make_layer(64, 64, 3, 1), _make_layer(64, 128, 4, 2),
            _make_layer(128, 256, 6, 2), _make_layer(256, 512, 3, 2),
            nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())

    def __init__(self, num_commands=4, dropout=0.0):
        super().__init__()
        # generic like the reference's ctor (autonomous_drive.py:362, 380-381); every caller of the
        # reference passes 4.  The engine lays its arena out per plan for 1..8 branches; the
        # single-launch persistent frame kernel and the evaluation report exist for 4 only.
        if not (isinstance(num_commands, int) and 1 <= num_commands <= 8):
            raise ValueError("num_commands must be an integer in 1..8")
        self.num_commands = num_commands
        self.dropout = float(dropout)
        self.visual_encoder = self._trunk()
        feat = self.FEATURES
        self.speed_encoder = nn.Sequential(
            nn.Linear(1, 128), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(128, 128), nn.ReLU(inplace=True))
        self.control_branches = nn.ModuleList([
            nn.Sequential(
                nn.Linear(feat + 128, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Linear(256, 3))
            for _ in range(num_commands)])
        self.speed_predictor = nn.Sequential(
            nn.Linear(feat, 256), nn.ReLU(inplace=True), nn.Dropout(dropout),
            nn.Linear(256, 256), nn.ReLU(inplace=True),
            nn.Linear(256, 1))
        # torchvision.models.resnet34(pretrained=False) initialisation (:365)
        for m in self.visual_encoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._engine = None
        self._dropout_calls = 0
        # anything that may rewrite parameters / buffers invalidates what the engine derived
        # from them for inference (engine.weights_key)
        self.register_load_state_dict_post_hook(lambda module, _keys: module.weights_changed())

    _live = None          # weak set of modules with an engine (global optimizer hook below)

    def weights_changed(self):
        """Tell the engine that parameters or BatchNorm buffers were modified in place (it then
        re-derives its cached inference state: folded BatchNorm scale/shift, 16-bit weights).
        Called automatically by load_state_dict, by train()/eval() switches and by the engine's
        own training kernels; needed only after other in-place edits made in eval mode."""
        if self._engine is not None:
            self._engine.weights_epoch += 1

    def train(self, mode=True):
        if mode != self.training:
            self.weights_changed()
        return super().train(mode)

    # -- engine plumbing ---------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.cpu() re-create parameter storage: the flat arena must be rebuilt
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self) -> Engine:
        """The HIP engine bound to this module's parameters (built on first use)."""
        eng = self._engine
        if eng is None or not eng.is_attached():
            # architecture code of the C-ABI: trunk | num_commands << 8 (0 = the reference's 4)
            code = self.VARIANT | ((self.num_commands << 8) if self.num_commands != 4 else 0)
            eng = Engine(self, code)
            self._engine = eng
            _track(self)
        return eng

    # -- the reference's forward signature -----------------------------------------------------
    def forward(self, image, speed, command):
        eng = self.engine()
        seed = 0
        p = self.dropout if self.training else 0.0
        if p > 0.0:
            from .train import dropout_seed
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._dropout_calls += 1
            seed = dropout_seed(torch.initial_seed(), self._dropout_calls, rank)
        return eng.forward(image, speed, command, self.training, p, seed)


def _track(module):
    """In-place parameter writes made while the module stays in eval mode (frozen-BatchNorm
    fine-tuning with torch.optim, ...) must invalidate the engine's cached inference state.  One
    global optimizer post-step hook bumps the weights epoch of every live engine whose parameters
    the stepping optimizer owns."""
    import weakref
    if CILRS._live is None:
        CILRS._live = weakref.WeakSet()

        def _after_step(optimizer, *_a, **_k):
            owned = {id(p) for g in optimizer.param_groups for p in g["params"]}
            for m in list(CILRS._live):
                eng = m._engine
                if eng is not None and id(eng._first_param) in owned:
                    eng.weights_epoch += 1
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(_after_step)
    CILRS._live.add(module)


class CILRSResNet50(CILRS):
    """BASELINE.json configs[3]: the same CILRS heads on a ResNet-50 trunk (torchvision-style
    Bottleneck stacks [3,4,6,3], stride on the 3x3 convolution; 2048-d features, so the first
    Linear of every branch is 2176 wide and the speed predictor's 2048 wide).  The reference has
    no such model (model/autonomous_drive.py:365 is resnet34 only); state_dict keys follow the
    same re-wrapping (visual_encoder.{4..7}.{blk}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample}).
    Trains in fp32 through the same HIP kernels as the reference network (train-mode forward,
    backward, Trainer); inference in fp32, or through Engine.run_forward_u8(half="bf16") with the
    trunk on the bf16 matrix pipe."""
    VARIANT = 1
    FEATURES = 2048

    def _trunk(self):
        return nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1),
            _make_bottleneck_layer(64, 64, 3, 1), _make_bottleneck_layer(256, 128, 4, 2),
            _make_bottleneck_layer(512, 256, 6, 2), _make_bottleneck_layer(1024, 512, 3, 2),
            nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())

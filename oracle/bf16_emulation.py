"""CPU emulation of the bf16 training mode (CILRS_PLAN_BF16_TRAIN) -- TEST INFRASTRUCTURE ONLY.

Not a restatement of reference code: the reference trains in fp32 (notebook/notebook.ipynb:440-477,
549-555); BASELINE.json configs[3] asks for a "bf16 MFMA path".  This file DEFINES what that mode
computes so that the HIP kernels can be checked tightly instead of against a loose "close to fp32"
bound.  Parity for this mode is "parity unpinned" by the reference.

Round 4 definition -- 16-bit tensors end to end (what autocast-style mixed precision stores):

* every trunk tensor after the stem lives in bf16: the max-pool output, every convolution's raw
  output y, every post-BatchNorm / ReLU / residual tensor z, and in the backward pass every gradient
  tensor (d block output, the identity-path gradient, dy of every convolution);
* a convolution multiplies bf16 operands and accumulates in fp32 (float64 when the model is
  .double()); its result is rounded to bf16 ONCE, after the optional fp32 addition of the other
  gradient branch in the backward pass;
* BatchNorm is the fp32 BatchNorm of the STORED tensor: batch statistics, normalisation and the
  backward reductions all read the rounded y; its output (after ReLU / the residual add) and its
  input gradient are rounded to bf16 once;
* weight gradients are fp32 (fp32 accumulation of bf16 x and bf16 dy); the stem, the pooling, the
  heads, the loss, Adam and the master weights are the fp32 oracle unchanged; the gradient handed
  from layer1 to the stem is NOT rounded (the HIP kernel writes that one tensor in fp32).

One known difference from the HIP order of operations: in a block with a down-sample branch the HIP
path rounds d(block input) twice (after the main branch's data gradient and again after adding the
down-sample branch's), autograd here adds both in fp32 and rounds once -- a 2^-9 relative effect on
four (ResNet-34: three) tensors per step, far inside the tolerance of the tests.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _r(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        xr, wr = _r(x), _r(w)
        ctx.save_for_backward(xr, wr)
        ctx.sp = (stride, pad)
        return F.conv2d(xr, wr, None, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        s, p = ctx.sp
        dyr = _r(dy)
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, s, p)
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, s, p)
        return dx, dw, None, None


class RoundedConv2d(nn.Conv2d):
    def forward(self, x):
        return _RoundedConv.apply(x, self.weight, self.stride[0], self.padding[0])


class _Round(torch.autograd.Function):
    """A tensor stored in bf16: the value is rounded in the forward pass, its gradient (also a
    stored bf16 tensor) in the backward pass."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundForward(torch.autograd.Function):
    """Rounded value, gradient passed through unrounded (the max-pool output: its gradient is
    the one tensor the HIP path hands to the fp32 stem in fp32)."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _PoolThenRound(nn.Module):
    def __init__(self, pool):
        super().__init__()
        self.pool = pool

    def forward(self, x):
        return _RoundForward.apply(self.pool(x))


def _block_forward(self, x):
    """BasicBlock / Bottleneck forward (oracle/cilrs_oracle.py:66-74, oracle/resnet50_oracle.py:
    45-53) with every stored tensor rounded."""
    pairs = [(self.conv1, self.bn1), (self.conv2, self.bn2)]
    if hasattr(self, "conv3"):
        pairs.append((self.conv3, self.bn3))
    out = x
    for i, (conv, bn) in enumerate(pairs):
        t = bn(_Round.apply(conv(out)))
        if i + 1 < len(pairs):
            out = _Round.apply(F.relu(t))
    identity = x
    if self.downsample is not None:
        identity = _Round.apply(self.downsample[1](_Round.apply(self.downsample[0](x))))
    return _Round.apply(F.relu(t + identity))


def to_bf16_emulation(model):
    """In place: every convolution of visual_encoder.{4..7} (layer1..layer4, BasicBlock or
    Bottleneck, down-sample branches included) becomes a RoundedConv2d sharing its Parameter, every
    residual block stores its tensors in bf16, and the max-pool output is rounded.  The stem
    (visual_encoder.0 .. 2) stays fp32, like the HIP mode.  state_dict keys are unchanged."""
    import types
    for li in (4, 5, 6, 7):
        layer = model.visual_encoder[li]
        for mod in layer.modules():
            for name, child in list(mod.named_children()):
                if type(child) is nn.Conv2d:
                    rc = RoundedConv2d(child.in_channels, child.out_channels, child.kernel_size,
                                       child.stride, child.padding, bias=False)
                    rc.weight = child.weight
                    setattr(mod, name, rc)
        for blk in layer:
            blk.forward = types.MethodType(_block_forward, blk)
    if not isinstance(model.visual_encoder[3], _PoolThenRound):
        model.visual_encoder[3] = _PoolThenRound(model.visual_encoder[3])
    return model

"""CPU emulation of the bf16 training mode (CILRS_PLAN_BF16_TRAIN) -- TEST INFRASTRUCTURE ONLY.

Not a restatement of reference code: the reference trains in fp32 (notebook/notebook.ipynb:440-477,
549-555); BASELINE.json configs[3] asks for a "bf16 MFMA path".  This file DEFINES what that mode
computes so that the HIP kernels can be checked tightly instead of against a loose "close to fp32"
bound: every trunk convolution after the stem rounds its three GEMM operand tensors to bf16 --
activations and weights in the forward and in the weight gradient, the output gradient in both
gradients -- and multiplies / accumulates in fp32 (float64 when the model is .double()); everything
else (BatchNorm, ReLU, residual adds, the stem, pooling, the heads, the loss, Adam and the master
weights) is the fp32 oracle unchanged.  Parity for this mode is "parity unpinned" by the reference.
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


def to_bf16_emulation(model):
    """In place: every convolution of visual_encoder.{4..7} (layer1..layer4, BasicBlock or
    Bottleneck, down-sample branches included) becomes a RoundedConv2d sharing its Parameter.
    The stem (visual_encoder.0) stays fp32, like the HIP mode."""
    for li in (4, 5, 6, 7):
        layer = model.visual_encoder[li]
        for mod in layer.modules():
            for name, child in list(mod.named_children()):
                if type(child) is nn.Conv2d:
                    rc = RoundedConv2d(child.in_channels, child.out_channels, child.kernel_size,
                                       child.stride, child.padding, bias=False)
                    rc.weight = child.weight
                    setattr(mod, name, rc)
    return model

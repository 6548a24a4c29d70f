#!/usr/bin/env python3
"""The stem convolution of the training step: csrc/stem_f32.hip (cilrs_stem_conv_fwd) against the
implicit GEMM on the channel-padded image (cilrs_conv2d_fwd, Cin = 4), forward with nothing else
running; B = 128 at 88x200 (the reference) and B = 64 at 176x400 (the ResNet-50 variant)."""
import ctypes as C
import sys
sys.path.insert(0, "cilrs-autonomous-driving-carla_amd")
import torch
from cilrs_mi355 import _lib as L

lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, H, W) in ((128, 88, 200), (64, 176, 400)):
    Ho, Wo = H // 2, W // 2
    x4 = torch.randn(N, H, W, 4, device="cuda")
    x4[..., 3] = 0
    w3 = torch.randn(64, 7, 7, 3, device="cuda") / 147 ** 0.5
    w4 = torch.zeros(64, 7, 7, 4, device="cuda")
    w4[..., :3] = w3
    y = torch.empty(N, Ho, Wo, 64, device="cuda")
    y2 = torch.empty_like(y)
    part = torch.empty(2 * 64 * (N * Ho * Wo // 64 + 64), device="cuda")
    scratch = torch.empty(1 << 20, device="cuda")
    rows = C.c_int(0)

    def new():
        L.check(lib.cilrs_stem_conv_fwd(L.ptr(x4), L.ptr(w3), L.ptr(y), L.ptr(part), N, H, W,
                                        C.byref(rows), st))

    def old():
        L.check(lib.cilrs_conv2d_fwd(L.ptr(x4), L.ptr(w4), L.ptr(y2), N, H, W, 4, 64, 7, 7, 2, 3, -1, 0,
                                     L.ptr(scratch), scratch.numel(), st))
    tn, to = timed(new), timed(old)
    fl = 2.0 * N * Ho * Wo * 64 * 147
    print(f"stem fwd N={N} {H}x{W}: stem_f32 {tn:7.1f} us ({fl / tn / 1e6:6.1f} TF, {rows.value} tiles)   "
          f"implicit GEMM (Cin 4) {to:7.1f} us ({fl / to / 1e6:6.1f} TF)   max |diff| "
          f"{(y - y2).abs().max().item():.2e}", flush=True)
    # weight gradient
    dy = torch.randn(N, Ho, Wo, 64, device="cuda")
    need = lib.cilrs_stem_conv_wgrad_scratch_floats(N, H, W)
    sc2 = torch.empty(max(need, 64 << 20), device="cuda")
    dw = torch.empty(64, 7, 7, 3, device="cuda")
    dw4 = torch.empty(64, 7, 7, 3, device="cuda")

    def wnew():
        L.check(lib.cilrs_stem_conv_wgrad(L.ptr(x4), L.ptr(dy), L.ptr(dw), L.ptr(sc2), sc2.numel(), N, H, W, st))

    def wold():
        L.check(lib.cilrs_conv2d_wgrad(L.ptr(x4), L.ptr(dy), L.ptr(dw4), L.ptr(sc2), N, H, W, 4, 64, 7, 7,
                                       2, 3, 3, st))
    tn, to = timed(wnew), timed(wold)
    print(f"stem wgrad N={N} {H}x{W}: stem_f32 {tn:7.1f} us ({fl / tn / 1e6:6.1f} TF)   implicit GEMM (Cin 4) "
          f"{to:7.1f} us ({fl / to / 1e6:6.1f} TF)   max |diff| {(dw - dw4).abs().max().item():.2e} "
          f"(max |dw| {dw4.abs().max().item():.1f})", flush=True)

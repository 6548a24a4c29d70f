#!/usr/bin/env python3
"""Winograd F(2x2,3x3) against the implicit-GEMM kernel on the trunk's 3x3 / stride-1 shapes at
the benchmark batch: device time per launch (hipEvents, 20 launches back to back), effective
TFLOP/s in DIRECT-convolution flops, and the filter-transform launch on its own."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch
from cilrs_mi355 import _lib as L

lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, H, W, Cc in (("layer1", 22, 50, 64), ("layer2", 11, 25, 128), ("layer3", 6, 13, 256),
                       ("layer4", 3, 7, 512)):
    x = torch.randn(N, H, W, Cc, device="cuda")
    w = torch.randn(Cc, 3, 3, Cc, device="cuda") / (9 * Cc) ** 0.5
    y = torch.empty(N, H, W, Cc, device="cuda")
    scratch = torch.empty(max(8 * y.numel(), lib.cilrs_conv2d_wino_scratch_floats(Cc, Cc)), device="cuda")
    fl = 2.0 * N * H * W * Cc * 9 * Cc

    def igemm():
        L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, H, W, Cc, Cc, 3, 3, 1, 1, -1, 0,
                                     L.ptr(scratch), scratch.numel(), st))

    def wino_pre():
        L.check(lib.cilrs_conv2d_wino_pre(L.ptr(x), L.ptr(scratch), L.ptr(y), None, N, H, W, Cc, Cc, st))

    def filt():
        L.check(lib.cilrs_wino_filter_transform(L.ptr(w), L.ptr(scratch), Cc, Cc, 0, st))

    def filt_d():
        L.check(lib.cilrs_wino_filter_transform(L.ptr(w), L.ptr(scratch), Cc, Cc, 1, st))

    def wino():
        L.check(lib.cilrs_conv2d_wino_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, H, W, Cc, Cc, L.ptr(scratch), st))
    t_i, t_w = timed(igemm), timed(wino)
    if os.environ.get("WINO_STAMPS"):
        stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
        lib.cilrs_conv2d_wino_stamps(L.ptr(stamps))
        filt(); wino_pre(); torch.cuda.synchronize()
        lib.cilrs_conv2d_wino_stamps(None)
        v = stamps.cpu().tolist()
        for wn, o in (("wave 0 (multiply first)", 0), ("wave 4 (refill first)", 8)):
            print(f"   block 0 {wn}: prologue {v[o]} multiply {v[o+1]} refill {v[o+2]} barrier wait {v[o+3]} "
                  f"epilogue {v[o+4]} K loop {v[o+5]} cycles ({Cc // 8} chunks)")
    t_f, t_fd = timed(filt), timed(filt_d)
    filt()
    t_p = timed(wino_pre)
    print(f"   filter transform {t_f:.1f} us (dgrad form {t_fd:.1f} us), convolution on a ready U {t_p:.1f} us "
          f"({fl / t_p / 1e6:.1f} TF effective)")
    print(f"{name} N={N} {H}x{W}x{Cc}: implicit GEMM {t_i:7.1f} us ({fl / t_i / 1e6:6.1f} TF)   "
          f"Winograd incl. filter transform {t_w:7.1f} us ({fl / t_w / 1e6:6.1f} TF effective)   x{t_i / t_w:.2f}")

# ---- weight gradient: direct kernel (its own split plan + slab reduce) against the Winograd-domain one
print("weight gradient (incl. the slab reduce):")
for name, H, W, Cc in (("layer1", 22, 50, 64), ("layer2", 11, 25, 128), ("layer3", 6, 13, 256),
                       ("layer4", 3, 7, 512)):
    x = torch.randn(N, H, W, Cc, device="cuda")
    dy = torch.randn(N, H, W, Cc, device="cuda")
    dw = torch.empty(Cc, 3, 3, Cc, device="cuda")
    n_d = lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, Cc, Cc, 3, 3, 1, 1)
    n_w = lib.cilrs_conv2d_wino_wgrad_scratch_floats(N, H, W, Cc, Cc)
    sc = torch.empty(max(n_d, n_w, 4), device="cuda")
    fl = 2.0 * N * H * W * Cc * 9 * Cc

    def direct():
        L.check(lib.cilrs_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(sc), N, H, W, Cc, Cc, 3, 3, 1, 1,
                                       Cc, st))

    def wino_w():
        L.check(lib.cilrs_conv2d_wino_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, W, Cc, Cc, L.ptr(sc), sc.numel(), st))
    t_d, t_w = timed(direct), timed(wino_w)
    print(f"{name} N={N} {H}x{W}x{Cc}: direct {t_d:7.1f} us ({fl / t_d / 1e6:6.1f} TF)   Winograd {t_w:7.1f} us "
          f"({fl / t_w / 1e6:6.1f} TF effective, {n_w * 4 / 1e6:.1f} MB of slabs)   x{t_d / t_w:.2f}")

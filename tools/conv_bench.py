#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the real trunk shapes at B=128 (tuning aid).
Usage: python tools/conv_bench.py [--batch 128] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import _lib as L  # noqa: E402

SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad
    ("layer1 3x3", 22, 50, 64, 64, 3, 1, 1),
    ("layer2 3x3", 11, 25, 128, 128, 3, 1, 1),
    ("layer2.0 s2", 22, 50, 64, 128, 3, 2, 1),
    ("layer3 3x3", 6, 13, 256, 256, 3, 1, 1),
    ("layer4 3x3", 3, 7, 512, 512, 3, 1, 1),
    ("layer4.0 s2", 6, 13, 256, 512, 3, 2, 1),
]

R50_SHAPES = [  # the ResNet-50 variant at 176x400 (pool output 44x100); run with --r50 --batch 64
    ("l1 1x1 64>64", 44, 100, 64, 64, 1, 1, 0),
    ("l1 3x3 64", 44, 100, 64, 64, 3, 1, 1),
    ("l1 1x1 64>256", 44, 100, 64, 256, 1, 1, 0),
    ("l1 1x1 256>64", 44, 100, 256, 64, 1, 1, 0),
    ("l2 1x1 256>128", 44, 100, 256, 128, 1, 1, 0),
    ("l2 3x3 s2 128", 44, 100, 128, 128, 3, 2, 1),
    ("l2 1x1 128>512", 22, 50, 128, 512, 1, 1, 0),
    ("l2 1x1 512>128", 22, 50, 512, 128, 1, 1, 0),
    ("l2 3x3 128", 22, 50, 128, 128, 3, 1, 1),
    ("l3 1x1 256>1024", 11, 25, 256, 1024, 1, 1, 0),
    ("l3 1x1 1024>256", 11, 25, 1024, 256, 1, 1, 0),
    ("l3 3x3 256", 11, 25, 256, 256, 3, 1, 1),
    ("l4 1x1 512>2048", 6, 13, 512, 2048, 1, 1, 0),
    ("l4 1x1 2048>512", 6, 13, 2048, 512, 1, 1, 0),
    ("l4 3x3 512", 6, 13, 512, 512, 3, 1, 1),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--wgrad-only", action="store_true")
    ap.add_argument("--r50", action="store_true", help="the ResNet-50 variant's shapes")
    args = ap.parse_args()
    lib = L.lib()
    N = args.batch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, H, W, Cin, Cout, k, s, p in (R50_SHAPES if args.r50 else SHAPES):
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(N, H, W, Cin, device="cuda")
        w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
        y = torch.empty(N, Ho, Wo, Cout, device="cuda")
        dy = torch.randn(N, Ho, Wo, Cout, device="cuda")
        dx = torch.empty(N, H, W, Cin, device="cuda")
        dw = torch.empty(Cout, k, k, Cin, device="cuda")
        scratch = torch.empty(8 * max(y.numel(), dx.numel()), device="cuda")
        wsc = torch.empty(lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, Cin, Cout, k, k, s, p),
                          device="cuda")
        flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        res = []
        fwd_cfgs = [] if args.wgrad_only else [(-1, 0), (0, 1), (1, 1), (2, 1), (2, 2), (0, 3), (3, 1), (4, 1), (5, 1), (3, 2), (4, 2), (5, 2), (3, 4), (4, 4), (5, 3)]
        for cfg, sk in fwd_cfgs:
            if cfg in (0, 3) and Cout % 128:
                continue

            def f():
                L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, H, W, Cin, Cout, k,
                                             k, s, p, cfg, sk, L.ptr(scratch), scratch.numel(), st))
            us = timeit(f, args.iters)
            res.append(f"fwd[c{cfg},k{sk}] {us:7.1f}us {flops / us / 1e6:6.1f}TF")
        for cfg, sk in ([] if args.wgrad_only else [(-1, 0), (1, 1), (2, 1), (2, 2), (3, 1), (4, 1), (5, 1), (4, 2), (5, 2)]):
            if cfg in (0, 3) and Cin % 128:
                continue

            def f():
                L.check(lib.cilrs_conv2d_dgrad(L.ptr(dy), L.ptr(w), L.ptr(dx), None, N, H, W, Cin,
                                               Cout, k, k, s, p, cfg, sk, L.ptr(scratch),
                                               scratch.numel(), st))
            us = timeit(f, args.iters)
            res.append(f"dgrad[c{cfg},k{sk}] {us:7.1f}us {flops / us / 1e6:6.1f}TF")

        def f():
            L.check(lib.cilrs_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(wsc), N, H, W,
                                           Cin, Cout, k, k, s, p, Cin, st))
        us = timeit(f, args.iters)
        res.append(f"wgrad {us:7.1f}us {flops / us / 1e6:6.1f}TF")
        print(f"== {name}: M={N * Ho * Wo} K={k * k * Cin} N={Cout}  {flops / 1e9:.2f} GFLOP")
        for r in res:
            print("   ", r)


if __name__ == "__main__":
    main()

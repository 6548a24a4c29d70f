#!/usr/bin/env python3
"""B=1 forward convs: time of every tile config x split-K (launch + reduce), vs the auto choice."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import _lib as L  # noqa: E402

SHAPES = [("layer1 3x3", 22, 50, 64, 64, 3, 1, 1), ("layer2 3x3", 11, 25, 128, 128, 3, 1, 1),
          ("layer2.0 s2", 22, 50, 64, 128, 3, 2, 1), ("layer3 3x3", 6, 13, 256, 256, 3, 1, 1),
          ("layer3.0 s2", 11, 25, 128, 256, 3, 2, 1), ("layer4 3x3", 3, 7, 512, 512, 3, 1, 1),
          ("layer4.0 s2", 6, 13, 256, 512, 3, 2, 1)]


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


lib = L.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, H, W, Cin, Cout, k, s, p in SHAPES:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, Ho, Wo, Cout, device="cuda")
    scratch = torch.empty(64 * y.numel() + (1 << 20), device="cuda")
    out = []
    for cfg in (-1, 2, 1, 0):
        if cfg == 0 and Cout % 128:
            continue
        for sk in ((0,) if cfg == -1 else (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48)):
            def f():
                L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, H, W, Cin, Cout, k, k,
                                             s, p, cfg, sk, L.ptr(scratch), scratch.numel(), st))
            try:
                out.append((timeit(f), cfg, sk))
            except RuntimeError:
                pass
    auto = [o for o in out if o[1] == -1][0][0]
    best = sorted(o for o in out if o[1] != -1)[:4]
    print(f"{name:12s} M={N * Ho * Wo:5d} K={k * k * Cin:5d} N={Cout:4d}  auto {auto:6.1f}us | best " +
          "  ".join(f"c{c},k{sk}:{t:5.1f}" for t, c, sk in best))

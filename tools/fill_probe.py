#!/usr/bin/env python3
"""How fast is each implicit-GEMM tile configuration when the grid fills the chip EXACTLY (no
block-count quantisation, no partial last round)?  Separates what a tile shape loses inside a CU
from what a launch loses to quantisation / drain.  3x3 s1 convs, 16x16 frames (256 pixels), batch
chosen so that tiles = rounds x 256 CUs x blocks-per-CU."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import _lib as L  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # cfg -> (BM, BN, blocks/CU)
    tiles = {0: (128, 128, 2), 1: (128, 64, 2), 2: (64, 64, 5)}
    for C_ in (64, 128, 256):
        for cfg, (bm, bn, occ) in tiles.items():
            if C_ % bn:
                continue
            for rounds in (1, 2, 4):
                ntile_m = rounds * 256 * occ // (C_ // bn)
                M = ntile_m * bm
                N = M // 256
                if N * 256 != M or N < 1:
                    continue
                x = torch.randn(N, 16, 16, C_, device="cuda")
                w = torch.randn(C_, 3, 3, C_, device="cuda") * 0.05
                y = torch.empty(N, 16, 16, C_, device="cuda")
                flops = 2.0 * M * C_ * 9 * C_
                for mode in ("fwd", "dgrad"):
                    def f():
                        if mode == "fwd":
                            L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, 16, 16,
                                                         C_, C_, 3, 3, 1, 1, cfg, 1, None, 0, st))
                        else:
                            L.check(lib.cilrs_conv2d_dgrad(L.ptr(x), L.ptr(w), L.ptr(y), None, N,
                                                           16, 16, C_, C_, 3, 3, 1, 1, cfg, 1,
                                                           None, 0, st))
                    us = timeit(f)
                    print(f"C={C_:3d} cfg{cfg} {bm}x{bn} occ{occ} rounds={rounds} tiles="
                          f"{ntile_m * (C_ // bn):5d} M={M:6d} {mode:5s} {us:8.1f}us "
                          f"{flops / us / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()

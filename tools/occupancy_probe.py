#!/usr/bin/env python3
"""How does one implicit-GEMM launch scale with the NUMBER of 64x64 tiles below and above one
full round (256 CUs x 5 co-resident blocks = 1,280)?  If the dispatcher spread a partial round
evenly over the CUs, a launch of 640 tiles would run at half the co-residency and finish early; if
it packs blocks CU by CU, time stays at the full-round value.  3x3 s1 conv, 8x8 frames, C = 256
(72 K-tiles of 32)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import _lib as L  # noqa: E402
from fill_probe import timeit  # noqa: E402


def main():
    cfg = int(os.environ.get("CFG", "2"))          # 2: 64x64 register-staged; 5: 64x64 LDS-DMA; 0: 128x128
    bm = 128 if cfg in (0, 3) else 64
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for C_ in (256, 64):
        kt = 9 * C_ // 32
        for tiles in (64, 128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1536, 1920, 2560):
            if bm == 128 and (C_ < 128 or tiles > 1024):
                continue
            N = tiles // (C_ // bm) * (bm // 64)
            M = N * 64
            x = torch.randn(N, 8, 8, C_, device="cuda")
            w = torch.randn(C_, 3, 3, C_, device="cuda") * 0.05
            y = torch.empty(N, 8, 8, C_, device="cuda")
            flops = 2.0 * M * C_ * 9 * C_

            def f():
                L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y), N, 8, 8, C_, C_, 3, 3,
                                             1, 1, cfg, 1, None, 0, st))
            us = timeit(f)
            print(f"C={C_:3d} tiles={tiles:5d} ({tiles / 256:5.2f} per CU) {us:8.1f}us "
                  f"{flops / us / 1e6:6.1f} TF  {(us - 12.0) / kt:6.3f} us per K-tile", flush=True)


if __name__ == "__main__":
    main()

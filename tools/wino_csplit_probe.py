#!/usr/bin/env python3
"""Winograd forward with the input channels split over several workgroups per tile
(cilrs_conv2d_wino_split: the library's cost model picks the split) against the implicit GEMM, on
the under-filled trunk shapes: layer3 (168 blocks at B=128) and layer4 (128 blocks, 3x7 maps -- 2x4
Winograd tiles cover 32 pixels for 21).  Prints the split taken and both times (filters already
transformed)."""
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


for (N, H, W, Cc) in ((128, 6, 13, 256), (128, 3, 7, 512)):
    x = torch.randn(N, H, W, Cc, device="cuda")
    w = torch.randn(Cc, 3, 3, Cc, device="cuda") / (9 * Cc) ** 0.5
    U = torch.empty(16 * Cc * Cc, device="cuda")
    y = torch.empty(N, H, W, Cc, device="cuda")
    y2 = torch.empty_like(y)
    part = torch.empty(2 * Cc * 4096, device="cuda")
    slabs = torch.empty(4 * N * H * W * Cc, device="cuda")
    scratch = torch.empty(64 << 20, device="cuda")
    cs, rows = C.c_int(0), C.c_int(0)
    L.check(lib.cilrs_wino_filter_transform(L.ptr(w), L.ptr(U), Cc, Cc, 0, st))

    def wino():
        L.check(lib.cilrs_conv2d_wino_split(L.ptr(x), L.ptr(U), L.ptr(y), None, L.ptr(part), N, H, W,
                                            Cc, Cc, L.ptr(slabs), slabs.numel(), C.byref(cs),
                                            C.byref(rows), st))

    def igemm():
        L.check(lib.cilrs_conv2d_fwd(L.ptr(x), L.ptr(w), L.ptr(y2), N, H, W, Cc, Cc, 3, 3, 1, 1, -1, 0,
                                     L.ptr(scratch), scratch.numel(), st))

    def filt():
        L.check(lib.cilrs_wino_filter_transform(L.ptr(w), L.ptr(U), Cc, Cc, 0, st))
    tw, ti, tf = timed(wino), timed(igemm), timed(filt)
    err = (y - y2).abs().max().item()
    fl = 2.0 * N * H * W * Cc * Cc * 9
    print(f"N={N} {H}x{W}x{Cc}: Winograd csplit={cs.value} rows={rows.value} {tw:7.1f} us "
          f"({fl / tw / 1e6:6.1f} TF effective), implicit GEMM {ti:7.1f} us ({fl / ti / 1e6:6.1f} TF), "
          f"filter transform {tf:5.1f} us, max |diff| {err:.2e}", flush=True)

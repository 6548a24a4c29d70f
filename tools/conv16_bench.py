#!/usr/bin/env python3
"""Per-shape timing of the bf16 training convolution (cilrs_conv2d_train_16, forward form with
BatchNorm partials) under the tile plans of csrc/conv16.hip.  The plan switch is read once per
process, so every setting runs in a child process.
Usage: conv16_bench.py [resnet34|resnet50] [tile settings, e.g. 0 1 2 auto]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))

SHAPES = {
    # N, H, W, Cin, Cout, k, stride, pad
    "resnet34": [(128, 22, 50, 64, 64, 3, 1, 1), (128, 22, 50, 64, 128, 3, 2, 1),
                 (128, 11, 25, 128, 128, 3, 1, 1), (128, 11, 25, 128, 256, 3, 2, 1),
                 (128, 6, 13, 256, 256, 3, 1, 1), (128, 6, 13, 256, 512, 3, 2, 1),
                 (128, 3, 7, 512, 512, 3, 1, 1), (128, 22, 50, 64, 128, 1, 2, 0)],
    "resnet50": [(64, 44, 100, 64, 64, 1, 1, 0), (64, 44, 100, 64, 64, 3, 1, 1),
                 (64, 44, 100, 64, 256, 1, 1, 0), (64, 44, 100, 256, 64, 1, 1, 0),
                 (64, 44, 100, 256, 128, 1, 1, 0), (64, 44, 100, 128, 128, 3, 2, 1),
                 (64, 22, 50, 128, 512, 1, 1, 0), (64, 22, 50, 512, 128, 1, 1, 0),
                 (64, 22, 50, 128, 128, 3, 1, 1), (64, 22, 50, 512, 256, 1, 1, 0),
                 (64, 11, 25, 256, 256, 3, 1, 1), (64, 11, 25, 256, 1024, 1, 1, 0),
                 (64, 11, 25, 1024, 256, 1, 1, 0), (64, 11, 25, 1024, 512, 1, 1, 0),
                 (64, 6, 13, 512, 512, 3, 1, 1), (64, 6, 13, 512, 2048, 1, 1, 0),
                 (64, 6, 13, 2048, 512, 1, 1, 0)],
}


def child(net):
    import torch
    from cilrs_mi355 import _lib as L
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (N, H, W, Cin, Cout, k, s, p) in SHAPES[net]:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        M = N * Ho * Wo
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / (k * k * Cin) ** 0.5).to(torch.bfloat16)
        y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
        part = torch.empty(2 * Cout * ((M + 63) // 64), device="cuda")
        rows = C.c_int(0)

        def run():
            L.check(lib.cilrs_conv2d_train_16(L.ptr(x), L.ptr(w), L.ptr(y), None, None, L.ptr(part),
                                              None, None, None, 0, None, N, H, W, Cin, Ho, Wo, Cout,
                                              k, s, p, 0, 1, C.byref(rows), st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * M * Cout * k * k * Cin
        by = 2.0 * (N * H * W * Cin + M * Cout + Cout * k * k * Cin)
        print(f"RESULT {N}x{H}x{W} {Cin}->{Cout} k{k}s{s} tile_rows={(M + rows.value - 1) // rows.value if rows.value else 0:4d} "
              f"{us:8.1f} us {fl / us / 1e6:7.1f} TF {by / us / 1e3:7.1f} GB/s", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    net = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    settings = sys.argv[2:] or ["0", "1", "auto"]
    for st in settings:
        env = dict(os.environ)
        env.pop("CILRS_CONV16_TILE", None)
        if st != "auto":
            env["CILRS_CONV16_TILE"] = st
        print(f"== {net}, CILRS_CONV16_TILE={st}", flush=True)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", net], env=env,
                             capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("RESULT"):
                print("  " + line[7:], flush=True)
        if out.returncode != 0:
            print(out.stderr[-2000:])


if __name__ == "__main__":
    main()

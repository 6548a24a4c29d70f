#!/usr/bin/env python3
"""Per-kernel device time of one train step in the fp32 and bf16 modes (hipEvent brackets, serial).
Usage: bf16_train_probe.py [resnet50|resnet34] [batch] [H W]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import CILRS, CILRSResNet50, CONFIG_A, Trainer  # noqa: E402


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ((176, 400) if net == "resnet50" else (88, 200))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    batch = (torch.randn(B, 3, H, W, generator=g).to(dev), torch.rand(B, generator=g).to(dev),
             torch.randint(0, 4, (B,), generator=g).to(dev), torch.rand(B, 3, generator=g).to(dev))
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        m = (CILRSResNet50 if net == "resnet50" else CILRS)(4, 0.0).to(dev)
        tr = Trainer(m, CONFIG_A, precision=prec)
        for _ in range(3):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        pl = tr.eng.plan(B, H, W)
        pl.profile_reset(); pl.profile(True)
        for _ in range(3):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        t = pl.profile_table(); pl.profile(False)
        fam = {}
        for k, v in t.items():
            f = k.split(".")[0]
            a = fam.setdefault(f, [0.0, 0.0, 0])
            a[0] += v["ms"] / 3; a[1] += v["flops"] / 3; a[2] += v["calls"] // 3
        print(f"== {net} B={B} {H}x{W} {prec}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} frames/s; serial device sum "
              f"{sum(a[0] for a in fam.values()):.3f} ms")
        for f, a in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            tf = f" {a[1] / a[0] / 1e9:7.1f} TF" if a[1] else ""
            print(f"   {f:12s} {a[0]:7.3f} ms  {a[2]:3d} launches{tf}")
        for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms"])[:14]:
            print(f"      {k:22s} {v['ms'] / 3:7.3f} ms {v['calls'] // 3:3d}x")
        del tr, m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

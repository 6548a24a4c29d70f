#!/usr/bin/env python3
"""Per-label device time of the B=64 fp16-trunk forward (hipEvent brackets in the plan)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch
from cilrs_mi355 import CILRS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = CILRS().cuda().eval()
eng = m.engine()
u8 = torch.randint(0, 256, (B, 88, 200, 3), dtype=torch.uint8, device="cuda")
spd = torch.rand(B, device="cuda"); cmd = torch.randint(0, 4, (B,), device="cuda")
for half in (False, True):
    for _ in range(3):
        eng.run_forward_u8(u8, spd, cmd, half=half)
    pl = eng.plan(B, 88, 200)
    pl.profile_reset(); pl.profile(True)
    for _ in range(10):
        eng.run_forward_u8(u8, spd, cmd, half=half)
    torch.cuda.synchronize()
    t = pl.profile_table(); pl.profile(False)
    print("half" if half else "fp32", {k: round(v["ms"] / 10 * 1e3, 1) for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms"])},
          "total", round(sum(v["ms"] for v in t.values()) / 10 * 1e3, 1), "us")

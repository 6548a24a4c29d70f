#!/usr/bin/env python3
"""The train step on torch's default (null) stream against a stream of its own: does the legacy
default stream's implicit synchronisation cost anything?  Three rounds, round-robin."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cilrs-autonomous-driving-carla_amd"))
import torch
from cilrs_mi355 import CILRS, CONFIG_A, Trainer

B = 128
img = torch.randn(B, 3, 88, 200, device="cuda")
spd = torch.rand(B, device="cuda")
cmd = torch.randint(0, 4, (B,), device="cuda")
tgt = torch.rand(B, 3, device="cuda")
m = CILRS(4, 0.0).cuda()
tr = Trainer(m, CONFIG_A)
own = torch.cuda.Stream()


def run(steps=50):
    for _ in range(5):
        tr.train_step(img, spd, cmd, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(img, spd, cmd, tgt)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for rep in range(3):
    a = run()
    with torch.cuda.stream(own):
        b = run()
    print(f"default stream {a:.3f} ms/step   own stream {b:.3f} ms/step", flush=True)

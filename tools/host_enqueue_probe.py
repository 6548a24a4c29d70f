#!/usr/bin/env python3
"""Host time to ENQUEUE a train step against the device time of the step, single-process and through
the data-parallel path at world size 1: is the step ever waiting for the host?"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cilrs-autonomous-driving-carla_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch
import torch.distributed as dist
from cilrs_mi355 import CILRS, CONFIG_A, Trainer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
B = 128
img = torch.randn(B, 3, 88, 200, device="cuda")
spd = torch.rand(B, device="cuda")
cmd = torch.randint(0, 4, (B,), device="cuda")
tgt = torch.rand(B, 3, device="cuda")
for dp in (False, True):
    m = CILRS(4, 0.0).cuda()
    tr = Trainer(m, CONFIG_A, process_group=dist.group.WORLD if dp else None)
    for _ in range(10):
        tr.train_step(img, spd, cmd, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.train_step(img, spd, cmd, tgt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # two steps from an idle device: nothing can push back on the host
    h = []
    for _ in range(5):
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        tr.train_step(img, spd, cmd, tgt)
        tr.train_step(img, spd, cmd, tgt)
        h.append((time.perf_counter() - s0) / 2)
    torch.cuda.synchronize()
    print(f"{'data-parallel (world 1)' if dp else 'single process        '}: host enqueue {1e3*(t1-t0)/50:.3f} ms/step "
          f"over 50 steps (queue back-pressure included), {1e3*min(h):.3f} ms/step from an idle device; "
          f"wall {1e3*(t2-t0)/50:.3f} ms/step", flush=True)
dist.destroy_process_group()

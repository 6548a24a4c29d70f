#!/usr/bin/env python3
"""Per-label hipEvent times of the BatchNorm launches of a B=128 train step (the plan's own
profile table): `CILRS_BN_FUSED=0 python tools/bn_profile.py` for the separate-finalize form."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch
from cilrs_mi355 import CILRS, Trainer, TrainConfig

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = CILRS().cuda()
tr = Trainer(m, TrainConfig())
batch = [torch.randn(B, 3, 88, 200, device="cuda"), torch.rand(B, device="cuda"),
         torch.randint(0, 4, (B,), device="cuda"), torch.rand(B, 3, device="cuda")]
for _ in range(3):
    tr.train_step(*batch)
pl = tr.eng.plan(B, 88, 200)
pl.profile_reset()
pl.profile(True)
for _ in range(5):
    tr.train_step(*batch)
torch.cuda.synchronize()
t = pl.profile_table()
pl.profile(False)
tot = 0.0
for k, r in sorted(t.items()):
    if k.startswith("bn_"):
        tot += r["ms"] / 5
        print(f"{k:20s} calls/step {r['calls'] // 5:3d}  us/call {r['ms'] / r['calls'] * 1e3:7.2f}")
print(f"BatchNorm total {tot:.3f} ms/step  (CILRS_BN_FUSED={os.environ.get('CILRS_BN_FUSED', '0')})")

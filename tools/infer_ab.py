"""Single-frame latency under different routing of the B=1 convolutions (env switches are read
once per process: run one configuration per invocation)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "cilrs-autonomous-driving-carla_amd"))
import numpy as np, torch
from cilrs_mi355 import CILRS
from cilrs_mi355.predict import Predictor
torch.manual_seed(0)
m = CILRS(4, 0.0).cuda().eval()
pr = Predictor(m)
frame = np.random.default_rng(0).integers(0, 256, (88, 200, 3), dtype=np.uint8)
for _ in range(50): pr.predict_controls(frame, 25.0, 0)
lat = []
for _ in range(2000):
    t = time.perf_counter(); pr.predict_controls(frame, 25.0, 0); lat.append((time.perf_counter() - t) * 1e3)
lat.sort()
pl = m.engine().plan(1, 88, 200); pl.profile_reset(); pl.profile(True)
for _ in range(10): pr.predict_controls(frame, 25.0, 0)
torch.cuda.synchronize(); t = pl.profile_table(); pl.profile(False)
dev = {k: round(v["ms"] / 10 * 1e3, 1) for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms"])}
print(os.environ.get("CILRS_SMALL_BLOCKS"), os.environ.get("CILRS_SMALL_K"), "median %.4f p99 %.4f ms" % (lat[1000], lat[1980]), dev, flush=True)

#!/usr/bin/env python3
"""Input-pipeline throughput: JPEG decode threads -> pinned batch -> H2D -> fused augmentation, with
and without a train step consuming the batches.  Writes a synthetic dataset in the reference's
on-disk format to a temp dir first.  Usage: python tools/loader_bench.py [--frames 4096]"""
import argparse
import csv
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from cilrs_mi355 import CILRS, CONFIG_A, Trainer  # noqa: E402
from cilrs_mi355 import data as D  # noqa: E402

COLS = ["frame", "image_filename", "steer", "throttle", "brake", "speed_kmh", "speed_normalized",
        "high_level_command", "command_name", "position_x", "position_y", "position_z", "yaw",
        "timestamp"]


def measure(frames=4096, batch=128, workers=None, threads=False, trainer=None, log=print):
    """Writes `frames` synthetic 200x88 JPEGs in the reference's on-disk format (a session folder
    with measurements.csv + images/, data/collect_data.py:545-564, 683-716) to a temp dir, then
    times (a) the loader alone -- process-pool JPEG decode -> pinned batch -> H2D -> fused
    augmentation kernel -- and (b) the loader feeding the fused train step.  Returns a dict."""
    if workers is None:
        workers = min(16, len(os.sched_getaffinity(0)))
    rng = np.random.default_rng(0)
    names = ["LANEFOLLOW", "LEFT", "RIGHT", "STRAIGHT"]
    res = {"frames": frames, "batch": batch, "workers": workers,
           "pool": "threads" if threads else "processes"}
    with tempfile.TemporaryDirectory() as root:
        sdir = os.path.join(root, "session1")
        os.makedirs(os.path.join(sdir, "images"))
        t0 = time.perf_counter()
        with open(os.path.join(sdir, "measurements.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(COLS)
            for k in range(frames):
                base = rng.integers(0, 256, (22, 50, 3), dtype=np.uint8)
                img = Image.fromarray(base).resize((200, 88), Image.BILINEAR)
                fn = f"frame_{k:08d}.jpg"
                img.save(os.path.join(sdir, "images", fn), quality=95)
                c = int(rng.choice(4, p=[0.5, 0.27, 0.14, 0.09]))
                wr.writerow([k, fn, 0.1, 0.5, 0.0, 30.0, 0.333333, c, names[c], 0, 0, 0, 0, k * 0.05])
        log(f"wrote {frames} JPEGs in {time.perf_counter() - t0:.1f}s")
        s = D.Sessions(root)
        dev = torch.device("cuda")
        idx = np.arange(len(s))
        ld = D.BatchLoader(s, idx, batch, dev, train=True, seed=1, workers=workers,
                           processes=not threads)
        for _ in ld:                  # warm the page cache and the kernels
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for img, spd, cmd, tgt in ld:
            n += img.size(0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["loader_frames_per_s"] = round(n / dt, 1)
        log(f"loader alone: {n / dt:,.0f} frames/s ({workers} decode "
            f"{'threads' if threads else 'processes'}, B={batch})")
        if trainer is None:
            m = CILRS(4, dropout=0.0).to(dev)
            trainer = Trainer(m, CONFIG_A)
        for img, spd, cmd, tgt in ld:
            trainer.train_step(img, spd, cmd, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for img, spd, cmd, tgt in ld:
            trainer.train_step(img, spd, cmd, tgt)
            n += img.size(0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["loader_plus_train_frames_per_s"] = round(n / dt, 1)
        log(f"loader + train step: {n / dt:,.0f} frames/s")
        # the augmentation kernel alone
        fr = torch.randint(0, 256, (batch, 88, 200, 3), dtype=torch.uint8, device=dev)
        p = D.draw_aug_params(np.random.default_rng(3), batch)
        for _ in range(3):
            D.augment_u8(fr, p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            D.augment_u8(fr, p)
        e1.record()
        torch.cuda.synchronize()
        ld.close()
        res["augment_us_per_batch"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        log(f"augment_u8 (incl. parameter upload), B={batch}: {res['augment_us_per_batch']:.1f} us")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--workers", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--threads", action="store_true", help="thread pool instead of processes")
    args = ap.parse_args()
    import json
    res = measure(args.frames, args.batch, args.workers, args.threads,
                  log=lambda m: print(m, flush=True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

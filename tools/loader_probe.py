#!/usr/bin/env python3
"""Where the loader's time goes: raw decode pool throughput vs staging vs the full iterator."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from cilrs_jpeg_worker import decode_chunk  # noqa: E402


def main():
    import multiprocessing as mp
    rng = np.random.default_rng(0)
    root = tempfile.mkdtemp()
    paths = []
    for k in range(4096):
        base = rng.integers(0, 256, (22, 50, 3), dtype=np.uint8)
        p = os.path.join(root, f"f{k}.jpg")
        Image.fromarray(base).resize((200, 88), Image.BILINEAR).save(p, quality=95)
        paths.append(p)
    print("cpus:", len(os.sched_getaffinity(0)), "file KB:", os.path.getsize(paths[0]) / 1024, flush=True)
    t = time.perf_counter()
    decode_chunk((paths[:512], 88, 200))
    print(f"single process decode: {512 / (time.perf_counter() - t):.0f} frames/s", flush=True)
    chunks = [(paths[i:i + 8], 88, 200) for i in range(0, len(paths), 8)]
    for nw in (4, 8, 16):
        with mp.get_context("spawn").Pool(nw) as pp:
            list(pp.imap(decode_chunk, chunks[:32]))
            t = time.perf_counter()
            list(pp.imap(decode_chunk, chunks))
            print(f"{nw} procs: {len(paths) / (time.perf_counter() - t):.0f} frames/s", flush=True)
    import torch
    t = time.perf_counter()
    for _ in range(10):
        h = torch.empty(128, 88, 200, 3, dtype=torch.uint8).pin_memory()
    print(f"pin_memory alloc: {(time.perf_counter() - t) / 10 * 1e3:.2f} ms per batch", flush=True)
    d = torch.empty(128, 88, 200, 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    print(f"H2D 6.7 MB: {(time.perf_counter() - t) / 10 * 1e3:.2f} ms", flush=True)
    import shutil
    shutil.rmtree(root)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""BASELINE configs[4]: five 64-frame streams through the fp16 trunk.  Sequential on one plan
(what round 1 measured) vs. five lanes on five HIP streams (eager and hipGraph replay) vs. one
B=320 batch.  Frames are resident uint8 tensors; outputs stay on the device."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import torch  # noqa: E402
from cilrs_mi355 import CILRS  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = CILRS(4, 0.0).to(dev).eval()
    eng = m.engine()
    half = os.environ.get("HALF", "f16")
    half = True if half == "f16" else (False if half == "f32" else half)
    u = torch.randint(0, 256, (5, 64, 88, 200, 3), dtype=torch.uint8, device=dev)
    spd = torch.rand(64, device=dev)
    cmd = torch.randint(0, 4, (64,), device=dev)
    reps = 20

    def timed(fn, sync):
        for _ in range(3):
            fn()
        sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        return (time.perf_counter() - t) / reps

    def seq():
        for i in range(5):
            eng.run_forward_u8(u[i], spd, cmd, half=half)
    dt = timed(seq, lambda: torch.cuda.synchronize(dev))
    print(f"sequential, one plan      : {dt * 1e3:7.3f} ms per 320 frames  {320 / dt:9.0f} frames/s")

    streams = [torch.cuda.Stream(device=dev) for _ in range(5)]
    outs = [(torch.empty(64, 3, device=dev), torch.empty(64, device=dev)) for _ in range(5)]
    for graph in (False, True):
        def lanes():
            for i in range(5):
                with torch.cuda.stream(streams[i]):
                    eng.run_forward_u8(u[i], spd, cmd, out=outs[i], graph=graph, half=half,
                                       lane=i + 1)

        def sync():
            for s in streams:
                s.synchronize()
        dt = timed(lanes, sync)
        print(f"5 lanes on 5 streams{' (graph)' if graph else '        '}: {dt * 1e3:7.3f} ms per 320 frames  "
              f"{320 / dt:9.0f} frames/s")
    # reference: the lanes give the same numbers as the single plan
    a = eng.run_forward_u8(u[2], spd, cmd, half=half)[0]
    torch.cuda.synchronize(dev)
    print("lane 3 == plan 0:", bool(torch.equal(a, outs[2][0])))
    u320 = u.view(320, 88, 200, 3)
    spd320, cmd320 = spd.repeat(5), cmd.repeat(5)
    dt = timed(lambda: eng.run_forward_u8(u320, spd320, cmd320, half=half),
               lambda: torch.cuda.synchronize(dev))
    print(f"one B=320 batch           : {dt * 1e3:7.3f} ms per 320 frames  {320 / dt:9.0f} frames/s")


if __name__ == "__main__":
    main()

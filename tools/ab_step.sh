#!/bin/bash
# A/B aid: selected GPU tests, then the default bench line (overlapped) and a serialised one; prints
# step time, roofline.frac and the per-kernel table of both.  Usage: ab_step.sh <tag> [pytest -k expr]
T=${1:-ab}; K=${2:-"wino or b128 or train_steps_golden or fused_backward"}
O=gpurun_out/$T; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "$K" > $O/tests.log 2>&1
tail -4 $O/tests.log
python bench.py --no-cpu-baseline --no-infer --no-loader > $O/bench.log 2> $O/bench.err || exit 1
CILRS_OVERLAP=0 python bench.py --no-cpu-baseline --no-infer --no-loader --steps 20 > $O/bench_serial.log 2> $O/bench_serial.err || exit 1
python - <<P
import json
for f in ["$O/bench.log", "$O/bench_serial.log"]:
    d = [json.loads(l) for l in open(f) if l.startswith("{")][-1]
    print(f, d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
    print("  ", {k: v["ms_per_step"] for k, v in d["kernels_by_layer"].items()})
    for k in ("train_bf16", "resnet50_train_bf16", "resnet50_train_f32"):
        if k in d: print("  ", k, d[k].get("ms_per_step"))
P

#!/usr/bin/env python3
"""Fold the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into traffic.json.
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads -- MI355X_MICROARCH.md
'HBM'); both counters are in KiB."""
import csv
import glob
import json
import sys

out = sys.argv[1]


def load(sub, counter):
    import os
    f = max(glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"), key=os.path.getmtime)   # newest run
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        # "igemm" = the forward + data-gradient family bench.py's roofline object is about: the
        # implicit-GEMM kernel and (round 3) the Winograd kernel that took over layers 1-3; the two
        # are also listed on their own
        fams = ["igemm", "igemm_only"] if "conv_igemm_kernel" in n else \
            ["igemm", "wino_only"] if "conv_wino" in n else \
            ["igemm", "stem_only"] if "stem_f32_kernel" in n else \
            ["wgrad", "wgrad_direct_only"] if "conv_wgrad_kernel" in n else \
            ["wgrad", "wgrad_stem_only"] if "stem_wgrad_f32_kernel" in n else \
            ["wgrad", "wgrad_wino_only"] if "wino_wgrad_kernel" in n else []
        # a Winograd convolution may be two kernel launches (64-tile blocks + the 16-tile tail,
        # conv_wino_q_kernel): the tail's bytes count, the tail is not a launch of its own -- so
        # "per launch" stays "per convolution", like bench.py's flops_per_launch
        is_tail = "conv_wino_q_kernel" in n
        for fam in fams:
            a = agg.setdefault(fam, [0, 0.0])
            a[0] += 0 if is_tail else 1
            a[1] += float(r["Counter_Value"])
    return agg


fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on `bench.py --steps 3 --warmup 1` "
                 "(tools/pmc_traffic.sh); FETCH_SIZE x2 per the gfx950 correction; KiB -> bytes; "
                 "igemm = forward + data-gradient family (conv_igemm_kernel + conv_wino_kernel + stem_f32_kernel)",
       "commit": sys.argv[2] if len(sys.argv) > 2 else "unknown"}
for fam in fe:
    n = fe[fam][0]
    rd = 2.0 * fe[fam][1] * 1024.0 / n
    wb = wr[fam][1] * 1024.0 / max(wr[fam][0], 1)
    res[fam] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wb),
                "bytes_per_launch": round(rd + wb)}
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))

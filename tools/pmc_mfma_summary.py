#!/usr/bin/env python3
"""Per kernel family: MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x
1024 SIMDs).  GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counter over all 1,024 SIMDs
(checked: a pure-MFMA probe reads 97 %)."""
import csv
import glob
import json
import sys

out = sys.argv[1]
f = glob.glob(f"{out}/pmc/*/*_counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv_wino_q_kernel" in n:
        fam = "conv_wino_q_kernel"
    elif "conv_wino_kernel<true>" in n:
        fam = "conv_wino_kernel_dgrad_epilogue"          # (the epilogue-prefetch variant, also counted below)
    elif "conv_wino_kernel" in n:
        fam = "conv_wino_kernel"
    elif "stem_wgrad_f32_kernel" in n:
        fam = "stem_wgrad_f32_kernel"
    elif "stem_f32_kernel" in n:
        fam = "stem_f32_kernel"
    elif "wino_wgrad_kernel" in n:
        fam = "wino_wgrad_kernel"
    elif "conv_igemm_kernel<128" in n:
        fam = "igemm_128x128"
    elif "conv_igemm_kernel<64" in n:
        fam = "igemm_64x64"
    elif "conv_wgrad_kernel<128" in n:
        fam = "wgrad_128"
    elif "conv_wgrad_kernel<64" in n:
        fam = "wgrad_64"
    else:
        continue
    key = (fam, r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if fam == "conv_wino_kernel_dgrad_epilogue":
        agg.setdefault(("conv_wino_kernel", r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
res = {}
for (fam, _), c in agg.items():
    if "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"] < 8 * 20000:     # skip the heads' tiny launches
        continue
    a = res.setdefault(fam, {"launches": 0, "mfma_busy": 0.0, "gui": 0.0, "wave": 0.0, "wait": 0.0,
                             "valu": 0.0, "mops": 0.0})
    a["launches"] += 1
    a["mfma_busy"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a["gui"] += c["GRBM_GUI_ACTIVE"]
    a["wave"] += c.get("SQ_WAVE_CYCLES", 0.0)
    a["wait"] += c.get("SQ_WAIT_INST_ANY", 0.0)
    a["valu"] += c.get("SQ_ACTIVE_INST_VALU", 0.0)
    a["mops"] += c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
summary = {}
for fam, a in sorted(res.items()):
    summary[fam] = {"launches": a["launches"],
                    "mfma_busy_frac": round(a["mfma_busy"] / (a["gui"] / 8.0 * 1024.0), 4),
                    "wave_wait_inst_frac": round(a["wait"] / max(a["wave"], 1.0), 4),
                    "valu_active_per_simd_cycle": round(a["valu"] / (a["gui"] / 8.0 * 1024.0), 4),
                    "avg_gui_cycles_per_launch": round(a["gui"] / 8.0 / a["launches"], 1),
                    "mfma_mops_f32_per_launch": round(a["mops"] / a["launches"], 1)}
json.dump(summary, open(f"{out}/mfma_util.json", "w"), indent=1)
print(json.dumps(summary, indent=1))

#!/bin/bash
# Stall attribution of the conv kernels: VMEM / LDS queue levels and FIFO-full cycles (two PMC
# passes over the serialised bench step).  Writes gpurun_out/stall/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stall
rm -rf $OUT; mkdir -p $OUT
export CILRS_OVERLAP=0
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-loader --profile-steps 0"
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, json
for sub in "abc":
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs:
        print(sub, "no output"); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        fam = ("wino_q" if "conv_wino_q_kernel" in n else "wino" if "conv_wino_kernel" in n
               else "wino_wgrad" if "wino_wgrad_kernel" in n
               else "igemm128" if "conv_igemm_kernel<128" in n else "igemm64" if "conv_igemm_kernel<64" in n
               else "wgrad128" if "conv_wgrad_kernel<128" in n else "wgrad64" if "conv_wgrad_kernel<64" in n else None)
        if fam is None: continue
        agg.setdefault((fam, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    tot = {}
    for (fam, _), c in agg.items():
        if c.get("GRBM_GUI_ACTIVE", 0) < 8 * 20000: continue
        t = tot.setdefault(fam, {"n": 0})
        t["n"] += 1
        for k, v in c.items(): t[k] = t.get(k, 0.0) + v
    for fam, t in sorted(tot.items()):
        gui = t["GRBM_GUI_ACTIVE"] / 8.0
        print(sub, fam, "launches", t["n"], {k: round(v / gui, 3) for k, v in t.items() if k not in ("n", "GRBM_GUI_ACTIVE")})
PY

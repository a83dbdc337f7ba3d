#!/bin/bash
# GPU probe: SQ counters + effective clock of the ping-pong GEMM variants on one long-K shape (run through gpurun).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pp_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
VARS=${1:-8,12}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq -- python $R/tools/pp_variants.py $VARS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -- python $R/tools/pp_variants.py $VARS > $OUT/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:90] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        if "Start_Timestamp" in r and r["Dispatch_Id"] not in dur: pass
    tr = glob.glob("$OUT/%s/*/*_kernel_trace.csv" % d)
    tdur = collections.defaultdict(list)
    if tr:
        for r in csv.DictReader(open(tr[0])):
            tdur[r["Kernel_Name"][:90] + " grid=" + r.get("Grid_Size", "?")].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in agg.items():
        if "gemm_bf16_pp" not in k: continue
        nl = len(n[k]); gui = v["GRBM_GUI_ACTIVE"] / 8 / nl
        us = sum(tdur[k]) / max(len(tdur[k]), 1) / 1e3
        print(k); print("   launches", nl, "gpu cycles/launch %.0f" % gui, "avg us %.1f" % us, "=> clock %.2f GHz" % (gui / us / 1e3 if us else 0))
        for c, x in sorted(v.items()):
            print("   %-34s %.4g per launch; per CU-cycle %.4f" % (c, x / nl, x / nl / (gui * 256)))
PY

#!/bin/bash
# Counter passes of the headline step's SANM block kernel (run on the GPU box through gpurun): memory-side bytes, L2 hit rate, SQ wait shares.
# Each pass is its own run with --kernel-trace only (gpurun refuses --pmc combined with the other trace domains).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcb
rm -rf $OUT; mkdir -p $OUT
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- $B > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
python $R/tools/summarize_pmc.py $OUT $OUT/hbm_traffic.json > $OUT/summarize_pmc.log 2>&1
python $R/tools/summarize_sq_pmc.py $OUT/pmc_sq $OUT/mfma_util.json > $OUT/summarize_sq.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in acc.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    if h + m > 0: print(f"{k:60s} L2 hit rate {h / (h + m):.3f}  requests {h + m:.3e}")
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/hbm_traffic.json | head -40; cat $OUT/mfma_util.json | head -30

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the block kernel under a few switches (same box, same run): does the XCD's L2 merge the cluster's reads of an exchanged line?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcab
rm -rf $OUT; mkdir -p $OUT
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extras"
for v in base:0:0 plain:4:0 scatter:0:1; do
  name=${v%%:*}; rest=${v#*:}; opt=${rest%%:*}; sc=${rest#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    ASR_SANM_BLOCK8_OPT=$opt ASR_SANM_BLOCK_SCATTER=$sc rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${name}_$c -- $B > $OUT/${name}_$c.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
for name in ("base", "plain", "scatter"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = 0.0; n = 0
        for f in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (name, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "sanm_block8" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    tot += float(r["Counter_Value"]); n += 1
        out[c] = (tot / max(n, 1), n)
    print(name, "raw KiB per launch: FETCH_SIZE %.0f (%d launches)  WRITE_SIZE %.0f" % (out["FETCH_SIZE"][0], out["FETCH_SIZE"][1], out["WRITE_SIZE"][0]))
PY
find $OUT -name "*.csv" -size +1M -delete

# kernel-trace stats of the Whisper and the 256-stream streaming benches on the current build (the round's profile run predates the decode GEMM / stream_attn_kernel changes)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_whisper -- python $R/bench.py --workload whisper --steps 2 --warmup 2 --no-cpu-baseline > $OUT/stats_whisper.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_stream256 -- python $R/bench.py --workload paraformer-streaming --batch 256 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/stats_stream256.log 2>&1
for f in $(find $OUT/stats_whisper -name "*kernel_trace.csv"); do python $R/tools/trace_summary.py $f > $OUT/whisper_trace_summary.txt; done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_stats.csv" | head

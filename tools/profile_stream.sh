#!/bin/bash
# Round-4 streaming-Paraformer evidence on ONE box: bench lines of the fused launches and of the per-launch path, the phase clocks of an encoder layer and a
# decoder block, the option A/B, and the rocprofv3 kernel summary of the fused bench. Output under gpurun_out/stream/; tools/collect_profiles.py copies it.
set -x
OUT=gpurun_out/stream
mkdir -p $OUT
python bench.py --workload paraformer-streaming > $OUT/bench_fused.json 2> $OUT/bench_fused.err
ASR_STREAM_FUSED=0 python bench.py --workload paraformer-streaming --no-cpu-baseline > $OUT/bench_perlaunch.json 2> $OUT/bench_perlaunch.err
ASR_STREAM_FUSED=1 python bench.py --workload paraformer-streaming --no-cpu-baseline > $OUT/bench_encoder_only.json 2> $OUT/bench_encoder_only.err
{
  echo "# phase clocks (wall_clock64 stamps of thread 0 of every workgroup, mean over the 256 workgroups, us; 64 streams; eager launches)"
  echo "# stream_layers: wait x | LN1 + history | q|k|v GEMM | attention + FSMN + roll | exchange 0 (ctx) | out-proj | exchange 1 (x1) | LN2 | FFN-1 | exchange 2 (hid) | FFN-2 | store x + publish"
  ASR_STREAM_TIMES=20 ASR_NO_GRAPH=1 python bench.py --workload paraformer-streaming --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "stream_layers layer" | tail -3
  echo "# stream_dec: wait dec | LN | FFN-1 | slab | exchange 0 (hid f32) + LN over 2048 | FFN-2 | x1 out + k|v projection + wait | LN(x1) + FSMN | exchange 2 (x2) | LN(x2) + q | attention + roll | exchange 3 (ctx) | out-proj + publish"
  ASR_STREAM_TIMES=-5 ASR_NO_GRAPH=1 python bench.py --workload paraformer-streaming --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "stream_dec layer" | tail -3
  echo "# 8 streams (one cluster per XCD: no sharing of the L2 / fabric)"
  ASR_STREAM_TIMES=20 ASR_NO_GRAPH=1 python bench.py --workload paraformer-streaming --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "stream_layers layer" | tail -1
  echo "# tuning switches (ASR_STREAM_OPT, kernels.h: StreamLayersArgs::opt): 0 default, 1 no L2 warm-up, 2 FFN warm-up while waiting, 8 a stream's four heads on one XCD (placement of the first version)"
  bash tools/probes/stream_opt_ab.sh 2>&1 | grep -v "^  File\|^    \|Traceback\|json"
} > $OUT/stream_phase_clock.txt
# small batches: the tile kernel against the four-launch path, and its phase clock
bash tools/probes/tiles_b1.sh 2>&1 | grep -v amdgpu.ids > $OUT/sanm_tiles_small_batches.txt
{ echo "# intervals: wait x | LN1 | q|k|v GEMM | k/v meeting of the head's tiles | attention + FSMN | ctx exchange | out-proj | x1 exchange | LN2 | FFN-1 | hid exchange | FFN-2 | store x + publish (B = 1, then B = 7)"; bash tools/probes/tiles_clock.sh; } >> $OUT/sanm_tiles_small_batches.txt 2>&1
# Qwen3-ASR: paged KV cache against extents
for p in 1 0; do ASR_QWEN_KV_PAGED=$p python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_qwen_paged$p.json 2> $OUT/bench_qwen_paged$p.err; done
ASR_QWEN_KV_PAGED=1 python bench.py --workload qwen --beam 5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_qwen_beam5_paged1.json 2> $OUT/bench_qwen_beam5_paged1.err
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > $OUT/bench_mixed_beam5.json 2> $OUT/bench_mixed_beam5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --workload paraformer-streaming --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_stats.csv" | head
du -sh $OUT

#!/bin/bash
# Streaming-Paraformer fused launch: tuning switches A/B on ONE box (ASR_STREAM_OPT, kernels.h: StreamLayersArgs::opt), 64 streams.
for opt in 0 1 2 8 0; do
  echo "== ASR_STREAM_OPT=$opt"
  ASR_STREAM_OPT=$opt ASR_STREAM_TIMES=20 ASR_NO_GRAPH=1 timeout 200 python bench.py --workload paraformer-streaming --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "stream_layers layer" | tail -1
  ASR_STREAM_OPT=$opt timeout 200 python bench.py --workload paraformer-streaming --no-cpu-baseline 2>&1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'stream_layers', d['kernels'].get('stream_layers'))"
done

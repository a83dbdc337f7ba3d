#!/bin/bash
# Streaming Paraformer: the two cluster launches per chunk step (ASR_STREAM_FUSED_MAX=100000) against the per-launch path (ASR_STREAM_FUSED=0) by stream count, same box:
# where the per-launch GEMMs' weight amortisation overtakes a cluster per stream (sets the default of SvSession::st_fused_max).
for n in 64 96 128 160 192 256; do
  for mode in "ASR_STREAM_FUSED_MAX=100000" "ASR_STREAM_FUSED=0"; do
    echo "$n streams, $mode: $(env $mode python bench.py --workload paraformer-streaming --batch $n --steps 24 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per chunk step,', d['value'], 'audio-s/s')")"
  done
done

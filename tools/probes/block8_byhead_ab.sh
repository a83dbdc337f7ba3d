#!/bin/bash
# Headline (SenseVoiceSmall bf16, 64 x 8 s): the block kernel with a window's four heads on one XCD (default) against placement by head (ASR_SANM_BLOCK8_OPT=512), same box.
for o in 0 512 0 512; do
  echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, block', d['roofline']['avg_block_us'], 'us')")"
done

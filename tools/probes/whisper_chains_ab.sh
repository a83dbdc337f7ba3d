# decode chains (csrc/whisper.hip: enqueue_step, ASR_DECODE_CHAINS): parity test, then the A/B on the three Whisper bench shapes
set -x
mkdir -p gpurun_out/chains
python -m pytest tests/test_whisper_gpu.py -q -x -k "chains or batch64 or batch32" > gpurun_out/chains/pytest.txt 2>&1
tail -5 gpurun_out/chains/pytest.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s,', d.get('decode_ms_per_token'), 'ms per token')"; }
for c in 1 2 4; do echo "B=64 x 8 s, ASR_DECODE_CHAINS=$c: $(ASR_DECODE_CHAINS=$c python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --no-cpu-baseline 2>gpurun_out/chains/err_$c.txt | line)"; done > gpurun_out/chains/ab.txt 2>&1
for c in 1 2; do echo "B=32 x 8 s, ASR_DECODE_CHAINS=$c: $(ASR_DECODE_CHAINS=$c python bench.py --workload whisper --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"; done >> gpurun_out/chains/ab.txt 2>&1
for c in 1 2; do echo "B=32 x 30 s, ASR_DECODE_CHAINS=$c: $(ASR_DECODE_CHAINS=$c python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"; done >> gpurun_out/chains/ab.txt 2>&1
cat gpurun_out/chains/ab.txt
tail -3 gpurun_out/chains/err_2.txt

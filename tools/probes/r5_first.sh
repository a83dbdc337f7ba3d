set -x
mkdir -p gpurun_out/r5a
bash tools/probes/block8_byhead_ab.sh > gpurun_out/r5a/byhead_ab.txt 2>&1
python bench.py --workload paraformer-streaming --steps 16 --warmup 8 --no-cpu-baseline > gpurun_out/r5a/stream64.json 2> gpurun_out/r5a/stream64.err
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > gpurun_out/r5a/mixed.json 2> gpurun_out/r5a/mixed.err
python bench.py --workload whisper --batch 64 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r5a/whisper_b64.json 2> gpurun_out/r5a/whisper_b64.err

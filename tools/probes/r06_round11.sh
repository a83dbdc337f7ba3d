# round 6, eleventh call: FP8MM saturation counter + activation shift (tests), FP8MM bench line next to bf16 on one box
set -x
mkdir -p gpurun_out/r06k
python -m pytest tests/test_whisper_fp8_gpu.py -m gpu -q -x -k "fp8mm or fp8_session or gemm" -s > gpurun_out/r06k/pytest_fp8mm.txt 2>&1; tail -15 gpurun_out/r06k/pytest_fp8mm.txt
python bench.py --workload whisper --fp8mm --seconds 30 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r06k/bench_whisper30_fp8mm.json
python bench.py --workload whisper --seconds 30 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r06k/bench_whisper30_bf16.json
python - <<'PY'
import json
for n in ("fp8mm", "bf16"):
    try:
        d = json.load(open(f"gpurun_out/r06k/bench_whisper30_{n}.json")); print(n, d["value"], d["ms_per_step"], (d.get("kernels") or {}).get("gemm_ffn1"), (d.get("kernels") or {}).get("gemm_ffn2"))
    except Exception as e: print(n, "failed", e)
PY

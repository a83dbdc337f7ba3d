# round 6, twelfth call: Qwen3-ASR q / k norm + RoPE kernel with 16-byte accesses (sixteen lanes per head): parity tests + bench line
set -x
mkdir -p gpurun_out/r06l
python -m pytest tests/test_qwen_asr_gpu.py tests/test_qwen_fp8_gpu.py tests/test_shim_qwen_gpu.py -m gpu -q -x > gpurun_out/r06l/pytest_qwen.txt 2>&1; tail -5 gpurun_out/r06l/pytest_qwen.txt
python bench.py --workload qwen --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06l/bench_qwen.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06l/bench_qwen.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"].get("dec_rope"))
PY

# stream_attn_kernel after a change: parity tests of the streaming path, the co-tenancy determinism probe, the 64- / 256-stream bench lines
mkdir -p gpurun_out/sa
python -m pytest tests/test_paraformer_streaming_gpu.py tests/test_mixed_gpu.py tests/test_natural_audio_gpu.py tests/test_shim_paraformer_streaming_gpu.py -q -x > gpurun_out/sa/pytest.txt 2>&1
tail -3 gpurun_out/sa/pytest.txt
for env in "PROBE_PREC=0 ASR_STREAM_FUSED=0" "PROBE_PREC=1"; do
  echo "== $env"
  env $env ASR_SANM_BLOCK_MIN=99 ASR_STREAM_SHARE=0 python tools/probes/stream_determinism.py sensevoice 2>&1 | grep -v amdgpu.ids | grep "passes differ\|tap s0_ctx" | cut -c1-150 | head -6
done
for b in 64 256; do echo "streams $b: $(python bench.py --workload paraformer-streaming --batch $b --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"; done

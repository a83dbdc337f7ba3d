# the bf16 mixed test (streaming cluster launches beside a Qwen3-ASR session, share rule off) twelve times in fresh processes: the repetition that exposed stream_attn_kernel's co-tenancy fault (profiles/r05_stream_determinism.txt)
mkdir -p gpurun_out/mixed_x12
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    python -m pytest "tests/test_mixed_gpu.py::test_bf16_cluster_launches_beside_a_qwen_session[0]" -q -x -s > gpurun_out/mixed_x12/run_$i.txt 2>&1
    echo "run $i: $(tail -1 gpurun_out/mixed_x12/run_$i.txt)"; grep "AssertionError: (" gpurun_out/mixed_x12/run_$i.txt | cut -c1-900
done

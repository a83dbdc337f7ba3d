#!/usr/bin/env python
"""Fold what tools/profile_round.sh left under gpurun_out/prof into the tracked profiles/<round>_* files (run locally after gpurun)."""
import glob, os, shutil, subprocess, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
names = {"bench_sensevoice": "bench_n1", "bench_sensevoice_4launch": "bench_4launch_n1", "bench_paraformer": "bench_paraformer_n1", "bench_whisper": "bench_whisper_n1",
         "bench_whisper_b64": "bench_whisper_b64_n1", "bench_whisper30": "bench_whisper30_n1", "bench_paraformer_streaming": "bench_paraformer_streaming_n1",
         "bench_whisper30_fp8": "bench_whisper30_fp8_n1", "bench_whisper_fp8": "bench_whisper_fp8_n1",
         "bench_whisper30_mxfp4": "bench_whisper30_mxfp4_n1", "bench_whisper_mxfp4": "bench_whisper_mxfp4_n1",
         "bench_whisper_b64_fp8mm": "bench_whisper_b64_fp8mm_n1", "bench_whisper30_fp8mm": "bench_whisper30_fp8mm_n1",
         "bench_qwen": "bench_qwen_n1", "bench_qwen_beam5": "bench_qwen_beam5_n1", "bench_mixed_beam5": "bench_mixed_beam5_n1",
         "bench_paraformer_streaming_256": "bench_paraformer_streaming_256_n1", "bench_qwen_fp8": "bench_qwen_fp8_n1", "bench_qwen_mxfp4": "bench_qwen_mxfp4_n1"}
for a, b in names.items():
    p = os.path.join(src, a + ".json")
    if os.path.isfile(p) and os.path.getsize(p) > 10:
        shutil.copy(p, f"profiles/{rnd}_{b}.json")
for d, out in (("stats", "sensevoice_b64"), ("stats_whisper", "whisper_b32"), ("stats_whisper30", "whisper30_b32"), ("stats_qwen", "qwen_b64"), ("stats_paraformer", "paraformer_b64")):
    fs = glob.glob(os.path.join(src, d, "*", "*kernel_stats.csv"))
    if fs:
        shutil.copy(sorted(fs, key=os.path.getmtime)[-1], f"profiles/{rnd}_{out}_kernel_stats.csv")
for t in ("sensevoice_bf16_b1_trace_summary.txt", "sanm_block_phase_clock.txt", "sensevoice_f32_b1_profile.txt", "fp8_gemm_probe.txt", "gelu_epilogue_cost.txt", "sanm_block_ablations.txt", "sanm_block_batch_sweep.txt", "sanm_block_variants.txt", "sanm_block_min_sweep.txt", "sanm_block_clock_inside_A.txt", "fbank_ablations_final.txt"):
    p = os.path.join(src, t)
    if os.path.isfile(p):
        shutil.copy(p, f"profiles/{rnd}_{t}")
for w in ("whisper", "qwen"):
    p = os.path.join(src, f"{w}_trace_summary.txt")
    if os.path.isfile(p):
        shutil.copy(p, f"profiles/{rnd}_{w}_trace_summary.txt")
for j in ("hbm_traffic.json", "mfma_util.json", "whisper_mfma_util.json"):          # folded on the GPU box by profile_round.sh
    if os.path.isfile(os.path.join(src, j)) and os.path.getsize(os.path.join(src, j)) > 50:
        shutil.copy(os.path.join(src, j), f"profiles/{rnd}_{j}")
if not os.path.isfile(os.path.join(src, "hbm_traffic.json")) and glob.glob(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv")):
    subprocess.run([sys.executable, "tools/summarize_pmc.py", src, f"profiles/{rnd}_hbm_traffic.json"], check=True)
if not os.path.isfile(os.path.join(src, "mfma_util.json")) and glob.glob(os.path.join(src, "pmc_sq", "*", "*_counter_collection.csv")):
    subprocess.run([sys.executable, "tools/summarize_sq_pmc.py", os.path.join(src, "pmc_sq"), f"profiles/{rnd}_mfma_util.json"], check=True)
if not os.path.isfile(os.path.join(src, "whisper_mfma_util.json")) and glob.glob(os.path.join(src, "pmc_sq_whisper", "*", "*_counter_collection.csv")):
    subprocess.run([sys.executable, "tools/summarize_sq_pmc.py", os.path.join(src, "pmc_sq_whisper"), f"profiles/{rnd}_whisper_mfma_util.json"], check=True)
p = os.path.join(src, "pp_gemm_probe.txt")
if os.path.isfile(p):
    shutil.copy(p, f"profiles/{rnd}_pp_gemm_probe.txt")
print(sorted(f for f in os.listdir("profiles") if f.startswith(rnd)))

# round-4 streaming evidence (tools/profile_stream.sh -> gpurun_out/stream)
ssrc = "gpurun_out/stream"
if os.path.isdir(ssrc):
    for name, out in (("bench_fused.json", "bench_paraformer_streaming_n1.json"), ("bench_perlaunch.json", "bench_paraformer_streaming_perlaunch_n1.json"),
                      ("bench_encoder_only.json", "bench_paraformer_streaming_encoder_only_n1.json"), ("stream_phase_clock.txt", "stream_phase_clock.txt"),
                      ("sanm_tiles_small_batches.txt", "sanm_tiles_small_batches.txt"), ("bench_qwen_paged1.json", "bench_qwen_n1.json"),
                      ("bench_qwen_paged0.json", "bench_qwen_extents_n1.json"), ("bench_qwen_beam5_paged1.json", "bench_qwen_beam5_n1.json"),
                      ("bench_sensevoice.json", "bench_n1.json"), ("bench_mixed_beam5.json", "bench_mixed_beam5_n1.json")):
        p = os.path.join(ssrc, name)
        if os.path.isfile(p) and os.path.getsize(p) > 0:
            shutil.copy(p, f"profiles/{rnd}_{out}")
    fs = glob.glob(os.path.join(ssrc, "stats", "*", "*kernel_stats.csv"))
    if fs:
        shutil.copy(sorted(fs, key=os.path.getmtime)[-1], f"profiles/{rnd}_paraformer_streaming_kernel_stats.csv")

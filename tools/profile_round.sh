#!/bin/bash
# Round profile (run on the GPU box through gpurun): bench lines of every workload, rocprofv3 kernel-trace stats of the headline and of
# the other families, three PMC passes of the headline (FETCH_SIZE, WRITE_SIZE, SQ counters -- each in its own run, kernel-trace only).
# Everything lands under gpurun_out/prof; tools/collect_profiles.py then folds it into profiles/<round>_*.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
cd $R
# the headline's kernel table FIRST, copied over the committed one on this box: the bench line below cites profiles/<round>_sensevoice_b64_kernel_stats.csv and
# compares its own HIP-event launch time with that table's average -- both are then this box's
ROUND=${ROUND:-r06}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 1 > $OUT/stats.log 2>&1
for f in $(find $OUT/stats -name "*kernel_stats.csv" | head -1); do cp $f profiles/${ROUND}_sensevoice_b64_kernel_stats.csv; done
python bench.py --steps 20 --warmup 5 > $OUT/bench_sensevoice.json 2> $OUT/bench_sensevoice.err
ASR_SANM_BLOCK=0 python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_sensevoice_4launch.json 2> $OUT/bench_sensevoice_4launch.err
python bench.py --workload paraformer --steps 10 > $OUT/bench_paraformer.json 2> $OUT/bench_paraformer.err
python bench.py --workload whisper --steps 6 --warmup 3 --inflight 3 > $OUT/bench_whisper.json 2> $OUT/bench_whisper.err
python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --inflight 2 --no-cpu-baseline > $OUT/bench_whisper_b64.json 2> $OUT/bench_whisper_b64.err
python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --inflight 3 --no-cpu-baseline > $OUT/bench_whisper30.json 2> $OUT/bench_whisper30.err
python bench.py --workload whisper --fp8 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_whisper30_fp8.json 2> $OUT/bench_whisper30_fp8.err
python bench.py --workload whisper --fp8 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/bench_whisper_fp8.json 2> $OUT/bench_whisper_fp8.err
python bench.py --workload whisper --mxfp4 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_whisper30_mxfp4.json 2> $OUT/bench_whisper30_mxfp4.err
python bench.py --workload whisper --mxfp4 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/bench_whisper_mxfp4.json 2> $OUT/bench_whisper_mxfp4.err
python bench.py --workload whisper --fp8mm --batch 64 --steps 5 --warmup 3 --inflight 2 --no-cpu-baseline > $OUT/bench_whisper_b64_fp8mm.json 2> $OUT/bench_whisper_b64_fp8mm.err
python bench.py --workload whisper --fp8mm --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_whisper30_fp8mm.json 2> $OUT/bench_whisper30_fp8mm.err
python tools/fp8_gemm_probe.py > $OUT/fp8_gemm_probe.txt 2>&1
python tools/probes/gelu_cost.py > $OUT/gelu_epilogue_cost.txt 2>&1
# the 8-wave block kernel: timing-only ablations of its GEMM loops (ASR_SANM_BLOCK8_OPT bits 4..7; they exist for the round-4 form of the FFN pair and switch the K-split
# form off), the phase clock of one block of the long launch in both forms, and the same box's step times for the switches that are left
for a in 0 16 32 64 128 192 224; do echo "=== ASR_SANM_BLOCK8_OPT=$a (x16: 1 no MFMA, 2 no A fragment reads, 4 no W refills, 8 no chunk DMA; results are garbage by design)"; ASR_SANM_BLOCK8_OPT=$a ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > $OUT/sanm_block_ablations.txt 2>&1
for k in 1 0; do echo "=== ASR_SANM_BLOCK_FFNK=$k (1 = FFN-2 split over K, f16 partials exchanged: stamps 11->12 = the FFN-2 loop, 12->13 = partial out + publish, 13->14 = wait + foreign partials + epilogue)"; ASR_SANM_BLOCK_FFNK=$k ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -51; done > $OUT/sanm_block_phase_clock.txt 2>&1
for b in 1 16; do echo "=== batch $b (ASR_SANM_BLOCK_MIN=1): the per-cluster critical path with the chip idle"; CLOCK_B=$b ASR_SANM_BLOCK_MIN=1 ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > $OUT/sanm_block_batch_sweep.txt 2>&1
for v in "ASR_SANM_BLOCK_FFNK=1" "ASR_SANM_BLOCK_FFNK=0" "ASR_GEMM_AMAX_PP=0" "ASR_SANM_BLOCK_PERSIST=0" "ASR_SANM_BLOCK8_OPT=4" "ASR_SANM_BLOCK8_OPT=256" "ASR_SANM_BLOCK_FFNK=0" "ASR_SANM_BLOCK_FFNK=1"; do
  echo "$v: $(env $v python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s')")"
done > $OUT/sanm_block_variants.txt 2>&1
bash tools/probes/block_min_sweep.sh > $OUT/sanm_block_min_sweep.txt 2>&1
python tools/probes/f32_b1_profile.py > $OUT/sensevoice_f32_b1_profile.txt 2>&1
python bench.py --workload paraformer-streaming --steps 16 --warmup 8 > $OUT/bench_paraformer_streaming.json 2> $OUT/bench_paraformer_streaming.err
python bench.py --workload paraformer-streaming --batch 256 --steps 16 --warmup 8 --no-cpu-baseline > $OUT/bench_paraformer_streaming_256.json 2> $OUT/bench_paraformer_streaming_256.err
ASR_SANM_BLOCK8_OPT=2048 ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -23 > $OUT/sanm_block_clock_inside_A.txt
python bench.py --workload qwen --steps 6 --warmup 2 --inflight 3 > $OUT/bench_qwen.json 2> $OUT/bench_qwen.err
python bench.py --workload qwen --beam 5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_qwen_beam5.json 2> $OUT/bench_qwen_beam5.err
python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_qwen_fp8.json 2> $OUT/bench_qwen_fp8.err
python bench.py --workload qwen --mxfp4 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_qwen_mxfp4.json 2> $OUT/bench_qwen_mxfp4.err
for v in 0 1 2 3 4 8; do echo "ASR_FBANK_DBG=$v: $(ASR_FBANK_DBG=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step; fbank', d['kernels']['fbank']['ms_per_step'])")"; done > $OUT/fbank_ablations_final.txt 2>&1
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > $OUT/bench_mixed_beam5.json 2> $OUT/bench_mixed_beam5.err
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_whisper -- python $R/bench.py --workload whisper --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_whisper.log 2>&1
python $R/tools/pp_gemm_probe.py > $OUT/pp_gemm_probe.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_whisper -- python $R/bench.py --workload whisper --steps 2 --warmup 2 --no-cpu-baseline > $OUT/stats_whisper.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_whisper30 -- python $R/bench.py --workload whisper --seconds 30 --steps 2 --warmup 2 --no-cpu-baseline > $OUT/stats_whisper30.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_qwen -- python $R/bench.py --workload qwen --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_qwen.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_paraformer -- python $R/bench.py --workload paraformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_paraformer.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b1trace -- python $R/tools/probes/bf16_b1_trace.py > $OUT/b1trace.log 2>&1
for f in $(find $OUT/b1trace -name "*kernel_trace.csv"); do python $R/tools/trace_summary.py $f > $OUT/sensevoice_bf16_b1_trace_summary.txt; done
for w in whisper qwen; do for f in $(find $OUT/stats_$w -name "*kernel_trace.csv"); do python $R/tools/trace_summary.py $f > $OUT/${w}_trace_summary.txt; done; done
# fold the counter passes here (the raw per-dispatch CSVs are tens of MB; gpurun copies back at most 64 MiB)
python $R/tools/summarize_pmc.py $OUT $OUT/hbm_traffic.json > $OUT/summarize_pmc.log 2>&1
python $R/tools/summarize_sq_pmc.py $OUT/pmc_sq $OUT/mfma_util.json > $OUT/summarize_sq.log 2>&1
python $R/tools/summarize_sq_pmc.py $OUT/pmc_sq_whisper $OUT/whisper_mfma_util.json > $OUT/summarize_sq_whisper.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.csv" | head -40
du -sh $OUT

"""CPU: the evidence bench.py folds into its JSON line parses -- the committed rocprofv3 summaries under profiles/ (HBM traffic and SQ
counter passes, kernel-trace stats) -- and the committed bench lines keep the contract's keys."""
import glob
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_counter_summaries_feed_the_roofline():
    t = bench.hbm_traffic("gemm_bf16_t144")
    assert t and t["bytes_per_launch"] > 1e6 and t["source"].startswith("profiles/")
    m = bench.mfma_util("gemm_bf16_t144")
    assert m and 0.0 < m["mfma_busy_frac"] < 1.0
    shares = m["wave_cycles"]
    assert 0.9 < shares["wait_any_share"] + shares["wait_inst_any_share"] + shares["active_inst_share"] < 1.1   # disjoint buckets of the wave cycles
    r = bench.rocprof_avg_us(("gemm_bf16_t144<", "gemm_bf16_t288w<1,"))
    assert r and 5.0 < r["avg_launch_us"] < 200.0 and r["launches"] > 100


def test_committed_bench_lines_keep_the_contract():
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_bench_*n1.json")))
    assert len(files) >= 18 and os.path.join(ROOT, "profiles", "r02_bench_n1.json") in files
    for f in files:
        d = json.load(open(f))
        assert need <= set(d), (f, need - set(d))
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"], f
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]), f
    for rnd in ("r01", "r02"):
        head = json.load(open(os.path.join(ROOT, "profiles", rnd + "_bench_n1.json")))
        assert head["cpu_baseline"]["kind"] == "port" and head["cpu_baseline"]["cores"] >= 1 and head["cpu_baseline"]["sample"]
        assert head["roofline"]["bound"] == "mfma" and abs(head["roofline"]["frac"] - head["roofline"]["achieved"] / head["roofline"]["peak"]) < 1e-3
    for f in files:      # no line prices a kernel above the machine
        frac = json.load(open(f))["roofline"]["frac"]
        assert frac is None or 0.0 < frac < 1.0, f


def test_block_kernel_numerator_is_a_known_answer():
    """roofline.achieved of the headline line = algorithmic work of the 69 blocks the block-kernel launches walk (block 0 runs as three other
    launches outside the "sanm_block" scope) / their measured time: 901.9 MFLOP x 64 windows x 69 blocks = 3 982.8 GFLOP per 64 x 8 s step."""
    import importlib
    cfgmod = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.config")
    cfg = cfgmod.SenseVoiceConfig()
    lengths = [128000] * 64
    walked = bench.sensevoice_algorithmic_flops(cfg, lengths, blocks=(cfg.n_enc0, cfg.n_blocks))
    assert cfg.n_blocks - cfg.n_enc0 == 69
    assert abs(sum(walked.values()) / 1e9 - 3982.8) < 0.5
    assert walked["gemm_ctc"] == 0.0 and walked["fbank"] == 0.0
    whole = bench.sensevoice_algorithmic_flops(cfg, lengths)
    assert abs(sum(whole.values()) / 1e12 - 4.29) < 0.01                  # 67.03 GFLOP per window
    per_class = ("gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2", "attention", "fsmn")
    block0 = bench.sensevoice_algorithmic_flops(cfg, lengths, blocks=(0, cfg.n_enc0))
    for k in per_class:
        assert abs(walked[k] + block0[k] - whole[k]) < 1e-3 * whole[k]

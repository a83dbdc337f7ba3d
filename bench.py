#!/usr/bin/env python
"""Headline benchmark: SenseVoiceSmall, bf16 MFMA mode, batch = 64 x 8 s @ 16 kHz chunks per GPU
(BASELINE.json configs[1]), data-parallel over utterances for --gpus N (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the whole hot path (fbank -> LFR/CMVN -> 70 SANM blocks -> CTC arg-max +
collapse -> token ids on the host) over one batch of synthetic audio. Batch k's audio is resident in HBM
when step k starts; the pinned host -> device copy of batch k + 1 (copy stream) and every token-id download are
INSIDE the timed region (`audio_resident` is the same loop with the uploads left out).
Rank 0 prints ONE JSON line. Random-init weights of the exact architecture, synthetic int16-range audio
(no checkpoints / datasets exist offline).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (what the host driver supports): RCCL across processes needs it before the HIP runtime starts

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "automatic-speech-recognition-asr-onnx_amd"

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md:42
HBM_PEAK_GBS = 8000.0               # HBM3E spec, MI355X_MICROARCH.md:35


def sensevoice_algorithmic_flops(cfg, lengths, blocks=None):
    """Algorithmic FLOPs per kernel class for one batch (GEMM = 2MNK on un-padded rows, attention = 4 T^2 d
    per layer, no recompute, no padding) -- SURVEY.md section 8(d). `blocks` = (first, end): only the SANM blocks of that range
    (the block kernel's launches walk blocks n_enc0 .. n_blocks - 1; block 0, whose input is 560 wide, runs as three other launches),
    head and front-end left out."""
    d, dff, feat = cfg.d_model, cfg.d_ffn, cfg.feat_dim
    nfreq = cfg.nfft // 2 + 1
    out = dict.fromkeys(("gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2", "gemm_ctc", "attention", "fsmn", "fbank"), 0.0)
    b0, b1 = blocks if blocks else (0, cfg.n_blocks)
    n0 = max(0, min(b1, cfg.n_enc0) - b0)                 # blocks of the range whose input is feat_dim wide
    for n in lengths:
        T, frames = cfg.seq_len(n), cfg.n_frames(n)
        nb = b1 - b0
        out["gemm_qkv"] += 2.0 * T * 3 * d * (feat * n0 + d * (nb - n0))
        out["gemm_out"] += 2.0 * T * d * d * nb
        out["gemm_ffn1"] += 2.0 * T * d * dff * nb
        out["gemm_ffn2"] += 2.0 * T * d * dff * nb
        out["attention"] += 4.0 * T * T * d * nb
        out["fsmn"] += 2.0 * T * d * cfg.fsmn_kernel * nb
        if not blocks:
            out["gemm_ctc"] += 2.0 * T * d * cfg.vocab
            out["fbank"] += frames * (2.0 * cfg.win_length * 2 * nfreq + 2.0 * nfreq * cfg.n_mels)
    return out


PROFILE_ROUND = None        # "rNN": which round's committed profiles the roofline block cites (--round; default = newest on file)


def profile_path(stem):
    """profiles/<round>_<stem> for the round given by --round, else the newest round that has this file; None when none exists.
    The cited file name is stamped into the output line, so a stale citation is visible."""
    import glob
    import re
    if PROFILE_ROUND:
        p = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{stem}")
        return p if os.path.isfile(p) else None
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}")):
        m = re.match(r"r(\d\d)_", os.path.basename(p))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), p)
    return best[1] if best else None


def hbm_traffic(kernel):
    """Memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_hbm_traffic.json: FETCH_SIZE
    doubled per the gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE); None when no counter run is on file."""
    p = profile_path("hbm_traffic.json")
    try:
        with open(p) as f:
            ks = json.load(f)["kernels"]
        rec = ks.get(kernel)
        if rec is None:
            cand = [k for k in ks if k.startswith(kernel + "<") or k.startswith(kernel + "(")]
            rec = ks[cand[0]] if cand else None
        return None if rec is None else {"bytes_per_launch": rec["bytes_per_launch"], "unit": "B", "source": os.path.relpath(p, ROOT)}
    except (OSError, KeyError, ValueError, TypeError):
        return None


def mfma_util(kernel, stem="mfma_util.json"):
    """MFMA pipe utilisation of `kernel` from the committed SQ counter pass (profiles/rNN_mfma_util.json, tools/summarize_sq_pmc.py)."""
    p = profile_path(stem)
    try:
        with open(p) as f:
            ks = json.load(f)["kernels"]
        rec = ks.get(kernel)
        if rec is None:                                   # template instances are keyed with their arguments ("sanm_block_kernel<false>")
            cand = [k for k in ks if k.startswith(kernel + "<") or k.startswith(kernel + "(")]
            rec = ks[max(cand, key=lambda k: ks[k].get("launches", 0))] if cand else None
        if rec is None:
            return None
        out = {"mfma_busy_frac": rec["mfma_util"], "wave_cycles": {k: rec[k] for k in ("wait_any_share", "wait_inst_any_share", "active_inst_share")},
               "source": os.path.relpath(p, ROOT)}
        if "effective_clock_ghz" in rec:                  # GRBM_GUI_ACTIVE cycles / kernel-trace duration: the clock the chip held under this kernel
            out["effective_clock_ghz"] = rec["effective_clock_ghz"]
        return out
    except (OSError, KeyError, ValueError, TypeError):
        return None


def rocprof_avg_us(match, exclude=()):
    """Average kernel duration of the launches whose name contains one of `match`, from the committed rocprofv3 --kernel-trace --stats
    summary of this command (profiles/rNN_sensevoice_b64_kernel_stats.csv). The live figure next to it is bracketed by HIP events on the
    session stream and so includes the gap between consecutive launches (a few us each)."""
    import csv
    p = profile_path("sensevoice_b64_kernel_stats.csv")
    try:
        tot = calls = 0.0
        with open(p) as f:
            for r in csv.DictReader(f):
                if any(m in r["Name"] for m in match) and not any(x in r["Name"] for x in exclude):
                    tot += float(r["TotalDurationNs"]); calls += float(r["Calls"])
        return None if calls == 0 else {"avg_launch_us": round(tot / calls / 1e3, 2), "launches": int(calls), "source": os.path.relpath(p, ROOT)}
    except (OSError, KeyError, ValueError, TypeError):
        return None


def cpu_baseline(cfg, ck, audio_np, budget_s=15.0):
    """The oracle (torch CPU f32 restatement of the reference graph, batch 1 like the reference) timed on this
    host's cores on a bounded sample of the same workload. CHECKER ONLY -- never on the product path."""
    import torch
    from oracle.sensevoice_oracle import SenseVoiceOracle
    orc = SenseVoiceOracle(cfg, ck)
    orc(audio_np[0, 0], 0)                                  # warm-up
    # small GEMMs (T = 137 rows) scale poorly to every core of a big host: probe a few thread counts, keep the best
    best_t, best_n = None, None
    for n_thr in (8, 16, 32):
        if n_thr > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(n_thr)
        orc(audio_np[0, 0], 0)
        t0 = time.perf_counter()
        orc(audio_np[0, 0], 0)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, n_thr
    torch.set_num_threads(best_n)
    n_done, t0 = 0, time.perf_counter()
    while True:
        orc(audio_np[n_done % audio_np.shape[0], 0], 0)
        n_done += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n_done >= 4) or n_done >= 64:
            break
    secs = audio_np.shape[2] / cfg.sample_rate
    out = {"value": round(n_done * secs / el, 2), "unit": "audio-s/s", "cores": int(torch.get_num_threads()), "host_cores": int(os.cpu_count() or 0), "kind": "port",
           "sample": f"{n_done} x {secs:.0f} s utterances, batch 1, torch-CPU f32 oracle (oracle/sensevoice_oracle.py), "
                     f"{el:.1f} s wall; RTF {el / (n_done * secs):.4f}; `cores` = the fastest of 8 / 16 / 32 threads"}
    # the two settings SURVEY section 8(d) names, each on its own bounded sample: one thread here; every core of the host in a child process with a hard
    # time limit (at 256 threads torch's intra-op pool collapses on these 137-row GEMMs: one 8 s utterance took 225 s on the round-4 box)
    torch.set_num_threads(1)
    orc(audio_np[0, 0], 0)
    k, t0 = 0, time.perf_counter()
    while True:
        orc(audio_np[k % audio_np.shape[0], 0], 0)
        k += 1
        e2 = time.perf_counter() - t0
        if e2 >= 8.0 or k >= 32:
            break
    out["single_thread"] = {"value": round(k * secs / e2, 2), "cores": 1, "sample": f"{k} x {secs:.0f} s utterances, {e2:.1f} s wall"}
    out["all_host_cores"] = cpu_leg_all_cores(int(os.cpu_count() or 1), secs, limit_s=45.0)
    # ... and every core the way a CPU deployment of the reference would use them for a batch of independent utterances: nproc / best_n worker PROCESSES of
    # best_n threads each, one utterance at a time per worker (the reference is batch 1 per InferenceSession; utterances are independent -- SURVEY 8e)
    out["all_host_cores_multiproc"] = cpu_leg_multiproc(int(os.cpu_count() or 1), best_n, secs, limit_s=75.0)
    torch.set_num_threads(best_n)
    return out


def cpu_leg_multiproc(n_cores, n_thr, secs, limit_s):
    """nproc-wide CPU figure that does not depend on one intra-op pool spanning the host: n_cores / n_thr children of `bench.py --cpu-leg n_thr`, started
    together, each running utterances for 5 s; value = the sum of their rates (they overlap for all but process start-up). CHECKER ONLY."""
    import subprocess
    n_proc = max(1, n_cores // max(n_thr, 1))
    env = dict(os.environ, OMP_NUM_THREADS=str(n_thr), MKL_NUM_THREADS=str(n_thr))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg", str(n_thr)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(n_proc)]
    t0, total, done = time.perf_counter(), 0.0, 0
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=max(1.0, limit_s - (time.perf_counter() - t0)))
            total += float(json.loads(o.strip().splitlines()[-1])["value"])
            done += 1
        except (subprocess.TimeoutExpired, ValueError, IndexError, KeyError):
            pr.kill()
    if done == 0:
        return {"value": None, "cores": n_proc * n_thr, "sample": f"none of {n_proc} worker processes finished inside {limit_s:.0f} s"}
    return {"value": round(total, 2), "cores": done * n_thr,
            "sample": f"{done} of {n_proc} worker processes x {n_thr} threads, each running {secs:.0f} s utterances one at a time for 5 s (first pass included); sum of their rates"}


def cpu_leg_all_cores(n_thr, secs, limit_s):
    """`bench.py --cpu-leg N` in a child process: the SenseVoice oracle at N threads, utterances until 5 s have passed, at most `limit_s` of wall time
    (checkpoint synthesis included) -- a leg that does not finish one utterance reports the bound it proved instead of holding the bench line up."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", str(n_thr)], capture_output=True, text=True, timeout=limit_s)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": rec["value"], "cores": n_thr, "sample": rec["sample"]}
    except subprocess.TimeoutExpired:
        return {"value": None, "cores": n_thr, "sample": f"no {secs:.0f} s utterance finished inside the {limit_s:.0f} s limit of this leg (process start and checkpoint synthesis included)"}
    except (ValueError, IndexError, KeyError, OSError) as e:
        return {"value": None, "cores": n_thr, "sample": f"leg failed: {type(e).__name__}"}


def cpu_leg_main(n_thr):
    """Child of cpu_leg_all_cores (CHECKER ONLY, no GPU): same seeded checkpoint and audio as the bench, torch at n_thr threads."""
    import torch
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    from oracle.sensevoice_oracle import SenseVoiceOracle
    cfg = cfgm.sensevoice_small()
    ck = ckm.synth_sensevoice_checkpoint(cfg, seed=0)
    audio = ckm.synth_audio("kaldi", 4, 8 * cfg.sample_rate, seed=1234)
    torch.set_num_threads(n_thr)
    orc = SenseVoiceOracle(cfg, ck)
    k, t0 = 0, time.perf_counter()
    while True:                                              # (no warm-up pass: at this thread count one pass may be all the limit allows)
        orc(audio[k % 4, 0], 0)
        k += 1
        el = time.perf_counter() - t0
        if el >= 5.0 or k >= 32:
            break
    print(json.dumps({"value": round(k * 8.0 / el, 2), "sample": f"{k} x 8 s utterances, {el:.1f} s wall, first pass included"}))


def self_launch_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when no launcher set WORLD_SIZE: one rank per GPU under
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def dry_run(args):
    """ASR_BENCH_DRYRUN=1: the launch / rendezvous / data-parallel plumbing of a workload's step loop without a GPU (gloo on CPU) -- what
    tests/test_dist_cpu.py drives to prove that `python bench.py --workload W --gpus N` starts N ranks, broadcasts an arena from rank 0, shards the
    utterances, gathers every step's hypothesis slabs on rank 0, takes the slowest rank's clock and prints ONE line on rank 0 (BASELINE.json configs[3] /
    [4] are this path at N = 8 with the RCCL backend; no 8-GPU node was available to any round)."""
    import torch
    import torch.distributed as dist
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    cpu = torch.device("cpu")
    B = args.batch or (32 if args.workload == "whisper" else 64)
    width = {"whisper": int(args.decode_tokens or 4 * args.seconds), "qwen": 64, "mixed": 64}.get(args.workload, 137)
    blob = (np.arange(1 << 16, dtype=np.uint32).view(np.uint8) if rank == 0 else None)          # stands in for the weight arena rank 0 builds
    arena = dp.broadcast_arena(blob, cpu)
    assert int(arena.to(torch.int64).sum()) == int(np.arange(1 << 16, dtype=np.uint32).view(np.uint8).astype(np.int64).sum())
    dp.barrier(cpu)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    gathered = 0
    for k in range(args.warmup + args.steps):
        num = np.full(B, 1 + (rank + k) % width, dtype=np.int32)
        tok = np.tile(np.arange(width, dtype=np.int32)[None] + 1000 * rank, (B, 1))
        slabs = dp.gather_hypotheses(dp.pack_hypotheses(tok, num, width), cpu)
        if rank == 0:
            assert len(slabs) == world
            for r_, sl in enumerate(slabs):
                hyp = dp.unpack_hypotheses(sl)
                assert len(hyp) == B and all(h.size == 1 + (r_ + k) % width and (h.size == 0 or h[0] == 1000 * r_) for h in hyp)
            if k >= args.warmup:
                gathered += world * B
    time.sleep(0.01 * (rank + 1))
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, cpu)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "workload": args.workload, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "global_batch": world * B, "gathered_hypotheses": gathered, "arena_broadcast_bytes": int(arena.numel()),
                          "max_rank_seconds": round(elapsed, 4)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (sensevoice default 200 = 1.7 s of timed region; other workloads 20)")
    ap.add_argument("--round", default=None, help="rNN: cite this round's committed profiles in the roofline block (default: the newest on file)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default 64; whisper: 32 = BASELINE.json configs[2])")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="sensevoice: skip the PCIe-inclusive and batches-in-flight legs (rocprofv3 --stats runs: their overlapped launches would skew per-kernel averages)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--workload", choices=("sensevoice", "whisper", "paraformer", "paraformer-streaming", "qwen", "mixed"), default="sensevoice",
                    help="sensevoice = BASELINE.json configs[1] (default, the headline line); whisper = large-v3 encoder + greedy decode")
    ap.add_argument("--inflight", type=int, default=1, help="whisper / qwen: also measure N batches in flight on N sessions / HIP streams")
    ap.add_argument("--streams", type=int, default=256, help="mixed: concurrent Paraformer streams per GPU")
    ap.add_argument("--beam", type=int, default=1, help="qwen / mixed: beam width (1 = greedy; BASELINE.json configs[4] names beam 5)")
    ap.add_argument("--fp8mm", action="store_true", help="whisper: opt-in precision mode ASR_PRECISION_FP8MM (FP8W + the encoder's FFN pair on the FP8 matrix pipe); a secondary figure, never the headline")
    ap.add_argument("--mxfp4", action="store_true", help="whisper / qwen: opt-in precision mode ASR_PRECISION_MXFP4W (decoder projections as OCP MXFP4; Whisper: cross-K/V as e4m3); a secondary figure, never the headline")
    ap.add_argument("--fp8", action="store_true", help="whisper / qwen: opt-in precision mode ASR_PRECISION_FP8W (decoder projections -- Whisper: and cross-K/V -- as e4m3 bytes); a secondary figure, never the headline")
    ap.add_argument("--decode-tokens", type=int, default=0, help="whisper: generated tokens per utterance (default 4 per audio second)")
    ap.add_argument("--cpu-leg", type=int, default=0, help=argparse.SUPPRESS)        # child process of the CPU baseline's every-core leg
    args = ap.parse_args()
    if args.cpu_leg:
        return cpu_leg_main(args.cpu_leg)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # started by hand: become the launcher the contract describes
        os.execv(sys.executable, self_launch_command(args.gpus, sys.argv[1:]))
    if os.environ.get("ASR_BENCH_DRYRUN") == "1":
        return dry_run(args)
    global PROFILE_ROUND
    PROFILE_ROUND = args.round
    if args.workload != "sensevoice" and "--steps" not in " ".join(sys.argv):
        args.steps = 20
    if args.batch is None:
        args.batch = 32 if args.workload == "whisper" else 64
    if args.workload == "whisper":
        return main_whisper(args)
    if args.workload == "paraformer":
        return main_paraformer(args)
    if args.workload == "paraformer-streaming":
        return main_paraformer_streaming(args)
    if args.workload == "qwen":
        return main_qwen(args)
    if args.workload == "mixed":
        return main_mixed(args)

    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")

    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    cfg = cfgm.sensevoice_small()
    n_samples = int(args.seconds * cfg.sample_rate)
    B = args.batch

    # ---- weights: rank 0 converts the checkpoint, RCCL broadcast of the bf16 arena, borrowed in place
    ck = None
    blob = None
    if rank == 0:
        ck = ckm.synth_sensevoice_checkpoint(cfg, seed=0)
        blob = arena.build_sensevoice_arena(cfg, ck, arena.PRECISION_BF16)
    t0 = time.perf_counter()
    arena_dev = dp.broadcast_arena(blob, device)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    sess = eng.SenseVoiceSession(cfg, arena_dev, arena.PRECISION_BF16, local_rank, arena_device_ptr=arena_dev.data_ptr(),
                                 arena_bytes=arena_dev.numel())

    # ---- synthetic audio: step k's batch is resident in HBM when step k starts; the NEXT batch's host -> device copy runs on a copy
    #      stream under it (two device buffers), so the timed region contains every upload and every token download
    audio_np = ckm.synth_audio("kaldi", B, n_samples, seed=1234 + rank)
    audio_pin = torch.from_numpy(audio_np).pin_memory()
    audio_dev = [torch.empty_like(audio_pin, device=device) for _ in range(2)]
    audio_dev[0].copy_(audio_pin)
    copy_stream = torch.cuda.Stream(device=device)
    copy_done = torch.cuda.Event()
    torch.cuda.synchronize()
    offsets = np.arange(B + 1, dtype=np.int64) * n_samples
    lang = np.zeros(B, dtype=np.int32)
    lengths = [n_samples] * B
    max_t = cfg.seq_len(n_samples)

    def step(k=0, upload=True):
        if upload:                                         # batch k + 1: pinned host -> HBM, overlapped with batch k's kernels
            with torch.cuda.stream(copy_stream):
                audio_dev[(k + 1) & 1].copy_(audio_pin, non_blocking=True)
                copy_done.record(copy_stream)
        tok, num = sess.run_packed(None, offsets, lang, audio_device_ptr=audio_dev[k & 1].data_ptr())
        if upload:
            copy_done.synchronize()
        if world > 1:                                      # hypotheses to rank 0 (latency-bound, < 120 KB per rank)
            dp.gather_hypotheses(dp.pack_hypotheses(tok, num, max_t), device)
        return tok, num

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    def timed(n_steps, upload):
        fence()
        t0 = time.perf_counter()
        for k in range(n_steps):
            step(k, upload)
        fence()
        el = time.perf_counter() - t0
        return dp.max_over_ranks(el, device)

    for k in range(args.warmup):
        step(k)
    elapsed = timed(args.steps, True)                      # the headline: EXACTLY --steps steps, uploads inside
    tok, num = step(0, False)
    elapsed_resident = timed(args.steps, False)            # same steps with the audio left in HBM (what round 1 reported as `value`)
    steady = None
    if args.steps < 200 and not args.no_extras:            # a short --steps run is noisy (20 x 8 ms = 0.17 s): also time 200 steps
        steady = timed(200, True)

    # ---- roofline leg: per-kernel-class HIP-event timing on the session stream (separate profiled steps)
    sess.profile(True)
    sess.profile_reset()
    for _ in range(args.profile_steps):
        sess.run_packed(None, offsets, lang, audio_device_ptr=audio_dev[0].data_ptr())
    prof = sess.profile_read()
    sess.profile(False)
    # serving option, reported next to the headline and never as `value`: two batches in flight on two sessions / HIP streams (the
    # second session borrows the same arena); the launch gaps and round tails of one graph replay are filled by the other
    import threading
    t_inflight = None

    def inflight_worker(s_, n_):
        torch.cuda.set_device(local_rank)
        for _ in range(n_):
            s_.run_packed(None, offsets, lang, audio_device_ptr=audio_dev[0].data_ptr())

    if not args.no_extras:
        sess2 = eng.SenseVoiceSession(cfg, arena_dev, arena.PRECISION_BF16, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
        inflight_worker(sess2, 3)
        torch.cuda.synchronize()
        per = max(min(args.steps, 50), 10)
        ths = [threading.Thread(target=inflight_worker, args=(s_, per)) for s_ in (sess, sess2)]
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        t_inflight = (time.perf_counter() - t1) / (2 * per)
        del sess2

    if rank == 0:
        audio_s_per_step = world * B * n_samples / cfg.sample_rate
        ms_per_step = elapsed / args.steps * 1e3
        value = audio_s_per_step * args.steps / elapsed
        flops = sensevoice_algorithmic_flops(cfg, lengths)
        total_flops = sum(flops.values())                  # base classes only: the fused kernel's work is q|k|v + attention + FSMN, counted once
        if "sanm_fused" in prof:      # q|k|v projection + attention + FSMN run as one kernel per (utterance, head)
            flops["sanm_fused"] = flops["gemm_qkv"] + flops["attention"] + flops["fsmn"]
        if "sanm_block" in prof:      # one persistent launch per SANM block: every GEMM, the attention and the FSMN of the block
            # ... of the blocks the block kernel walks: n_enc0 .. n_blocks - 1 (the scope "sanm_block" times exactly those launches; block 0 is not in it)
            flops["sanm_block"] = sum(sensevoice_algorithmic_flops(cfg, lengths, blocks=(cfg.n_enc0, cfg.n_blocks)).values())
        kernels = {}
        for name, p in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            ms = p["total_ms"] / args.profile_steps
            k = {"ms_per_step": round(ms, 4), "launches_per_step": p["launches"] // args.profile_steps}
            # with the block kernel active the per-class GEMM scopes only see the first block's separate launches: no rate for those
            covered = name in flops and not ("sanm_block" in prof and name in ("gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2", "sanm_fused", "attention", "fsmn"))
            if covered and ms > 0:
                k["tflops"] = round(flops[name] / (ms * 1e-3) / 1e12, 1)
            kernels[name] = k
        if "sanm_block" in kernels:
            dom, dom_match = ["sanm_block"], ("sanm_block8_kernel",)
            dom_desc = ("sanm_block8_kernel (q|k|v + attention + FSMN + out-proj + FFN of a SANM block, clusters of four workgroups per window; one launch walks a "
                        "whole run of blocks: 49 + 20 per step; csrc/sanm_block8.hip)")
        else:
            dom = [n for n in kernels if n.startswith("gemm_") and n != "gemm_ctc"]
            dom_desc = ("gemm_bf16_t144 / gemm_bf16_t288w (SANM out-proj / ffn2: 144 x 128 tiles, ffn1: 288 x 256 tiles; the q|k|v projection "
                        "runs inside sanm_qkv_attn_kernel and is listed under kernels.sanm_fused)")
            dom_match = ("gemm_bf16_t144<", "gemm_bf16_t288w<1,")
        gemm_ms = sum(kernels[n]["ms_per_step"] for n in dom)
        gemm_flops = sum(flops[n] for n in dom)
        gemm_launches = sum(kernels[n]["launches_per_step"] for n in dom)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        pmc_key = "sanm_block8_kernel" if "sanm_block" in kernels else "gemm_bf16_t144"
        n_blocks_in_dom = cfg.n_blocks - cfg.n_enc0 if "sanm_block" in kernels else None        # blocks the block-kernel launches walk per step
        traffic = hbm_traffic(pmc_key)
        if traffic and n_blocks_in_dom:                       # a launch walks a run of blocks: the per-block figure is the comparable one
            traffic["bytes_per_block"] = int(traffic["bytes_per_launch"] * gemm_launches / n_blocks_in_dom)
        out = {
            "metric": "audio-sec/s, SenseVoiceSmall, 8 s @ 16 kHz chunks, batch 64 per GPU (RTF = 1/value per GPU-stream)",
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SenseVoiceSmall bf16 (234 M params, 70 SANM blocks), batch={B} x {args.seconds:g} s per GPU, "
                                   "greedy CTC; step k's audio is resident in HBM when the step starts, batch k+1's host->device copy (pinned, copy stream) "
                                   "and every token-id download are inside the timed region",
                       "global_batch": world * B, "audio_seconds_per_step": audio_s_per_step,
                       "parallelism": f"dp{world} (utterance sharding, RCCL arena broadcast + hypothesis gather)"},
            "rtf": round(elapsed / (audio_s_per_step * args.steps), 8),
            "audio_s_per_s_per_gpu": round(value / world, 1),
            "audio_resident": {"audio_s_per_s_per_gpu": round(audio_s_per_step * args.steps / elapsed_resident / world, 1),
                               "ms_per_step": round(elapsed_resident / args.steps * 1e3, 3), "what": "same steps, no uploads (audio left in HBM)"},
            "steady": None if steady is None else {"steps": 200, "ms_per_step": round(steady / 200 * 1e3, 3),
                                                   "audio_s_per_s_per_gpu": round(audio_s_per_step * 200 / steady / world, 1)},
            "model_tflops_per_gpu": round(total_flops / (ms_per_step * 1e-3) / 1e12, 1),
            "roofline": {"bound": "mfma", "kernel": dom_desc,
                         "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "pmc": mfma_util(pmc_key),
                         "launches_per_step": gemm_launches, "avg_launch_us": round(gemm_ms * 1e3 / max(gemm_launches, 1), 2),
                         "blocks_per_step": n_blocks_in_dom, "avg_block_us": None if not n_blocks_in_dom else round(gemm_ms * 1e3 / n_blocks_in_dom, 2),
                         "rocprof": rocprof_avg_us(dom_match),
                         "algorithmic_gflop_per_step": round(gemm_flops / 1e9, 1),
                         "whole_step_frac": round(total_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
            "kernels": kernels,
            "inflight": None if t_inflight is None else {"batches_in_flight": 2, "audio_s_per_s_per_gpu": round(B * n_samples / cfg.sample_rate / t_inflight, 1),
                                                         "ms_per_batch": round(t_inflight * 1e3, 3)},
            "arena_broadcast_s": round(t_bcast, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, ck, audio_np)
            # spot check AT THE TIMED BATCH SIZE (the batch-64 dispatch: 288 x 256 FFN-1 / CTC tiles): CTC logits and frame arg-max of
            # utterances 0 and B-1 tapped from the full batch, bf16 engine vs f32 oracle (CHECKER ONLY)
            from oracle.sensevoice_oracle import SenseVoiceOracle
            orc = SenseVoiceOracle(cfg, ck)
            sess.taps(True)
            sess.run_packed(None, offsets, lang, audio_device_ptr=audio_dev[0].data_ptr())
            lg_all = sess.tap("logits")
            rows = sess.utterance_rows(lengths)
            sess.taps(False)
            checks = []
            for b in sorted({0, B - 1}):
                st = orc.stages(audio_np[b, 0], 0)
                r0, T = rows[b]
                lg = lg_all[r0:r0 + T]
                srt = np.sort(st["logits"], axis=1)
                diff = float(np.abs(lg - st["logits"]).max())
                safe = (srt[:, -1] - srt[:, -2]) > 2.0 * diff               # frames whose top-1 / top-2 margin exceeds twice the measured error
                checks.append({"utterance": b, "max_abs_diff": round(diff, 4), "logit_abs_max": round(float(np.abs(st["logits"]).max()), 3),
                               "frames": int(safe.size), "frames_with_safe_margin": int(safe.sum()),
                               "argmax_equal_on_those": bool(np.array_equal(lg.argmax(1)[safe], st["frame_ids"][safe]))})
            del lg_all
            out["parity_spotcheck"] = {"what": f"CTC logits tapped from the timed batch of {B}, bf16 engine vs f32 oracle", "utterances": checks}
            if not args.no_extras:
                out["secondary"] = secondary_lines(cfg, ck, audio_np, local_rank, device, cpu_leg=True)
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


def secondary_lines(cfg, ck, audio_np, local_rank, device, cpu_leg=False):
    """The other two figures BASELINE.json's metric names, measured in the same driver-run process (secondary keys, never `value`):
    configs[0] SenseVoiceSmall f32 mode, one 8 s chunk (the mode whose tokens equal the reference's) and Whisper-large-v3 bf16 on 8 s
    chunks at batch 1, 32 and 64 (encoder + prefill + 31 greedy decode steps; random weights never emit EOS)."""
    import torch
    ckm = importlib.import_module(PKG + ".checkpoints")
    cfgm = importlib.import_module(PKG + ".config")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    out = {}
    n_samples = audio_np.shape[2]
    # ---- configs[0]: f32 verification mode, batch 1 (host audio in, ids out: the reference's call shape)
    s32 = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=arena.PRECISION_F32, device_id=local_rank)
    one = [audio_np[0, 0]]
    for _ in range(3):
        s32.run(one, [0])
    t0 = time.perf_counter()
    for _ in range(10):
        s32.run(one, [0])
    dt = (time.perf_counter() - t0) / 10
    out["sensevoice_f32_b1"] = {"what": "SenseVoiceSmall f32 mode (logits within 1e-3 of the reference, tokens equal), one 8 s chunk, host audio in / ids out",
                                "ms_per_chunk": round(dt * 1e3, 3), "audio_s_per_s": round(n_samples / cfg.sample_rate / dt, 1),
                                "rtf": round(dt / (n_samples / cfg.sample_rate), 6)}
    # the token-exact mode at the headline's batch: 64 x 8 s, f32 operands (exact-f32 MFMA GEMMs)
    try:
        B64 = min(64, audio_np.shape[0])
        batch = [audio_np[b, 0] for b in range(B64)]
        s32.run(batch, [0] * B64)
        t0 = time.perf_counter()
        for _ in range(3):
            s32.run(batch, [0] * B64)
        dt = (time.perf_counter() - t0) / 3
        out["sensevoice_f32_b64"] = {"what": f"SenseVoiceSmall f32 mode, batch {B64} x 8 s, host audio in / ids out",
                                     "ms_per_step": round(dt * 1e3, 2), "audio_s_per_s": round(B64 * n_samples / cfg.sample_rate / dt, 1),
                                     "rtf": round(dt / (B64 * n_samples / cfg.sample_rate), 7)}
    except Exception as e:
        out["sensevoice_f32_b64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    s16 = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=arena.PRECISION_BF16, device_id=local_rank)
    for _ in range(3):
        s16.run(one, [0])
    t0 = time.perf_counter()
    for _ in range(20):
        s16.run(one, [0])
    dt = (time.perf_counter() - t0) / 20
    w_bytes = 2.0 * 234.0e6                                  # SenseVoiceSmall, 234.0 M parameters in bf16, read once per batch: the algorithmic bytes of a batch-1 pass
    out["sensevoice_bf16_b1"] = {"what": "SenseVoiceSmall bf16, ONE 8 s chunk per call (the metric's batch-1 point), host audio in / ids out; SANM blocks on sanm_tiles_kernel (36 workgroups per window)",
                                 "ms_per_chunk": round(dt * 1e3, 3), "audio_s_per_s": round(n_samples / cfg.sample_rate / dt, 1),
                                 "rtf": round(dt / (n_samples / cfg.sample_rate), 6),
                                 "roofline": {"bound": "hbm", "achieved": round(w_bytes / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(w_bytes / dt / 1e9 / HBM_PEAK_GBS, 4),
                                              "traffic": None, "floor_us": round(w_bytes / 6.29e12 * 1e6, 1),
                                              "note": "a latency chain (five meetings per block at memory-side latency), not a bandwidth-bound stream: DESIGN.md 4.13"}}
    del s32, s16
    # ---- Whisper-large-v3 bf16, 8 s chunks (the other model the metric names)
    try:
        wcfg = cfgm.whisper_large_v3()
        wck = ckm.synth_whisper_checkpoint(wcfg, seed=0)
        blob = arena.build_whisper_arena(wcfg, wck, arena.PRECISION_BF16, ckm.whisper_suppress_tokens(wcfg), ckm.whisper_begin_suppress_tokens(wcfg))
        a_dev = torch.from_numpy(blob).to(device)
        del blob
        ws = eng.WhisperSession(wcfg, a_dev, arena.PRECISION_BF16, local_rank, arena_device_ptr=a_dev.data_ptr(), arena_bytes=a_dev.numel())
        # BASELINE's target points are batch 1 and batch 64 on 8 s chunks (32 is configs[2]'s batch); the last entry IS configs[2]: 32 x 30 s,
        # 128 tokens per utterance (SURVEY section 8d's decode length)
        for Bw, secs, n_tok in ((1, 8, 32), (32, 8, 32), (64, 8, 32), (32, 30, 128)):
            ns = int(secs * wcfg.sample_rate)
            wav = torch.from_numpy(ckm.synth_audio("unit", Bw, ns, seed=4321)).to(device)
            offs = np.arange(Bw + 1, dtype=np.int64) * ns
            prompt = np.tile(np.array([[wcfg.sot_id, wcfg.first_language_id, wcfg.transcribe_id, wcfg.no_timestamps_id]], np.int32), (Bw, 1))
            parts = np.zeros(3)
            reps = 3
            for it in range(reps + 2):
                t0 = time.perf_counter()
                ws.encode_packed(None, offs, audio_device_ptr=wav.data_ptr())
                t1 = time.perf_counter()
                ws.prefill(prompt, want_logits=False)
                t2 = time.perf_counter()
                ws.generate(n_tok, eos_id=-1)
                t3 = time.perf_counter()
                if it >= 2:
                    parts += (t1 - t0, t2 - t1, t3 - t2)
            parts /= reps
            tot = float(parts.sum())
            alg = whisper_algorithmic(wcfg, ns, Bw, n_tok, 4)
            enc_flops, dec_bytes = alg["encoder_flops"], alg["decode_bytes_per_step"]
            out[f"whisper_large_v3_bf16_b{Bw}x{secs}s"] = {
                "ms_per_batch": round(tot * 1e3, 2), "audio_s_per_s": round(Bw * ns / wcfg.sample_rate / tot, 1),
                "rtf": round(tot / (Bw * ns / wcfg.sample_rate), 7),
                "ms": {"encode": round(parts[0] * 1e3, 2), "prefill": round(parts[1] * 1e3, 2), "decode": round(parts[2] * 1e3, 2)},
                "decode_ms_per_token": round(parts[2] / (n_tok - 1) * 1e3, 3), "tokens_per_utterance": n_tok,
                "roofline_encode": {"bound": "mfma", "achieved": round(enc_flops / parts[0] / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(enc_flops / parts[0] / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                    "what": "whole encode (log-mel, conv stem, 32 layers incl. attention and LayerNorms, cross-K/V) over its wall time"},
                "roofline_decode": {"bound": "hbm", "achieved": round(dec_bytes / (parts[2] / (n_tok - 1)) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(dec_bytes / (parts[2] / (n_tok - 1)) / 1e9 / HBM_PEAK_GBS, 4),
                                    "what": "algorithmic bytes of one decode step (decoder weights once + cross-K/V and self-K/V of every sequence) over the step time"}}
            if secs == 30:
                out[f"whisper_large_v3_bf16_b{Bw}x{secs}s"]["config"] = "BASELINE.json configs[2]"
            del wav
        del ws, a_dev
        if cpu_leg:
            # the metric's second model on this host's cores: the CPU restatement (CHECKER ONLY), batch 1 like the reference, on a bounded sample
            # (one warm-up + two timed 8 s utterances: encoder + prefill(4) + 31 decode steps each)
            from oracle.whisper_oracle import WhisperOracle
            orc = WhisperOracle(wcfg, wck, ckm.whisper_suppress_tokens(wcfg), ckm.whisper_begin_suppress_tokens(wcfg))
            n_thr = min(32, os.cpu_count() or 8)
            torch.set_num_threads(n_thr)
            ns = 8 * wcfg.sample_rate
            clips = ckm.synth_audio("unit", 3, ns, seed=4321)
            prm = [wcfg.sot_id, wcfg.first_language_id, wcfg.transcribe_id, wcfg.no_timestamps_id]
            orc.greedy([clips[0, 0]], [prm], 8)
            t0 = time.perf_counter()
            for k in (1, 2):
                orc.greedy([clips[k, 0]], [prm], 32)
            el = time.perf_counter() - t0
            out["whisper_large_v3_cpu_baseline"] = {"value": round(2 * 8.0 / el, 3), "unit": "audio-s/s", "cores": n_thr, "host_cores": int(os.cpu_count() or 0), "kind": "port",
                                                    "sample": f"2 x 8 s utterances, batch 1, encoder + prefill(4) + 31 decode steps, torch-CPU f32 oracle "
                                                              f"(oracle/whisper_oracle.py), {el:.1f} s wall"}
        del wck
    except Exception as e:                                   # a secondary must never take the headline line down
        out["whisper_large_v3_bf16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ---- BASELINE configs[4]'s streaming leg: Paraformer-large online, 64 concurrent streams, one 0.5 s chunk per step (the own line: --workload paraformer-streaming)
    try:
        pcfg = cfgm.paraformer_large()
        pck = ckm.synth_paraformer_checkpoint(pcfg, seed=0)
        S, chunk, n_chunks = 64, 8000, 8
        ps = eng.ParaformerStreamSession(pcfg, pck, precision=0, device_id=local_rank, chunk=chunk, max_streams=S)
        pa = torch.from_numpy(np.ascontiguousarray(ckm.synth_audio("kaldi", S, n_chunks * chunk, seed=1234)[:, 0].reshape(S, n_chunks, chunk).transpose(1, 0, 2))).to(device)
        sids = list(range(S))

        def pstep(i):
            if i % n_chunks == 0:
                ps.reset(-1)
            ps.step(None, sids, audio_device_ptr=pa[i % n_chunks].data_ptr())
        for i in range(n_chunks):
            pstep(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3 * n_chunks):
            pstep(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (3 * n_chunks)
        out["paraformer_large_streaming_bf16_64streams"] = {"ms_per_chunk_step": round(dt * 1e3, 3), "audio_s_per_s": round(S * chunk / pcfg.sample_rate / dt, 1),
                                                            "what": "64 streams x one 8000-sample chunk per step, audio resident in HBM, token ids returned to host; encoder layers 1..49 and "
                                                                    "the decoder blocks as one cluster launch each (csrc/stream_layers.hip, stream_dec.hip)"}
        del ps, pa, pck
    except Exception as e:
        out["paraformer_large_streaming_bf16_64streams"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def paraformer_algorithmic_flops(cfg, n_samples, n_tokens):
    """GEMM = 2MNK on un-padded rows, attention = 4 T_q T_k d; encoder as SenseVoice, decoder over `n_tokens` fired tokens."""
    d, dff, dd, feat = cfg.d_model, cfg.d_ffn, cfg.d_dec_ffn, cfg.feat_dim
    T, nb = cfg.seq_len(n_samples), cfg.n_enc0 + cfg.n_enc
    enc = 2.0 * T * (3 * d * (feat * cfg.n_enc0 + d * cfg.n_enc) + nb * (d * d + 2 * d * dff)) + 4.0 * T * T * d * nb
    cif = 2.0 * T * d * 3 * d
    N = n_tokens
    dec = cfg.n_dec * (2.0 * T * d * 2 * d + 2.0 * N * (2 * d * dd + 2 * d * d) + 4.0 * N * T * d) + cfg.n_dec3 * 2.0 * N * 2 * d * dd
    return enc + cif + dec + 2.0 * N * d * cfg.vocab


def main_paraformer(args):
    """Paraformer-large (non-streaming) bf16, batch x 8 s: fbank -> 50 SANM blocks -> CIF -> 16+1 decoder layers -> ids."""
    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus and torch.cuda.is_available()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = cfgm.paraformer_large()
    n_samples, B = int(args.seconds * cfg.sample_rate), args.batch
    ck = blob = None
    if rank == 0:
        ck = ckm.synth_paraformer_checkpoint(cfg, seed=0)
        blob = arena.build_paraformer_arena(cfg, ck, arena.PRECISION_BF16)
    arena_dev = dp.broadcast_arena(blob, device)
    sess = eng.ParaformerSession(cfg, arena_dev, arena.PRECISION_BF16, local_rank, arena_device_ptr=arena_dev.data_ptr(),
                                 arena_bytes=arena_dev.numel())
    audio_np = ckm.synth_audio("kaldi", B, n_samples, seed=1234 + rank)
    audio_dev = torch.from_numpy(audio_np).to(device)
    offsets = np.arange(B + 1, dtype=np.int64) * n_samples
    max_t = cfg.seq_len(n_samples)

    def step():
        tok, num = sess.run_packed(None, offsets, audio_device_ptr=audio_dev.data_ptr())
        if world > 1:
            dp.gather_hypotheses(dp.pack_hypotheses(tok, num, max_t), device)
        return tok, num

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tok, num = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = dp.max_over_ranks(elapsed, device)
    sess.profile(True)
    sess.profile_reset()
    for _ in range(args.profile_steps):
        sess.run_packed(None, offsets, audio_device_ptr=audio_dev.data_ptr())
    prof = sess.profile_read()
    sess.profile(False)
    if rank == 0:
        audio_s = world * B * n_samples / cfg.sample_rate
        ms = elapsed / args.steps * 1e3
        flops = sum(paraformer_algorithmic_flops(cfg, n_samples, int(n)) for n in num)
        kernels = {k: {"ms_per_step": round(v["total_ms"] / args.profile_steps, 4), "launches_per_step": v["launches"] // args.profile_steps}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        gemm_ms = sum(v["ms_per_step"] for k, v in kernels.items() if k.startswith("gemm_"))
        T, nb, d = cfg.seq_len(n_samples), cfg.n_enc0 + cfg.n_enc, cfg.d_model
        enc_gemm = B * 2.0 * T * (3 * d * (cfg.feat_dim * cfg.n_enc0 + d * cfg.n_enc) + nb * (d * d + 2 * d * cfg.d_ffn))
        if "sanm_block" in kernels:        # one launch per encoder block: its GEMMs + attention + FSMN (block 0 keeps the separate launches)
            enc_gemm += B * (4.0 * T * T * d + 2.0 * T * d * cfg.fsmn_kernel) * nb
            enc_ms = kernels["sanm_block"]["ms_per_step"] + sum(kernels[k]["ms_per_step"] for k in ("sanm_fused", "gemm_qkv", "gemm_ffn1", "gemm_ffn2") if k in kernels)
            dom_desc = "sanm_block8_kernel (one launch walks the SANM encoder blocks: q|k|v + attention + FSMN + out-proj + FFN; csrc/sanm_block8.hip)"
        else:
            enc_ms = sum(kernels[k]["ms_per_step"] for k in ("gemm_qkv", "gemm_ffn1", "gemm_ffn2") if k in kernels)
            enc_ms += kernels.get("gemm_out", {"ms_per_step": 0})["ms_per_step"]
            dom_desc = "gemm_bf16_pipe (encoder qkv/out/ffn launches; the out class also holds the vocab GEMM)"
        ach = enc_gemm / (enc_ms * 1e-3) / 1e12 if enc_ms else 0.0
        out = {"metric": "audio-sec/s, Paraformer-large (non-streaming), 8 s @ 16 kHz chunks, batch %d per GPU" % B,
               "value": round(audio_s * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "Paraformer-large bf16 (50 SANM blocks, CIF predictor, 16+1 decoder layers, vocab 8404), batch=%d x %g s "
                                      "per GPU, audio resident in HBM, token ids returned to host" % (B, args.seconds),
                          "global_batch": world * B, "audio_seconds_per_step": audio_s, "tokens_per_step": int(np.sum(num)),
                          "parallelism": f"dp{world}"},
               "rtf": round(elapsed / (audio_s * args.steps), 8),
               "model_tflops_per_gpu": round(flops / (ms * 1e-3) / 1e12, 1),
               "roofline": {"bound": "mfma", "kernel": dom_desc,
                            "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "gemm_ms_per_step": round(gemm_ms, 3)},
               "kernels": kernels}
        if world == 1 and not args.no_cpu_baseline:
            from oracle.paraformer_oracle import ParaformerOracle
            orc = ParaformerOracle(cfg, ck)
            torch.set_num_threads(min(16, os.cpu_count() or 8))
            orc(audio_np[0, 0])
            n_done, t1 = 0, time.perf_counter()
            while True:
                orc(audio_np[n_done % B, 0])
                n_done += 1
                el = time.perf_counter() - t1
                if (el >= 10.0 and n_done >= 4) or n_done >= 64:
                    break
            out["cpu_baseline"] = {"value": round(n_done * args.seconds / el, 2), "unit": "audio-s/s", "cores": int(torch.get_num_threads()),
                                   "kind": "port", "sample": f"{n_done} x {args.seconds:g} s utterances, batch 1, torch-CPU f32 oracle "
                                   f"(oracle/paraformer_oracle.py), {el:.1f} s wall"}
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


def main_paraformer_streaming(args):
    """Paraformer-large streaming, chunk = 8000 samples (0.5 s): `--batch` concurrent streams advance one chunk per step."""
    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus and torch.cuda.is_available()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = cfgm.paraformer_large()
    S, chunk = args.batch, 8000
    ck = ckm.synth_paraformer_checkpoint(cfg, seed=0)                   # every rank builds the same arena (streams pin to their GPU)
    sess = eng.ParaformerStreamSession(cfg, ck, precision=0, device_id=local_rank, chunk=chunk, max_streams=S)
    n_chunks = 8
    audio_np = ckm.synth_audio("kaldi", S, n_chunks * chunk, seed=1234 + rank)[:, 0].reshape(S, n_chunks, chunk)
    audio_dev = torch.from_numpy(np.ascontiguousarray(audio_np.transpose(1, 0, 2))).to(device)      # [chunk index][stream][samples]
    sids = list(range(S))
    tokens = 0

    def step(i):
        nonlocal tokens
        k = i % n_chunks
        if k == 0:
            sess.reset(-1)
        out = sess.step(None, sids, audio_device_ptr=audio_dev[k].data_ptr())
        tokens += sum(o.size for o in out)

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    tokens = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = dp.max_over_ranks(elapsed, device)
    sess.profile(True)
    sess.profile_reset()
    for i in range(args.profile_steps):
        step(i)
    prof = sess.profile_read()
    sess.profile(False)
    if rank == 0:
        audio_s = world * S * chunk / cfg.sample_rate
        ms = elapsed / args.steps * 1e3
        kernels = {k: {"ms_per_step": round(v["total_ms"] / args.profile_steps, 4), "launches_per_step": v["launches"] // args.profile_steps}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        wbytes = 2.0 * sum(v.size for k, v in ck.items() if not k.startswith("frontend."))
        out = {"metric": "audio-sec/s, Paraformer-large streaming (chunk = 8000 samples), %d concurrent streams per GPU" % S,
               "value": round(audio_s * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "Paraformer-large streaming bf16: %d streams x one 0.5 s chunk per step (13 encoder rows per stream, K/V histories "
                                      "36 / 9 rows), decoder on fired frames, audio resident in HBM, token ids returned to host" % S,
                          "global_batch": world * S, "audio_seconds_per_step": audio_s, "tokens_per_step": round(tokens / max(args.steps, 1), 1),
                          "parallelism": f"dp{world} (streams pinned to their GPU)",
                          "path": "two cluster launches per step (stream_layers / stream_dec)" if "stream_layers" in kernels else
                                  "per-launch path (more active streams than the session's fused_max: the per-launch GEMMs amortise the weights over all rows)",
                          "dispatch": sess.stream_stats()},
               "rtf": round(elapsed / (audio_s * args.steps), 8), "chunk_latency_ms": round(ms, 3),
               "roofline": {"bound": "hbm", "kernel": "whole chunk step (weights streamed once per step; 13 x %d rows keep every GEMM weight-bound). Encoder layers 1..49 "
                                                       "and the decoder blocks are one launch each (stream_layers_kernel / stream_dec_kernel: clusters of four "
                                                       "workgroups per stream; every workgroup streams a quarter of each layer's weights through its CU's 64 B/clk vector-memory "
                                                       "path -- 1.57 MB per layer = 11.7 us -- and meets its cluster 4 / 5 times per layer through memory)" % S,
                            "achieved": round(wbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(wbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                            "launches_per_step": sum(v["launches_per_step"] for v in kernels.values()),
                            "fused_layer_us": {k: round(kernels[k]["ms_per_step"] * 1e3 / nl, 2) for k, nl in (("stream_layers", cfg.n_enc0 + cfg.n_enc - 1), ("stream_dec", cfg.n_dec + cfg.n_dec3)) if k in kernels}},
               "kernels": kernels}
        if world == 1 and not args.no_cpu_baseline:
            from oracle.paraformer_streaming_oracle import ParaformerStreamingOracle
            orc = ParaformerStreamingOracle(cfg, ck, chunk=chunk)
            torch.set_num_threads(min(16, os.cpu_count() or 8))
            orc.run(audio_np[0, :2].reshape(-1))
            n_done, t1 = 0, time.perf_counter()
            while True:
                orc.run(audio_np[n_done % S].reshape(-1))
                n_done += 1
                el = time.perf_counter() - t1
                if (el >= 10.0 and n_done >= 2) or n_done >= 16:
                    break
            out["cpu_baseline"] = {"value": round(n_done * n_chunks * chunk / cfg.sample_rate / el, 2), "unit": "audio-s/s", "cores": int(torch.get_num_threads()),
                                   "kind": "port", "sample": f"{n_done} streams x {n_chunks} chunks, one stream at a time, torch-CPU f32 oracle "
                                   f"(oracle/paraformer_streaming_oracle.py), {el:.1f} s wall"}
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


def whisper_algorithmic(cfg, n_samples, B, n_tokens, prompt_len, fp8=False, weight_bits=None):
    """Algorithmic FLOPs / bytes: encoder GEMM 2MNK + attention 4 T^2 d per layer + conv stem + cross-KV
    projection; decode: weights read once per step for the batch + cross-KV + self-KV streamed per utterance."""
    d, dff, Le, Ld, T = cfg.d_model, cfg.d_ffn, cfg.n_enc_layers, cfg.n_dec_layers, cfg.n_enc_pos(n_samples)
    F = cfg.n_frames(n_samples)
    enc = Le * (2.0 * T * d * (4 * d + 2 * dff) + 4.0 * T * T * d) + 2.0 * F * d * 3 * cfg.n_mels + 2.0 * T * d * 3 * d \
        + 2.0 * T * d * 2 * Ld * d + F * (2.0 * cfg.nfft * 2 * (cfg.nfft // 2 + 1) + 2.0 * (cfg.nfft // 2 + 1) * cfg.n_mels)
    dec_w_params = Ld * (4 * d * d + 2 * d * d + 2 * d * dff) + cfg.vocab * d
    dec_tok = 2.0 * dec_w_params + Ld * 4.0 * T * d
    step_bytes = 2.0 * dec_w_params + B * Ld * 2 * T * d * 2.0          # bf16 weights once per step + bf16 cross-KV per utterance
    if fp8:                                                              # FP8W mode: the decoder layers' projections and the cross-K/V are bytes (proj_out stays bf16)
        step_bytes = (weight_bits / 8.0 if weight_bits else 1.0) * (dec_w_params - cfg.vocab * d) + 2.0 * cfg.vocab * d + B * Ld * 2 * T * d * 1.0      # (MXFP4W: 4.25 bits per weight)
    return {"encoder_flops": B * enc, "decode_flops_per_step": B * dec_tok, "decode_bytes_per_step": step_bytes}


def main_whisper(args):
    """Whisper-large-v3 bf16: encoder + fused cross-KV once per batch, then prefill(4-token prompt) + greedy decode of a
    FIXED number of tokens (random weights never emit EOS; SURVEY.md section 8d: 32 tokens per 8 s chunk, 128 per 30 s)."""
    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = cfgm.whisper_large_v3()
    B = args.batch
    n_samples = int(args.seconds * cfg.sample_rate)
    n_tok = args.decode_tokens or 32 * int(np.ceil(args.seconds / 8.0))        # 32 tokens per 8 s chunk, 128 per 30 s
    blob = ck = None
    if rank == 0:
        ck = ckm.synth_whisper_checkpoint(cfg, seed=0)
        blob = arena.build_whisper_arena(cfg, ck, arena.PRECISION_BF16, ckm.whisper_suppress_tokens(cfg), ckm.whisper_begin_suppress_tokens(cfg))
        if world > 1 or args.no_cpu_baseline:
            ck = None
    t0 = time.perf_counter()
    arena_dev = dp.broadcast_arena(blob, device)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    blob = None
    if args.fp8mm or args.mxfp4: args.fp8 = True
    prec = arena.PRECISION_FP8MM if args.fp8mm else arena.PRECISION_MXFP4W if args.mxfp4 else arena.PRECISION_FP8W if args.fp8 else arena.PRECISION_BF16
    sess = eng.WhisperSession(cfg, arena_dev, prec, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
    audio_np = ckm.synth_audio("unit", B, n_samples, seed=1234 + rank)
    audio_dev = torch.from_numpy(audio_np).to(device)
    offsets = np.arange(B + 1, dtype=np.int64) * n_samples
    prompt = np.tile(np.array([[cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]], np.int32), (B, 1))
    t_parts = {"encode": 0.0, "prefill": 0.0, "decode": 0.0}

    def step(record=False):
        t0 = time.perf_counter()
        sess.encode_packed(None, offsets, audio_device_ptr=audio_dev.data_ptr())
        t1 = time.perf_counter()
        sess.prefill(prompt, want_logits=False)
        t2 = time.perf_counter()
        toks = sess.generate(n_tok, eos_id=-1)
        t3 = time.perf_counter()
        if record:
            t_parts["encode"] += t1 - t0; t_parts["prefill"] += t2 - t1; t_parts["decode"] += t3 - t2
        if world > 1:
            tok = np.stack(toks)
            dp.gather_hypotheses(dp.pack_hypotheses(tok, np.full(B, n_tok, np.int32), n_tok), device)
        return toks

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = dp.max_over_ranks(elapsed, device)
    # serving option (--inflight N, reported next to the headline, never as `value`): N sessions on N HIP streams, each working through
    # its own batches of the same size -- the encoder of one batch (compute-bound) runs inside the decode of another (a latency-bound
    # chain that leaves most CUs idle)
    inflight = None
    if args.inflight > 1:
        import threading
        others = [eng.WhisperSession(cfg, arena_dev, prec, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
                  for _ in range(args.inflight - 1)]
        sessions = [sess] + others

        def worker(s_, n_):
            torch.cuda.set_device(local_rank)
            for _ in range(n_):
                s_.encode_packed(None, offsets, audio_device_ptr=audio_dev.data_ptr())
                s_.prefill(prompt, want_logits=False)
                s_.generate(n_tok, eos_id=-1)

        for s_ in others:
            worker(s_, 1)
        fence()
        per = max(args.steps // args.inflight, 1)
        ths = [threading.Thread(target=worker, args=(s_, per)) for s_ in sessions]
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        fence()
        el2 = time.perf_counter() - t1
        inflight = {"batches_in_flight": args.inflight, "audio_s_per_s_per_gpu": round(B * args.seconds * per * args.inflight / el2, 1),
                    "ms_per_batch": round(el2 / (per * args.inflight) * 1e3, 2)}
    sess.profile(True)
    sess.profile_reset()
    step()
    prof = sess.profile_read()
    sess.profile(False)
    if rank == 0:
        alg = whisper_algorithmic(cfg, n_samples, B, n_tok, 4, fp8=args.fp8, weight_bits=4.25 if args.mxfp4 else None)
        audio_s = world * B * args.seconds
        kernels = {k: {"ms_per_step": round(v["total_ms"], 3), "launches_per_step": v["launches"]}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        enc_gemm = [k for k in kernels if k.startswith("gemm_")]
        enc_gemm_ms = sum(kernels[k]["ms_per_step"] for k in enc_gemm)
        d, dff, Le, T = cfg.d_model, cfg.d_ffn, cfg.n_enc_layers, cfg.n_enc_pos(n_samples)
        enc_gemm_flops = B * (Le * 2.0 * T * d * (4 * d + 2 * dff) + 2.0 * T * d * 2 * cfg.n_dec_layers * d)
        achieved = enc_gemm_flops / (enc_gemm_ms * 1e-3) / 1e12 if enc_gemm_ms else 0.0
        enc_peak = MFMA_BF16_PEAK_TFLOPS
        if args.fp8mm:                                                     # fc1 / fc2 run on the FP8 pipe (2 x the bf16 rate): price the launch set against the blended peak
            ffn_share = B * Le * 2.0 * T * d * 2 * dff / enc_gemm_flops
            enc_peak = 1.0 / ((1.0 - ffn_share) / MFMA_BF16_PEAK_TFLOPS + ffn_share / (2.0 * MFMA_BF16_PEAK_TFLOPS))
        t_dec = t_parts["decode"] / args.steps
        out = {
            "metric": "audio-sec/s, Whisper-large-v3, %g s @ 16 kHz chunks, batch %d per GPU, greedy, %d tokens/utterance" % (args.seconds, B, n_tok)
                      + (" [opt-in MXFP4W mode: decoder projections stored as OCP MXFP4 (e2m1 + e8m0 per 32), cross-K/V as e4m3, NOT the reference's precision]" if args.mxfp4 else
                         " [opt-in FP8W mode: decoder projections + cross-K/V stored as e4m3, NOT the reference's precision]" if args.fp8 else "")
                      + (" [+ FP8MM: encoder fc1 / fc2 on the FP8 matrix pipe, e4m3 activations]" if args.fp8mm else ""),
            "value": round(audio_s * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (MXFP4 storage of decoder weights, e4m3 cross-K/V)" if args.mxfp4 else "bf16 (e4m3 storage of decoder weights and cross-K/V)" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": "Whisper-large-v3 %s (1.54 B params), batch=%d x %g s per GPU, encoder + cross-KV + prefill(4) + %d greedy decode steps, audio resident in HBM" % ("bf16 + FP8MM" if args.fp8mm else "bf16 + MXFP4W" if args.mxfp4 else "bf16 + FP8W" if args.fp8 else "bf16", B, args.seconds, n_tok - 1),
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "kv_cache": "self-KV paged (16-position pages, block table shared by the layers); cross-K/V ragged extents" if os.environ.get("ASR_KV_PAGED", "1") != "0" else "contiguous extents (ASR_KV_PAGED=0)"},
            "rtf": round(elapsed / (audio_s * args.steps), 7),
            "ms": {k: round(v / args.steps * 1e3, 2) for k, v in t_parts.items()},
            "decode_ms_per_token": round(t_dec / max(n_tok - 1, 1) * 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "encoder GEMM launches: qkv / out / fc1 / fc2 / cross-KV (gemm_bf16_ppp: persistent ping-pong 256 x 256 tiles, csrc/gemm_pp.hip, "
                                                      "where they fill whole rounds of the chip, else gemm_bf16_t144 / gemm_bf16_pipe)", "achieved": round(achieved, 1),
                         "peak": round(enc_peak, 1), "unit": "TFLOP/s", "frac": round(achieved / enc_peak, 4), "traffic": None,
                         **({"peak_note": "flop-weighted harmonic blend of 2.5 PF (bf16 launches) and 5 PF (fc1 / fc2 on v_mfma_scale_f32_16x16x128_f8f6f4)"} if args.fp8mm else {}),
                         "pmc": mfma_util("gemm_bf16_ppp", "whisper_mfma_util.json")},
            "roofline_decode": {"bound": "hbm", "kernel": "decode step (weights + cross-KV stream)",
                                "achieved": round(alg["decode_bytes_per_step"] / (t_dec / max(n_tok - 1, 1)) / 1e9, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(alg["decode_bytes_per_step"] / (t_dec / max(n_tok - 1, 1)) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels, "arena_broadcast_s": round(t_bcast, 3),
        }
        if inflight:
            out["inflight"] = inflight
        if world == 1 and not args.no_cpu_baseline:
            from oracle.whisper_oracle import WhisperOracle                # CHECKER ONLY: the CPU restatement, batch 1 like the reference
            orc = WhisperOracle(cfg, ck, ckm.whisper_suppress_tokens(cfg), ckm.whisper_begin_suppress_tokens(cfg))
            torch.set_num_threads(min(32, os.cpu_count() or 8))
            n_done, t1, first = 0, time.perf_counter(), None
            while True:
                r = orc.greedy([audio_np[n_done % B, 0]], [prompt[0].tolist()], n_tok)
                first = first or r
                n_done += 1
                el = time.perf_counter() - t1
                if el >= 15.0 or n_done >= 8:
                    break
            out["cpu_baseline"] = {"value": round(n_done * args.seconds / el, 2), "unit": "audio-s/s", "cores": int(torch.get_num_threads()),
                                   "host_cores": int(os.cpu_count() or 0), "kind": "port",
                                   "sample": f"{n_done} x {args.seconds:g} s utterances, batch 1, encoder + prefill(4) + {n_tok - 1} decode steps, torch-CPU f32 "
                                             f"oracle (oracle/whisper_oracle.py), {el:.1f} s wall"}
            # spot check at full size (random weights: margins are tiny, so logits are compared, not ids): utterance 0's prefill logits, bf16 path vs oracle
            sess.encode_packed(None, offsets, audio_device_ptr=audio_dev.data_ptr())
            _, lg = sess.prefill(prompt)
            ref0 = first["logits"][0][0]
            out["parity_spotcheck"] = {"what": "prefill logits of utterance 0, %s engine vs f32 oracle" % ("FP8MM" if args.fp8mm else "MXFP4W" if args.mxfp4 else "FP8W" if args.fp8 else "bf16"), "max_abs_diff": round(float(np.abs(lg[0] - ref0).max()), 4),
                                       "logit_abs_max": round(float(np.abs(ref0).max()), 3)}
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


def main_qwen(args):
    """Qwen3-ASR-0.6B bf16 greedy: one prefill launch (log-mel + conv stem + windowed encoder + prompt assembly + decoder prefill) and a
    FIXED number of decode steps (random weights never emit a stop id; SURVEY.md section 8d: 32 tokens per 8 s chunk, 128 per 30 s)."""
    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = cfgm.qwen_asr_0p6b()
    B = args.batch
    n_samples = int(args.seconds * cfg.sample_rate)
    n_tok = args.decode_tokens or 32 * int(np.ceil(args.seconds / 8.0))        # 32 tokens per 8 s chunk, 128 per 30 s
    blob = ck = None
    if rank == 0:
        ck = ckm.synth_qwen_asr_checkpoint(cfg, seed=0)
        blob = arena.build_qwen_asr_arena(cfg, ck, arena.PRECISION_BF16)
        if world > 1 or args.no_cpu_baseline:
            ck = None
    t0 = time.perf_counter()
    arena_dev = dp.broadcast_arena(blob, device)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    blob = None
    qprec = arena.PRECISION_MXFP4W if args.mxfp4 else arena.PRECISION_FP8W if args.fp8 else arena.PRECISION_BF16         # --fp8: the decoder's projections as e4m3 bytes (opt-in, a secondary figure)
    sess = eng.QwenAsrSession(cfg, arena_dev, qprec, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
    audio_np = ckm.synth_audio("unit", B, n_samples, seed=1234 + rank)
    audio_dev = torch.from_numpy(audio_np).to(device)
    offsets = np.arange(B + 1, dtype=np.int64) * n_samples
    # prompt geometry of the reference host: head (3) + suffix (6) ids before the audio, tail (8) + language tail (2) after it
    pre, post = [list(range(1000, 1009))], [list(range(2000, 2010))]
    t_parts = {"prefill": 0.0, "decode": 0.0}

    def step(record=False):
        t0 = time.perf_counter()
        sess.prefill_packed(None, offsets, pre, post, want_logits=False, audio_device_ptr=audio_dev.data_ptr())
        t1 = time.perf_counter()
        toks = sess.generate(n_tok, stop_ids=()) if args.beam <= 1 else [h[0][0] for h in sess.beam_search(args.beam, n_tok)]
        t2 = time.perf_counter()
        if record:
            t_parts["prefill"] += t1 - t0; t_parts["decode"] += t2 - t1
        if world > 1:
            dp.gather_hypotheses(dp.pack_hypotheses(np.stack(toks), np.full(B, n_tok, np.int32), n_tok), device)
        return toks

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = dp.max_over_ranks(elapsed, device)
    inflight = None
    if args.inflight > 1:                                  # serving option, see main_whisper: N batches in flight on N sessions / HIP streams
        import threading
        others = [eng.QwenAsrSession(cfg, arena_dev, qprec, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
                  for _ in range(args.inflight - 1)]

        def worker(s_, n_):
            torch.cuda.set_device(local_rank)
            for _ in range(n_):
                s_.prefill_packed(None, offsets, pre, post, want_logits=False, audio_device_ptr=audio_dev.data_ptr())
                s_.generate(n_tok, stop_ids=()) if args.beam <= 1 else s_.beam_search(args.beam, n_tok)

        for s_ in others:
            worker(s_, 1)
        fence()
        per = max(args.steps // args.inflight, 1)
        ths = [threading.Thread(target=worker, args=(s_, per)) for s_ in [sess] + others]
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        fence()
        el2 = time.perf_counter() - t1
        inflight = {"batches_in_flight": args.inflight, "audio_s_per_s_per_gpu": round(B * args.seconds * per * args.inflight / el2, 1),
                    "ms_per_batch": round(el2 / (per * args.inflight) * 1e3, 2)}
    sess.profile(True)
    sess.profile_reset()
    step()
    prof = sess.profile_read()
    sess.profile(False)
    if rank == 0:
        audio_s = world * B * args.seconds
        n_audio = sess.audio_tokens(n_samples)
        L = len(pre[0]) + n_audio + len(post[0])
        de, dff, d, I = cfg.enc_d, cfg.enc_ffn, cfg.d_model, cfg.d_ffn
        qkvn = (cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.d_head
        dec_params = cfg.n_layers * (qkvn * d + cfg.n_heads * cfg.d_head * d + 3 * I * d)
        # encoder + stem + decoder-prefill GEMM FLOPs per utterance (2MNK, unpadded channel counts)
        C, chunks = cfg.conv_channels, (cfg.n_frames(n_samples) + cfg.chunk - 1) // cfg.chunk
        stem = chunks * (2.0 * 3200 * C * 9 + 2.0 * 800 * C * 9 * C + 2.0 * 208 * C * 9 * C + 2.0 * 13 * de * 16 * C)
        enc = n_audio * (cfg.n_enc_layers * 2.0 * de * (4 * de + 2 * dff) + 2.0 * de * de + 2.0 * de * d)
        pre_dec = L * 2.0 * dec_params + 2.0 * d * cfg.vocab
        kernels = {k: {"ms_per_step": round(v["total_ms"], 3), "launches_per_step": v["launches"]}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        t_pre, t_dec = t_parts["prefill"] / args.steps, t_parts["decode"] / args.steps
        step_bytes = ((4.25 / 8 if args.mxfp4 else 1.0) if (args.fp8 or args.mxfp4) and B * max(args.beam, 1) <= 64 else 2.0) * dec_params + 2.0 * cfg.vocab * d + B * max(args.beam, 1) * cfg.n_layers * 2 * cfg.n_kv_heads * cfg.d_head * 2.0 * (L + n_tok / 2)
        per_tok = t_dec / max(n_tok - 1, 1)
        mode = "greedy" if args.beam <= 1 else "beam %d (%d hypothesis rows per step)" % (args.beam, B * args.beam)
        flops = B * (stem + enc + pre_dec)
        out = {
            "metric": "audio-sec/s, Qwen3-ASR-0.6B, %g s @ 16 kHz chunks, batch %d per GPU, %s, %d tokens/utterance" % (args.seconds, B, mode, n_tok),
            "value": round(audio_s * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (MXFP4 storage of the decoder projections)" if args.mxfp4 else "bf16 (e4m3 storage of the decoder projections)" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": "Qwen3-ASR-0.6B bf16 (18-layer windowed audio encoder + 28-layer Qwen3 decoder), batch=%d x %g s per GPU, prefill of "
                                   "%d positions (%d audio tokens) + %d %s decode steps, audio resident in HBM" % (B, args.seconds, L, n_audio, n_tok - 1, mode),
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "rtf": round(elapsed / (audio_s * args.steps), 7),
            "ms": {k: round(v / args.steps * 1e3, 2) for k, v in t_parts.items()},
            "decode_ms_per_token": round(per_tok * 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "prefill launch (stem + encoder + decoder-prefill GEMMs)", "achieved": round(flops / t_pre / 1e12, 1),
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / t_pre / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None},
            "roofline_decode": {"bound": "hbm", "kernel": "decode step (decoder weights + lm_head + KV cache stream)",
                                "achieved": round(step_bytes / per_tok / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(step_bytes / per_tok / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels, "arena_broadcast_s": round(t_bcast, 3),
        }
        if inflight:
            out["inflight"] = inflight
        if world == 1 and not args.no_cpu_baseline:
            from oracle.qwen_asr_oracle import QwenAsrOracle
            orc = QwenAsrOracle(cfg, ck, pre[0][:3], post[0], pre[0][3:])          # CHECKER ONLY: the CPU restatement, batch 1 like the reference
            torch.set_num_threads(min(32, os.cpu_count() or 8))
            n_done, t1, first = 0, time.perf_counter(), None
            while True:
                r = orc.greedy(audio_np[n_done % B, 0], n_tok)
                first = first or r
                n_done += 1
                el = time.perf_counter() - t1
                if el >= 15.0 or n_done >= 16:
                    break
            _, lg, _ = sess.prefill_packed(None, offsets, pre, post, want_logits=True, audio_device_ptr=audio_dev.data_ptr())
            out["parity_spotcheck"] = {"what": "prefill logits of utterance 0, %s engine vs f32 oracle" % ("FP8MM" if args.fp8mm else "FP8W" if args.fp8 else "bf16"), "max_abs_diff": round(float(np.abs(lg[0] - first["logits"][0]).max()), 4),
                                       "logit_abs_max": round(float(np.abs(first["logits"][0]).max()), 3)}
            out["cpu_baseline"] = {"value": round(n_done * args.seconds / el, 2), "unit": "audio-s/s", "cores": int(torch.get_num_threads()),
                                   "host_cores": int(os.cpu_count() or 0), "kind": "port",
                                   "sample": f"{n_done} x {args.seconds:g} s utterances, batch 1, prefill + {n_tok - 1} decode steps, torch-CPU f32 oracle "
                                             f"(oracle/qwen_asr_oracle.py), {el:.1f} s wall"}
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


def main_mixed(args):
    """BASELINE.json configs[4], the single-GPU part: Qwen3-ASR-0.6B (greedy -- the reference has no beam decoder) over batches of 8 s
    utterances AND Paraformer-large streaming chunk steps, CONCURRENTLY on one GPU: two native sessions, each on its own HIP stream,
    driven by two host threads (ctypes releases the GIL inside the C ABI). Both paths are chains of small dependent launches that
    leave most CUs idle, so they overlap. A step = one Qwen batch (prefill + decode); the streaming thread advances chunk steps for as
    long as the K Qwen steps take. value = audio seconds both paths processed / wall time; the solo rates are measured first."""
    import threading
    import torch
    import torch.distributed as dist
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    arena = importlib.import_module(PKG + ".arena")
    eng = importlib.import_module(PKG + ".engine")
    dp = importlib.import_module(PKG + ".dist")
    rank, local_rank, world = dp.init_from_env()
    assert world == args.gpus and torch.cuda.is_available()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # ---- Qwen3-ASR batch path
    qcfg = cfgm.qwen_asr_0p6b()
    B, n_samples = args.batch, int(args.seconds * qcfg.sample_rate)
    n_tok = args.decode_tokens or 32 * int(np.ceil(args.seconds / 8.0))        # 32 tokens per 8 s chunk, 128 per 30 s
    blob = None
    if rank == 0:
        blob = arena.build_qwen_asr_arena(qcfg, ckm.synth_qwen_asr_checkpoint(qcfg, seed=0), arena.PRECISION_BF16)
    arena_dev = dp.broadcast_arena(blob, device)
    blob = None
    qsess = eng.QwenAsrSession(qcfg, arena_dev, arena.PRECISION_BF16, local_rank, arena_device_ptr=arena_dev.data_ptr(), arena_bytes=arena_dev.numel())
    q_audio = torch.from_numpy(ckm.synth_audio("unit", B, n_samples, seed=1234 + rank)).to(device)
    q_offs = np.arange(B + 1, dtype=np.int64) * n_samples
    pre, post = [list(range(1000, 1009))], [list(range(2000, 2010))]
    # ---- Paraformer streaming path
    pcfg = cfgm.paraformer_large()
    S, chunk, n_chunks = args.streams, 8000, 8
    psess = eng.ParaformerStreamSession(pcfg, ckm.synth_paraformer_checkpoint(pcfg, seed=0), precision=0, device_id=local_rank, chunk=chunk, max_streams=S)
    p_np = ckm.synth_audio("kaldi", S, n_chunks * chunk, seed=4321 + rank)[:, 0].reshape(S, n_chunks, chunk)
    p_audio = torch.from_numpy(np.ascontiguousarray(p_np.transpose(1, 0, 2))).to(device)
    sids = list(range(S))

    def qwen_step():
        qsess.prefill_packed(None, q_offs, pre, post, want_logits=False, audio_device_ptr=q_audio.data_ptr())
        return qsess.generate(n_tok, stop_ids=()) if args.beam <= 1 else qsess.beam_search(args.beam, n_tok)

    def stream_step(i):
        k = i % n_chunks
        if k == 0:
            psess.reset(-1)
        psess.step(None, sids, audio_device_ptr=p_audio[k].data_ptr())

    def fence():
        if world > 1:
            dp.barrier(device)
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 1)):
        qwen_step()
    for i in range(2 * n_chunks):
        stream_step(i)
    # solo rates
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        qwen_step()
    fence()
    q_solo = (time.perf_counter() - t0) / args.steps
    t0 = time.perf_counter()
    for i in range(4 * n_chunks):
        stream_step(i)
    fence()
    p_solo = (time.perf_counter() - t0) / (4 * n_chunks)
    # concurrent
    stop = threading.Event()
    chunks_done = [0]

    def stream_loop():
        torch.cuda.set_device(local_rank)
        i = 0
        while not stop.is_set():
            stream_step(i)
            i += 1
        chunks_done[0] = i

    fence()
    th = threading.Thread(target=stream_loop)
    t0 = time.perf_counter()
    th.start()
    for _ in range(args.steps):
        qwen_step()
    stop.set()
    th.join()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = dp.max_over_ranks(elapsed, device)
        c = torch.tensor([chunks_done[0]], dtype=torch.float64, device=device)
        with dp.foreign_section(device):
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            total_chunks = float(c.item())
    else:
        total_chunks = float(chunks_done[0])
    if rank == 0:
        q_audio_s = world * args.steps * B * args.seconds
        p_audio_s = total_chunks * S * chunk / pcfg.sample_rate
        solo_q, solo_p = B * args.seconds / q_solo, S * chunk / pcfg.sample_rate / p_solo
        mode = "greedy" if args.beam <= 1 else "beam=%d" % args.beam
        out = {"metric": "audio-sec/s, Qwen3-ASR-0.6B %s (batch %d x %g s) + Paraformer-large streaming (%d streams, chunk 8000) concurrently on each GPU" % (mode, B, args.seconds, S),
               "value": round((q_audio_s + p_audio_s) / elapsed, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "config": {"workload": "mixed: Qwen3-ASR-0.6B bf16 batches (prefill + %d %s steps) and Paraformer-large streaming chunk steps, two sessions on "
                                      "two HIP streams driven by two host threads" % (n_tok - 1, mode),
                          "global_batch": world * B, "streams": world * S, "parallelism": f"dp{world}"},
               "concurrent": {"qwen_audio_s_per_s": round(q_audio_s / elapsed, 1), "streaming_audio_s_per_s": round(p_audio_s / elapsed, 1),
                              "streaming_chunk_steps": int(total_chunks), "streaming_ms_per_chunk_step": round(elapsed / max(total_chunks / world, 1) * 1e3, 2)},
               "solo_per_gpu": {"qwen_audio_s_per_s": round(solo_q, 1), "qwen_ms_per_step": round(q_solo * 1e3, 2), "streaming_audio_s_per_s": round(solo_p, 1),
                                "streaming_ms_per_chunk_step": round(p_solo * 1e3, 2)},
               # `value` is a SUM and rewards a tenant that starves the other (1 024 cluster workgroups took Qwen3-ASR from 3 415 to 319 audio-s/s with
               # the sum unchanged): each tenant's concurrent rate over its solo rate on this GPU, the smaller one first
               "tenant_slowdown": {"qwen": round(q_audio_s / world / elapsed / solo_q, 3), "streaming": round(p_audio_s / world / elapsed / solo_p, 3),
                                   "slower_tenant": round(min(q_audio_s / world / elapsed / solo_q, p_audio_s / world / elapsed / solo_p), 3)},
               "streaming_dispatch": psess.stream_stats(),
               "roofline": {"bound": "hbm", "kernel": "both launch chains are latency-bound; see the qwen and paraformer-streaming workloads", "achieved": None,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}}
        print(json.dumps(out))
    if world > 1:
        dp.barrier(device)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Whisper goldens from the REAL reference classes (build container only); see oracle/gen_golden.py."""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "automatic-speech-recognition-asr-onnx_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")

WHISPER_CASES = [
    # (fixture, config factory, ckpt seed, n_new tokens, [(audio seed, n_samples)])
    ("whisper_tiny", "whisper_tiny_test", 0, 6, [(2234, 16000), (2235, 48000), (2236, 7040)]),
    ("whisper_mid", "whisper_mid_test", 0, 5, [(2237, 128000)]),
]
FULL_CASE = ("whisper_large_v3", "whisper_large_v3", 0, 4, [(2238, 480000), (2239, 128000)])     # 30 s + 8 s at the real dimensions


PENALTY_VALUE, PENALTY_RANGE, PENALTY_STEPS = 0.8, 3, 12      # small range so the window fills inside a short golden


def reference_greedy(ref, cfg, audio, prompt, n_new, suppress, begin_suppress, penalty=None):
    """The reference's graphs driven like Inference_Whisper_ONNX.py: encoder once, prefill(prompt), decode steps.
    penalty = (value, range): the reference's APPLY_PENALTY + GREEDY_SEARCH heads with the host's rule that the multiplier is
    1.0 until `range` tokens were generated (Inference_Whisper_ONNX.py:606-632)."""
    enc, dec, model = ref["encoder"], ref["decoder"], ref["model"]
    L, H, hd, d = cfg.n_dec_layers, cfg.n_heads, cfg.d_head, cfg.d_model
    embed = model.model.decoder.embed_tokens
    pos_w = model.model.decoder.embed_positions.weight
    with torch.inference_mode():
        cross = enc(torch.from_numpy(audio).reshape(1, 1, -1))
        sk = [torch.zeros(1, H, hd, 0) for _ in range(L)]
        sv = [torch.zeros(1, H, 0, hd) for _ in range(L)]
        begin_bias = torch.zeros(cfg.vocab)
        if begin_suppress:
            begin_bias[list(begin_suppress)] = float("-inf")
        ids = torch.tensor([prompt], dtype=torch.long)
        hist, toks, all_logits = 0, [], []
        apply_penalty, greedy_search = ref["ns"]["APPLY_PENALTY"](), ref["ns"]["GREEDY_SEARCH"]()
        save_id = torch.zeros((1, 0), dtype=torch.int32)
        for step in range(n_new):
            n = ids.shape[1]
            mask = torch.triu(torch.full((1, n, hist + n), -128.0), diagonal=1)[:, :n, :hist + n] if step == 0 else torch.zeros(1, 1, hist + n)
            out = dec(*sk, *sv, *cross, embed(ids), pos_w[hist:hist + n].unsqueeze(0), mask)
            sk, sv, logits = list(out[:L]), list(out[L:2 * L]), out[-1]
            if penalty is not None:
                head = logits + (begin_bias if step == 0 else 0)
                if step > 0:
                    value = penalty[0] if save_id.shape[1] >= penalty[1] else 1.0
                    head = apply_penalty(head, save_id, torch.tensor(value, dtype=torch.float32), penalty[1])
                all_logits.append(head[0].clone())
                max_idx, save_id = greedy_search(head, save_id)
                tok = int(max_idx.reshape(-1)[0])
                toks.append(tok)
                hist += n
                ids = torch.tensor([[tok]], dtype=torch.long)
                continue
            all_logits.append(logits[0].clone())
            tok = int(torch.argmax(logits[0] + (begin_bias if step == 0 else 0)))
            toks.append(tok)
            hist += n
            ids = torch.tensor([[tok]], dtype=torch.long)
    return dict(cross=cross, logits=torch.stack(all_logits).numpy(), token_ids=np.asarray(toks, np.int32))


SAMPLING = dict(temperature=0.8, top_k=10, top_p=0.95, repetition_penalty=1.3, steps=10, seed=4242)


def reference_sampling(ref, cfg, audio, prompt, begin_suppress):
    """TOPK_TOPP_SAMPLING (Export_Whisper.py:263-308) driven like the host loop. The head draws torch.rand_like inside; the
    same uniforms are captured by re-seeding the generator and drawing the same shape first, so they can be committed."""
    enc, dec, model = ref["encoder"], ref["decoder"], ref["model"]
    head = ref["ns"]["TOPK_TOPP_SAMPLING"]()
    L, H, hd = cfg.n_dec_layers, cfg.n_heads, cfg.d_head
    embed, pos_w = model.model.decoder.embed_tokens, model.model.decoder.embed_positions.weight
    P = SAMPLING
    with torch.inference_mode():
        cross = enc(torch.from_numpy(audio).reshape(1, 1, -1))
        sk = [torch.zeros(1, H, hd, 0) for _ in range(L)]
        sv = [torch.zeros(1, H, 0, hd) for _ in range(L)]
        begin_bias = torch.zeros(cfg.vocab)
        if begin_suppress:
            begin_bias[list(begin_suppress)] = float("-inf")
        ids = torch.tensor([prompt], dtype=torch.long)
        prev = torch.zeros((1, 0), dtype=torch.long)
        hist, toks, noises, gaps = 0, [], [], []
        for step in range(P["steps"]):
            n = ids.shape[1]
            mask = torch.triu(torch.full((1, n, hist + n), -128.0), diagonal=1)[:, :n, :hist + n] if step == 0 else torch.zeros(1, 1, hist + n)
            out = dec(*sk, *sv, *cross, embed(ids), pos_w[hist:hist + n].unsqueeze(0), mask)
            sk, sv, logits = list(out[:L]), list(out[L:2 * L]), out[-1]
            logits = logits + (begin_bias if step == 0 else 0)
            torch.manual_seed(P["seed"] + step)
            noise = torch.rand((1, P["top_k"]))
            torch.manual_seed(P["seed"] + step)
            sampled, save = head(logits, torch.tensor(P["temperature"]), P["top_k"], torch.tensor(P["top_p"]),
                                 torch.tensor(P["repetition_penalty"]), prev)
            prev = save.long()
            tok = int(sampled.reshape(-1)[0])
            toks.append(tok)
            noises.append(noise[0].numpy().copy())
            hist += n
            ids = torch.tensor([[tok]], dtype=torch.long)
    return dict(token_ids=np.asarray(toks, np.int32), noise=np.stack(noises).astype(np.float32))


def gen_whisper(full=False):
    from oracle import reference_harness as rh
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    cases = [FULL_CASE] if full == "only" else list(WHISPER_CASES) + ([FULL_CASE] if full else [])
    for fixture, cfg_name, ck_seed, n_new, clips in cases:
        cfg = getattr(cfgm, cfg_name)()
        ck = ckm.synth_whisper_checkpoint(cfg, ck_seed)
        suppress = ckm.whisper_suppress_tokens(cfg)
        begin = ckm.whisper_begin_suppress_tokens(cfg)
        ref = rh.build_reference_whisper(cfg, ck, suppress_tokens=suppress)
        small = cfg.d_model <= 128
        out = {"ckpt_seed": np.int64(ck_seed), "n_cases": np.int64(len(clips)), "cfg_name": np.str_(cfg_name), "n_new": np.int64(n_new),
               "penalty_value": np.float32(PENALTY_VALUE), "penalty_range": np.int64(PENALTY_RANGE),
               "sampling_params": np.asarray([SAMPLING["temperature"], SAMPLING["top_k"], SAMPLING["top_p"], SAMPLING["repetition_penalty"]], np.float32)}
        for i, (seed, n) in enumerate(clips):
            audio = ckm.synth_audio("unit", 1, n, seed=seed)[0, 0]
            prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
            r = reference_greedy(ref, cfg, audio, prompt, n_new, suppress, begin)
            L = cfg.n_dec_layers
            keys = torch.stack([k.permute(0, 2, 1) for k in r["cross"][:L]]).numpy()     # (L, H, T, hd)
            vals = torch.stack(list(r["cross"][L:])).numpy()
            p = f"c{i}_"
            out[p + "audio_seed"], out[p + "n_samples"] = np.int64(seed), np.int64(n)
            out[p + "prompt"] = np.asarray(prompt, np.int32)
            out[p + "token_ids"] = r["token_ids"]
            srt = np.sort(r["logits"], axis=1)
            out[p + "margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
            if small:
                out[p + "cross_k"], out[p + "cross_v"], out[p + "logits"] = keys, vals, r["logits"]
            else:
                out[p + "cross_k"], out[p + "cross_v"] = keys[:, ::5, ::16].copy(), vals[:, ::5, ::16].copy()
                out[p + "logits"] = r["logits"][:, ::53].copy()
                out[p + "top1"] = srt[:, -1].astype(np.float32)
            if small:      # penalty-greedy (the reference host's default strategy) on the tiny config
                rp = reference_greedy(ref, cfg, audio, prompt, PENALTY_STEPS, suppress, begin, penalty=(PENALTY_VALUE, PENALTY_RANGE))
                out[p + "penalty_token_ids"] = rp["token_ids"]
                lg = np.where(np.isfinite(rp["logits"]), rp["logits"], -1e30)
                srt_p = np.sort(lg, axis=1)
                out[p + "penalty_margin"] = (srt_p[:, -1] - srt_p[:, -2]).astype(np.float32)
                plain = reference_greedy(ref, cfg, audio, prompt, PENALTY_STEPS, suppress, begin)["token_ids"]
                out[p + "plain_token_ids"] = plain
                print(fixture, i, "penalty tokens", rp["token_ids"], "plain", plain)
                rs = reference_sampling(ref, cfg, audio, prompt, begin)
                out[p + "sampling_token_ids"], out[p + "sampling_noise"] = rs["token_ids"], rs["noise"]
                print(fixture, i, "sampled", rs["token_ids"])
            print(fixture, i, n, "tokens", r["token_ids"], "min margin", float(out[p + "margin"].min()))
        np.savez_compressed(os.path.join(GOLDEN, fixture + ".npz"), **out)


if __name__ == "__main__":
    gen_whisper(full="only" if "--full-only" in sys.argv else "--full" in sys.argv)

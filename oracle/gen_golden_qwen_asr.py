"""Qwen3-ASR goldens from the REAL reference classes (build container only); see oracle/gen_golden.py."""
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

HEAD_IDS, TAIL_IDS, SUFFIX_IDS = [510, 511, 512], [520, 521, 512, 510, 522, 512, 530, 531], [521, 512, 510, 523, 512, 524]
CASES = [  # (fixture, config, ckpt seed, n_new, [(audio seed, n_samples, query ids, language tail ids)])
    ("qwen_asr_tiny", "qwen_asr_tiny", 0, 6, [(4401, 40000, [], []), (4402, 16000, [40, 41, 42], [77, 540]), (4403, 130000, [], [78, 540]),
                                              (4404, 7000, [], [])]),
    ("qwen_asr_mid", "qwen_asr_mid", 1, 6, [(4411, 128000, [], [77, 540]), (4412, 30000, [40, 41], []), (4413, 200000, [], [78, 540])]),
]
FULL_CASE = ("qwen_asr_0p6b", "qwen_asr_0p6b", 0, 4, [(4421, 128000, [], [77, 540]), (4422, 218080, [40, 41, 42], [])])   # the real 0.6 B geometry, sub-sampled


def reference_greedy(ref, cfg, audio, n_new, query_ids, tail_ids):
    enc, embed, rp, rd, main = ref["encoder"], ref["embed"], ref["rotary_prefill"], ref["rotary_decode"], ref["main"]
    L = cfg.n_layers
    with torch.inference_mode():
        q = embed(torch.tensor([query_ids], dtype=torch.int32).reshape(1, -1))
        base, _ = enc(torch.from_numpy(audio).reshape(1, 1, -1), q)
        tail = embed(torch.tensor([tail_ids], dtype=torch.int32).reshape(1, -1))
        concat, ids_len = ref["ns"]["CONCAT_EMBED"]()(base, tail)
        n_audio = base.shape[1] - len(HEAD_IDS) - len(query_ids) - len(SUFFIX_IDS) - len(TAIL_IDS)
        a0 = len(HEAD_IDS) + len(query_ids) + len(SUFFIX_IDS)
        audio_hidden = base[0, a0:a0 + n_audio].numpy().copy()
        keys = [torch.zeros(1, cfg.n_kv_heads, 1, cfg.d_head, 0) for _ in range(L)]
        vals = [torch.zeros(1, cfg.n_kv_heads, 1, 0, cfg.d_head) for _ in range(L)]
        cos, sin, mask, kv_len = rp(ids_len, torch.zeros(1, dtype=torch.int64))
        out = main(*keys, *vals, concat, cos, sin, mask)
        steps, toks = [out[-1][0].clone()], [int(out[-1].argmax(-1))]
        while len(toks) < n_new:
            cos, sin, kv_next = rd(kv_len)
            out = main(*out[:L], *out[L:2 * L], embed(torch.tensor([[toks[-1]]], dtype=torch.int32)), cos, sin, torch.zeros(1))
            kv_len = kv_next
            steps.append(out[-1][0].clone())
            toks.append(int(out[-1].argmax(-1)))
    return dict(audio_hidden=audio_hidden, ids_len=int(ids_len), logits=torch.stack(steps).numpy(), token_ids=np.asarray(toks, np.int32))


PENALTY = dict(value=0.8, range=3, steps=10)          # the host's REPEAT_PENALTY with a short window so that it bites within 10 steps
SAMPLING = dict(temperature=0.8, top_k=10, top_p=0.95, repetition_penalty=1.3, steps=8, seed=777)


def reference_heads(ref, cfg, audio, query_ids, tail_ids, mode):
    """The reference's head classes driven like its merged graphs (Shared_Merged.py): prefill = GREEDY_SEARCH / TOPK_TOPP_SAMPLING on the
    raw logits with an empty save_id; decode = APPLY_PENALTY (value, range passed straight through, Inference :563-575) + GREEDY_SEARCH, or
    the sampling head over every previous id. The sampling head draws torch.rand_like inside: the uniforms are captured by re-seeding."""
    enc, embed, rp, rd, main = ref["encoder"], ref["embed"], ref["rotary_prefill"], ref["rotary_decode"], ref["main"]
    ns, L = ref["ns"], cfg.n_layers
    apply_penalty, greedy_search, sampler = ns["APPLY_PENALTY"](), ns["GREEDY_SEARCH"](), ns["TOPK_TOPP_SAMPLING"]()
    steps = PENALTY["steps"] if mode == "penalty" else SAMPLING["steps"]
    with torch.inference_mode():
        q = embed(torch.tensor([query_ids], dtype=torch.int32).reshape(1, -1))
        base, _ = enc(torch.from_numpy(audio).reshape(1, 1, -1), q)
        concat, ids_len = ns["CONCAT_EMBED"]()(base, embed(torch.tensor([tail_ids], dtype=torch.int32).reshape(1, -1)))
        keys = [torch.zeros(1, cfg.n_kv_heads, 1, cfg.d_head, 0) for _ in range(L)]
        vals = [torch.zeros(1, cfg.n_kv_heads, 1, 0, cfg.d_head) for _ in range(L)]
        cos, sin, mask, kv_len = rp(ids_len, torch.zeros(1, dtype=torch.int64))
        out = main(*keys, *vals, concat, cos, sin, mask)
        save_id = torch.zeros((1, 0), dtype=torch.int32)
        toks, heads, noises = [], [], []
        for step in range(steps):
            if step:
                cos, sin, kv_next = rd(kv_len)
                out = main(*out[:L], *out[L:2 * L], embed(torch.tensor([[toks[-1]]], dtype=torch.int32)), cos, sin, torch.zeros(1))
                kv_len = kv_next
            logits = out[-1]
            if mode == "penalty":
                if step:
                    logits = apply_penalty(logits, save_id, torch.tensor(PENALTY["value"], dtype=torch.float32), PENALTY["range"])
                heads.append(logits[0].clone())
                tok, save_id = greedy_search(logits, save_id)
            else:
                P = SAMPLING
                torch.manual_seed(P["seed"] + step)
                noises.append(torch.rand((1, P["top_k"]))[0].numpy().copy())
                torch.manual_seed(P["seed"] + step)
                tok, save_id = sampler(logits, torch.tensor(P["temperature"]), P["top_k"], torch.tensor(P["top_p"]), torch.tensor(P["repetition_penalty"]),
                                       save_id.long())
                save_id = save_id.int()
            toks.append(int(tok.reshape(-1)[0]))
    r = dict(token_ids=np.asarray(toks, np.int32))
    if mode == "penalty":
        r["logits"] = torch.stack(heads).numpy()
    else:
        r["noise"] = np.stack(noises).astype(np.float32)
    return r


BEAM = dict(width=3, max_new=6)


def reference_beam(ref, cfg, audio, query_ids, tail_ids, stop_ids):
    """The build's beam search (oracle.qwen_asr_oracle.beam_search_core -- the reference has no beam code) over logits computed by the
    REFERENCE's encoder / decoder classes, one hypothesis per decoder call with its own KV tensors."""
    from oracle.qwen_asr_oracle import beam_search_core
    enc, embed, rp, rd, main_ = ref["encoder"], ref["embed"], ref["rotary_prefill"], ref["rotary_decode"], ref["main"]
    ns, L = ref["ns"], cfg.n_layers
    with torch.inference_mode():
        q = embed(torch.tensor([query_ids], dtype=torch.int32).reshape(1, -1))
        base, _ = enc(torch.from_numpy(audio).reshape(1, 1, -1), q)
        concat, ids_len = ns["CONCAT_EMBED"]()(base, embed(torch.tensor([tail_ids], dtype=torch.int32).reshape(1, -1)))
        keys = [torch.zeros(1, cfg.n_kv_heads, 1, cfg.d_head, 0) for _ in range(L)]
        vals = [torch.zeros(1, cfg.n_kv_heads, 1, 0, cfg.d_head) for _ in range(L)]
        cos, sin, mask, kv_len = rp(ids_len, torch.zeros(1, dtype=torch.int64))
        out = main_(*keys, *vals, concat, cos, sin, mask)

        def step(state, token):
            kv, kv_len_ = state
            cos_, sin_, kv_next = rd(kv_len_)
            o = main_(*kv, embed(torch.tensor([[token]], dtype=torch.int32)), cos_, sin_, torch.zeros(1))
            return o[-1][0].numpy(), (tuple(o[:2 * L]), kv_next)
        margins = []
        hyps = beam_search_core(out[-1][0].numpy(), (tuple(out[:2 * L]), kv_len), step, BEAM["width"], BEAM["max_new"], stop_ids, margins)
    toks = np.full((BEAM["width"], BEAM["max_new"]), -1, np.int32)
    for r, (t, _) in enumerate(hyps):
        toks[r, :t.size] = t
    return dict(tokens=toks, lens=np.asarray([t.size for t, _ in hyps], np.int32), scores=np.asarray([s for _, s in hyps], np.float32),
                margin=np.float32(min(margins)))


def main(full=False):
    from oracle import reference_harness as rh
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    for fixture, cfg_name, ck_seed, n_new, clips in ([FULL_CASE] if full else CASES):
        cfg = getattr(cfgm, cfg_name)()
        ck = ckm.synth_qwen_asr_checkpoint(cfg, ck_seed)
        ref = rh.build_reference_qwen_asr(cfg, ck, HEAD_IDS, TAIL_IDS, SUFFIX_IDS, max_seq_len=cfg.max_seq_len)
        out = {"ckpt_seed": np.int64(ck_seed), "n_cases": np.int64(len(clips)), "cfg_name": np.str_(cfg_name), "n_new": np.int64(n_new),
               "beam": np.asarray([BEAM["width"], BEAM["max_new"]], np.int64),
               "penalty": np.asarray([PENALTY["value"], PENALTY["range"]], np.float32),
               "sampling_params": np.asarray([SAMPLING["temperature"], SAMPLING["top_k"], SAMPLING["top_p"], SAMPLING["repetition_penalty"]], np.float32),
               "head_ids": np.asarray(HEAD_IDS, np.int32), "tail_ids": np.asarray(TAIL_IDS, np.int32), "suffix_ids": np.asarray(SUFFIX_IDS, np.int32)}
        for i, (seed, n, query, tail) in enumerate(clips):
            audio = ckm.synth_audio("unit", 1, n, seed=seed)[0, 0]
            r = reference_greedy(ref, cfg, audio, n_new, query, tail)
            p = f"c{i}_"
            out[p + "audio_seed"], out[p + "n_samples"] = np.int64(seed), np.int64(n)
            out[p + "query_ids"], out[p + "language_tail_ids"] = np.asarray(query, np.int32), np.asarray(tail, np.int32)
            out[p + "audio_hidden"], out[p + "ids_len"], out[p + "logits"], out[p + "token_ids"] = r["audio_hidden"], np.int64(r["ids_len"]), r["logits"], r["token_ids"]
            srt = np.sort(r["logits"], axis=1)
            out[p + "margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
            if full:                                      # sub-sampled: every 8th audio-embedding column, every 151st vocabulary column + the row maxima
                out[p + "audio_hidden"] = r["audio_hidden"][:, ::8].copy()
                out[p + "logits"] = r["logits"][:, ::151].copy()
                out[p + "top1"] = srt[:, -1].astype(np.float32)
            print(fixture, i, n, "audio tokens", r["audio_hidden"].shape[0], "prompt", r["ids_len"], "tokens", r["token_ids"], "min margin", float(out[p + "margin"].min()))
            if i < 2 and not full:                        # the other decode heads on two clips per fixture
                rp_ = reference_heads(ref, cfg, audio, query, tail, "penalty")
                srt = np.sort(rp_["logits"], axis=1)
                out[p + "penalty_token_ids"], out[p + "penalty_logits"] = rp_["token_ids"], rp_["logits"][:, ::7].copy()
                out[p + "penalty_margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
                rs_ = reference_heads(ref, cfg, audio, query, tail, "sampling")
                out[p + "sampling_token_ids"], out[p + "sampling_noise"] = rs_["token_ids"], rs_["noise"]
                print(fixture, i, "penalty", rp_["token_ids"], "sampling", rs_["token_ids"])
                best = None
                for tag in ("beam", "beamstop"):              # second run: the third id of the best hypothesis becomes a stop id
                    stop = [] if best is None else [int(best[2])]
                    rb = reference_beam(ref, cfg, audio, query, tail, stop)
                    best = rb["tokens"][0] if best is None else best
                    out[p + tag + "_tokens"], out[p + tag + "_lens"], out[p + tag + "_scores"], out[p + tag + "_margin"] = rb["tokens"], rb["lens"], rb["scores"], rb["margin"]
                    out[p + tag + "_stop"] = np.asarray(stop, np.int32)
                    print(fixture, i, tag, rb["tokens"].tolist(), rb["lens"], rb["scores"], "min ranking gap", float(rb["margin"]))
        np.savez_compressed(os.path.join(GOLDEN, fixture + ".npz"), **out)


if __name__ == "__main__":
    main(full="--full-only" in sys.argv)

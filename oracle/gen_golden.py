"""Mint golden vectors from the REAL reference modules (build container only).

    python -m oracle.gen_golden            # writes tests/golden/*.npz

The reference ships no tests or expected outputs, so parity is pinned by
running the reference's own `nn.Module`s (compiled in place from /root/reference by
oracle/reference_harness.py -- nothing is copied) on seeded synthetic checkpoints
(checkpoints.py) and seeded synthetic audio, and committing inputs' seeds + outputs as data.
Fixtures hold data only: seeds, shapes, and output tensors (sub-sampled at full model size).
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "automatic-speech-recognition-asr-onnx_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")

SENSEVOICE_CASES = [
    # (fixture, config factory, ckpt seed, [(audio seed, n_samples, language_idx)])
    ("sensevoice_tiny", "sensevoice_tiny", 0, [(1234, 38880, 2), (1235, 128000, 0), (1236, 400, 6), (1237, 5999, 3)]),
    ("sensevoice_small", "sensevoice_small", 0, [(1234, 128000, 2), (1238, 38880, 1)]),
]


def gen_sensevoice():
    from oracle import kaldi_mel, reference_harness as rh
    cfgm = importlib.import_module(PKG + ".config")
    ckm = importlib.import_module(PKG + ".checkpoints")
    for fixture, cfg_name, ck_seed, cases in SENSEVOICE_CASES:
        cfg = getattr(cfgm, cfg_name)()
        ck = ckm.synth_sensevoice_checkpoint(cfg, ck_seed)
        ref = rh.build_reference_sensevoice(cfg, ck, kaldi_mel.get_mel_banks)
        full = cfg.vocab <= 2000
        out = {"ckpt_seed": np.int64(ck_seed), "n_cases": np.int64(len(cases)), "cfg_name": np.str_(cfg_name)}
        for i, (seed, n, lang) in enumerate(cases):
            audio = ckm.synth_audio("kaldi", 1, n, seed=seed)[0, 0]
            st = rh.reference_sensevoice_stages(ref, audio, lang)
            p = f"c{i}_"
            out[p + "audio_seed"], out[p + "n_samples"], out[p + "lang"] = np.int64(seed), np.int64(n), np.int64(lang)
            out[p + "token_ids"], out[p + "num_id"] = st["token_ids"].astype(np.int32), st["num_id"].astype(np.int32)
            logits = st["logits"]
            srt = np.sort(logits, axis=1)
            out[p + "frame_ids"] = logits.argmax(1).astype(np.int32)
            out[p + "top1"] = srt[:, -1].astype(np.float32)
            out[p + "margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
            if full:
                for k in ("mel", "enc_in", "block0", "enc_out", "logits"):
                    out[p + k] = st[k].astype(np.float32)
            else:
                out[p + "mel"] = st["mel"][::8].astype(np.float32)
                out[p + "enc_in"] = st["enc_in"][::8].astype(np.float32)
                out[p + "block0"] = st["block0"][::8].astype(np.float32)
                out[p + "enc_out"] = st["enc_out"][::8].astype(np.float32)
                out[p + "logits_cols"] = logits[:, ::97].astype(np.float32)
            print(fixture, i, n, "tokens", st["num_id"], st["token_ids"][:8])
        np.savez_compressed(os.path.join(GOLDEN, fixture + ".npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    which = sys.argv[1:] or ["sensevoice", "whisper"]
    if "sensevoice" in which:
        gen_sensevoice()
    if "whisper" in which:
        try:
            from oracle.gen_golden_whisper import gen_whisper
        except ImportError:
            gen_whisper = None
        if gen_whisper:
            gen_whisper()

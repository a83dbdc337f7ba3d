"""Streaming Paraformer host loop -- the call surface of `Paraformer/Streaming/Inference_Paraformer_Streaming_ONNX.py`:

  pad_to_chunks()   = :343-356  the clip is extended to a whole number of chunks with white noise at the RMS of its tail
  transcribe()      = :401-449  per chunk: encoder step; when the CIF fired, decoder step -> ids -> stop ids stripped -> text piece;
                                 per-chunk RTF = step wall time / chunk duration
The 200+ cache tensors the reference re-binds between the two ONNX sessions every chunk (encoder_feedback / decoder_feedback /
encoder_decoder_bridge, :309-338) are device-resident state of `ParaformerStreamSession`, addressed by a stream id, so several
clips can advance in lock-step (`transcribe_many`).
"""
from __future__ import annotations

import time
from typing import Sequence

import numpy as np

from .paraformer import decode_tokens
from .sensevoice import prepare_audio_input


def pad_to_chunks(audio: np.ndarray, chunk: int, rng: np.random.Generator | None = None) -> np.ndarray:
    """(1, 1, L) -> (1, 1, ceil(L / chunk) * chunk), the tail filled with white noise scaled to the RMS of the samples it borders."""
    rng = rng or np.random.default_rng()
    n = audio.shape[-1]
    if n > chunk:
        windows = int(np.ceil((n - chunk) / chunk)) + 1
        pad = (windows - 1) * chunk + chunk - n
        ref = audio[:, :, -pad:].astype(np.float32) if pad else audio[:, :, :0].astype(np.float32)
    elif n < chunk:
        pad, ref = chunk - n, audio.astype(np.float32)
    else:
        pad, ref = 0, audio[:, :, :0].astype(np.float32)
    if pad == 0:
        return audio
    noise = (np.sqrt(np.mean(ref * ref)) * rng.normal(0.0, 1.0, size=(1, 1, pad))).astype(audio.dtype)
    return np.concatenate((audio, noise), axis=-1)


class ParaformerStreamTranscriber:
    def __init__(self, session, token_list: Sequence[str], stop_token_ids: Sequence[int] = (2,), decode_mode: str = "zh",
                 audio_pcm_scale: int = 1, sample_rate: int = 16000):
        self.sess, self.tokens = session, np.asarray(list(token_list), dtype=np.str_)
        self.stop, self.decode_mode = list(stop_token_ids), decode_mode
        self.audio_pcm_scale, self.sample_rate = audio_pcm_scale, sample_rate

    def transcribe_many(self, clips_int16: Sequence[np.ndarray], rng: np.random.Generator | None = None):
        """Concurrent streams, one per clip (<= session.max_streams). Returns per clip dict(text, pieces, token_ids), and stats."""
        chunk = self.sess.chunk
        prepared = [pad_to_chunks(prepare_audio_input(np.asarray(c, dtype=np.int16).reshape(1, 1, -1), "F32", audio_pcm_scale=self.audio_pcm_scale),
                                  chunk, rng)[0, 0] for c in clips_int16]
        n_chunks = [a.size // chunk for a in prepared]
        ids = list(range(len(prepared)))
        for i in ids:
            self.sess.reset(i)
        out = [dict(pieces=[], token_ids=[]) for _ in prepared]
        rtfs = []
        for k in range(max(n_chunks)):
            live = [i for i in ids if k < n_chunks[i]]
            t0 = time.time()
            fired = self.sess.step(np.stack([prepared[i][k * chunk:(k + 1) * chunk] for i in live]), live)
            rtfs.append((time.time() - t0) / (chunk / self.sample_rate))
            for i, tok in zip(live, fired):
                if tok.size:                                    # the decoder ran for this stream
                    tok = tok[~np.isin(tok, self.stop)]
                    out[i]["token_ids"].append(tok)
                    out[i]["pieces"].append(decode_tokens(self.tokens[tok].tolist(), self.decode_mode))
        for o in out:
            o["text"] = "".join(o["pieces"])
            o["token_ids"] = np.concatenate(o["token_ids"]) if o["token_ids"] else np.zeros(0, np.int32)
        return out, {"rtf_per_chunk": rtfs, "chunks": max(n_chunks)}

    def transcribe(self, audio_int16: np.ndarray, rng: np.random.Generator | None = None):
        out, stats = self.transcribe_many([audio_int16], rng)
        return out[0], stats

"""Qwen3-ASR host loop: audio in -> text out, RTF -- the call surface of `Qwen_ASR/Inference_Qwen_ASR_ONNX.py` (:424-766) on
the native session. The reference's three ORT sessions (Embed, merged prefill, merged decode) collapse into
`QwenAsrSession.prefill` / `.generate`; the prompt embeddings it feeds as `query_embed` / `language_tail_embed` tensors are token
ids here (the embedding gather runs on the device).

  special_token_ids()     = Export_Qwen_ASR.py:1500-1518  ids the exporter resolves from the tokenizer and ships as graph metadata
  supported_languages()   = :1519-1532                    language code -> name / aliases / prompt ids (= name ids + <asr_text>)
  prompt_ids()            = :1540-1586                    head = <|im_start|>system\\n ; suffix = <|im_end|>\\n<|im_start|>user\\n<|audio_start|> ;
                                                          tail = <|audio_end|><|im_end|>\\n<|im_start|>assistant\\n + "language " prefix
  resolve_language()      = resolve_supported_language (Inference :521-524): a code, a name or an alias
  parse_asr_output()      = Inference :106-123            "language X<asr_text>TEXT" -> (X, TEXT)
  transcribe()            = Inference :577-766            prompt [head | system-prompt ids | suffix | audio | tail | language tail];
                                                          generation_limit = max_seq_len - 10 - prompt length (:673); stop ids end a
                                                          sequence and are not emitted (:684-686,726-728); text = tokenizer.decode(ids,
                                                          skip_special_tokens=True); without a forced language the decoded text is
                                                          prefixed with "language " before parsing (:749-751)
Batch extension: a list of clips is one ragged batch with per-clip system prompt and language.
"""
from __future__ import annotations

import json
import time
from typing import Optional, Sequence, Tuple

import numpy as np

from .config import QwenAsrConfig
from .engine import QwenAsrSession
from .whisper import prepare_audio_input

ASR_TEXT_TAG = "<asr_text>"
LANG_PREFIX = "language "

# display name -> (code, aliases) of the languages the checkpoint family is trained on (Export_Qwen_ASR.py:1467-1498)
LANGUAGE_CODES = {
    "Chinese": ("zh", ["chinese", "mandarin", "cn", "中文"]), "English": ("en", ["english", "eng"]),
    "Cantonese": ("yue", ["cantonese", "粤语", "廣東話", "广东话"]),
    "Arabic": ("ar", ["arabic"]), "German": ("de", ["german"]), "French": ("fr", ["french"]), "Spanish": ("es", ["spanish"]),
    "Portuguese": ("pt", ["portuguese"]), "Indonesian": ("id", ["indonesian"]), "Italian": ("it", ["italian"]),
    "Korean": ("ko", ["korean", "한국어"]), "Russian": ("ru", ["russian"]), "Thai": ("th", ["thai"]), "Vietnamese": ("vi", ["vietnamese"]),
    "Japanese": ("ja", ["japanese", "日本語"]), "Turkish": ("tr", ["turkish"]), "Hindi": ("hi", ["hindi"]), "Malay": ("ms", ["malay"]),
    "Dutch": ("nl", ["dutch"]), "Swedish": ("sv", ["swedish"]), "Danish": ("da", ["danish"]), "Finnish": ("fi", ["finnish"]),
    "Polish": ("pl", ["polish"]), "Czech": ("cs", ["czech"]), "Filipino": ("fil", ["filipino", "tagalog"]), "Persian": ("fa", ["persian", "farsi"]),
    "Greek": ("el", ["greek"]), "Romanian": ("ro", ["romanian"]), "Hungarian": ("hu", ["hungarian"]), "Macedonian": ("mk", ["macedonian"]),
}


def _first_id(tokenizer, text: str) -> int:
    return int(tokenizer.encode(text, add_special_tokens=False)[0])


def special_token_ids(tokenizer) -> dict:
    vocab = tokenizer.get_vocab()
    im_end = int(vocab["<|im_end|>"])
    return {
        "stop": [int(vocab["<|endoftext|>"]), im_end],
        "asr_text": [int(vocab[ASR_TEXT_TAG])],
        "audio_start": int(vocab["<|audio_start|>"]), "audio_end": int(vocab["<|audio_end|>"]), "audio_pad": int(vocab["<|audio_pad|>"]),
        "im_start": int(vocab["<|im_start|>"]), "im_end": im_end,
        "system": _first_id(tokenizer, "system"), "user": _first_id(tokenizer, "user"), "assistant": _first_id(tokenizer, "assistant"),
        "newline": _first_id(tokenizer, "\n"),
        "language_prefix": [int(t) for t in tokenizer.encode(LANG_PREFIX, add_special_tokens=False)],
    }


def supported_languages(tokenizer, configured: Sequence[str], special: dict) -> dict:
    out = {}
    for name in configured:
        code, aliases = LANGUAGE_CODES[name]
        out[code] = {"name": name, "aliases": list(aliases),
                     "prompt_token_ids": [int(t) for t in tokenizer.encode(name, add_special_tokens=False)] + list(special["asr_text"])}
    return out


def prompt_ids(special: dict) -> Tuple[list, list, list]:
    """-> (head_ids, query_suffix_ids, tail_ids)"""
    head = [special["im_start"], special["system"], special["newline"]]
    suffix = [special["im_end"], special["newline"], special["im_start"], special["user"], special["newline"], special["audio_start"]]
    tail = [special["audio_end"], special["im_end"], special["newline"], special["im_start"], special["assistant"], special["newline"]] \
        + list(special["language_prefix"])
    return head, suffix, tail


def build_metadata(tokenizer, configured_languages: Sequence[str], cfg: QwenAsrConfig) -> dict:
    """The custom_metadata_map the exporter writes (:1821-1828): every value a string, structured ones JSON."""
    special = special_token_ids(tokenizer)
    return {"audio_pcm_scale": "32768", "max_seq_len": str(cfg.max_seq_len), "sample_rate": str(cfg.sample_rate),
            "special_token_ids": json.dumps(special), "supported_languages": json.dumps(supported_languages(tokenizer, configured_languages, special))}


def resolve_language(languages: dict, wanted: str) -> Tuple[str, dict]:
    key = str(wanted).strip().lower()
    for code, entry in languages.items():
        if key == code.lower() or key == entry["name"].lower() or key in (a.lower() for a in entry.get("aliases", ())):
            return code, entry
    raise ValueError(f"unsupported language {wanted!r}; known: {sorted(languages)}")


def parse_asr_output(raw: str, user_language: Optional[str] = None) -> Tuple[str, str]:
    text = str(raw).strip() if raw else ""
    if not text:
        return "", ""
    if user_language:
        return user_language, text
    if ASR_TEXT_TAG not in text:
        return "", text
    meta, body = text.split(ASR_TEXT_TAG, 1)
    at = meta.lower().find(LANG_PREFIX)
    language = meta[at + len(LANG_PREFIX):].strip() if at >= 0 else ""
    if language:
        language = language[:1].upper() + language[1:].lower()
    return language, body.strip()


def export_qwen_asr(cfg: QwenAsrConfig, ck: dict, path: str, metadata: dict, precision: int = 0) -> str:
    """Checkpoint (HF state-dict names) -> `.asrmodel` bundle: folded arena + config + the exporter's metadata map."""
    from .arena import build_qwen_asr_arena
    from .ort_shim import save_model
    save_model(path, "qwen_asr", cfg.to_dict(), build_qwen_asr_arena(cfg, ck, precision), dict(metadata), precision)
    return path


class QwenAsrTranscriber:
    def __init__(self, cfg: QwenAsrConfig, session: QwenAsrSession, metadata: dict, tokenizer=None, normalise_audio: bool = False,
                 repeat_penalty: float = 1.0, penalty_range: int = 10, use_sampling: bool = False, temperature: float = 0.8, top_k: int = 10,
                 top_p: float = 0.95, sampling_repetition_penalty: float = 1.0, seed: int = 0, beam_size: int = 1):
        self.cfg, self.sess, self.tokenizer = cfg, session, tokenizer
        # _resolve_strategy (:369-376): sampling wins; REPEAT_PENALTY == 1.0 selects greedy, any other value penalty-greedy (the
        # reference's defaults are 0.8 over PENALTY_RANGE = 10 ids)
        self.repeat_penalty, self.penalty_range = float(repeat_penalty), int(penalty_range)
        self.sampling = (bool(use_sampling), float(temperature), int(top_k), float(top_p), float(sampling_repetition_penalty), int(seed))
        # beam_size > 1: the README's "beam search" mode (README.md:38; no reference code -- include/asr_mi355x.h asr_qwen_beam_search);
        # it replaces the per-step head, so it excludes sampling and the repeat penalty
        self.beam_size = int(beam_size)
        if self.beam_size > 1 and (use_sampling or self.repeat_penalty != 1.0):
            raise ValueError("beam_size > 1 needs REPEAT_PENALTY = 1.0 and no sampling")
        self.audio_pcm_scale = int(metadata["audio_pcm_scale"])
        self.max_seq_len = int(metadata["max_seq_len"])
        special = metadata["special_token_ids"]
        self.special = json.loads(special) if isinstance(special, str) else special
        langs = metadata["supported_languages"]
        self.languages = json.loads(langs) if isinstance(langs, str) else langs
        self.stop = [int(t) for t in (self.special["stop"] if isinstance(self.special["stop"], list) else [self.special["stop"]])]
        self.head_ids, self.suffix_ids, self.tail_ids = prompt_ids(self.special)
        self.normalise_audio = normalise_audio

    def _query_ids(self, system_prompt) -> list:
        if not system_prompt:
            return []
        if isinstance(system_prompt, str):
            if self.tokenizer is None:
                raise ValueError("a text system prompt needs a tokenizer (pass token ids instead)")
            return [int(t) for t in self.tokenizer.encode(system_prompt, add_special_tokens=False)]
        return [int(t) for t in system_prompt]

    def transcribe(self, clips_int16: Sequence[np.ndarray], task_prompts: Sequence = ("",), language_prompts: Sequence[str] = ("",),
                   max_new: int | None = None):
        """int16 mono 16 kHz clips -> per clip dict(tokens, language, text, prompt_tokens); one prompt / language may serve all clips
        (the reference's TASK_PROMPTS / LANGUAGE_PROMPTS broadcasting, :494-509). `text` is None without a tokenizer."""
        B = len(clips_int16)
        tasks = list(task_prompts) * B if len(task_prompts) == 1 else list(task_prompts)
        langs = list(language_prompts) * B if len(language_prompts) == 1 else list(language_prompts)
        if len(tasks) != B or len(langs) != B:
            raise ValueError("task_prompts / language_prompts must have one entry or one per clip")
        audios = [prepare_audio_input(np.asarray(c, dtype=np.int16).reshape(-1), np.float32, audio_pcm_scale=self.audio_pcm_scale,
                                      normalise=self.normalise_audio)[:self.cfg.max_audio_len] for c in clips_int16]
        pre = [self.head_ids + self._query_ids(t) + self.suffix_ids for t in tasks]
        post = [self.tail_ids + (list(resolve_language(self.languages, l)[1]["prompt_token_ids"]) if l else []) for l in langs]
        t0 = time.time()
        self.sess.set_penalty(self.repeat_penalty, self.penalty_range)
        self.sess.set_sampling(*self.sampling)
        first, _, ids_len = self.sess.prefill(audios, pre, post, want_logits=False)
        limits = np.maximum(self.max_seq_len - 10 - ids_len, 0)
        if max_new is not None:
            limits = np.minimum(limits, max_new)
        if limits.max() <= 0:
            toks = [np.zeros(0, np.int32)] * B
        elif self.beam_size > 1:
            toks = [h[0][0] for h in self.sess.beam_search(self.beam_size, int(limits.max()), stop_ids=self.stop)]
        else:
            toks = self.sess.generate(int(limits.max()), stop_ids=self.stop)
        wall = time.time() - t0
        out = []
        for b in range(B):
            ids = toks[b][:limits[b]]
            text = language = None
            if self.tokenizer is not None:
                raw = self.tokenizer.decode(ids.tolist(), skip_special_tokens=True).strip()
                if not langs[b] and raw:
                    raw = LANG_PREFIX + raw
                language, text = parse_asr_output(raw)
                if langs[b]:
                    language = resolve_language(self.languages, langs[b])[1]["name"]
            out.append({"tokens": np.asarray(ids, dtype=np.int32), "language": language, "text": text, "prompt_tokens": int(ids_len[b])})
        total_s = sum(a.size for a in audios) / self.cfg.sample_rate
        return out, {"rtf": wall / total_s, "wall_s": wall}

"""Detokeniser paths of the host side, on tokenizer assets built in the test: the Whisper tail-repeat guard +
`_decode_asr`, SentencePiece decoding, the Qwen3-ASR metadata / prompt ids / output parsing. No GPU: these are the text halves of
tools/transcribe.py and of the transcribers; tests/test_transcribe_gpu.py runs the same assets through `transcribe.py run --tokenizer`."""
import importlib.util
import os

import numpy as np

from conftest import ROOT, sub
from tiny_tokenizers import qwen_tokenizer_dir, sentencepiece_model, whisper_tokenizer_dir


def _tool():
    spec = importlib.util.spec_from_file_location("transcribe_tool", os.path.join(ROOT, "tools", "transcribe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_whisper_text_is_repeat_guard_then_decode_asr(tmp_path):
    from transformers import AutoTokenizer
    cfg = sub("config").whisper_tiny_test()
    tok = AutoTokenizer.from_pretrained(whisper_tokenizer_dir(str(tmp_path / "wtok"), cfg))
    t = _tool()
    body = tok.encode(" hello there", add_special_tokens=False) + [300, 301]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    assert t.whisper_text(tok, prompt + body + [cfg.eot_id]) == " hello there tok300 tok301"          # specials dropped, byte tokens joined
    # Inference_Whisper_ONNX.py:129-139,702-715: a tail that repeats (>= 3 tokens, back to back) is cut once before decoding
    rep = [310, 311, 312, 313]
    looped = prompt + body + rep + rep
    guard = sub("whisper").remove_repeated_parts(list(looped), 3, len(looped))
    want, _ = tok._decode_asr([{"tokens": np.asarray(guard, dtype=np.int64).reshape(1, -1)}], return_timestamps=None, return_language=None, time_precision=0)
    assert t.whisper_text(tok, looped) == want and len(guard) < len(looped)
    assert t.whisper_text(tok, looped, remove_repeats=False).count("tok310") == 2
    # a timestamp token in the stream does not leak into the text
    assert "<|" not in t.whisper_text(tok, prompt[:3] + [cfg.no_timestamps_id + 1] + body + [cfg.no_timestamps_id + 3, cfg.eot_id])


def test_sentencepiece_decode_matches_piece_join(tmp_path):
    from sentencepiece import SentencePieceProcessor
    cfg = sub("config").sensevoice_tiny()
    sp = SentencePieceProcessor()
    sp.Load(sentencepiece_model(str(tmp_path / "tiny.model"), cfg.vocab))
    assert sp.GetPieceSize() == cfg.vocab
    ids = [5, 6, 7, 999, 10]
    # the transcriber's call (sensevoice.py:124 = Inference_SenseVoice_ONNX.py:305): decode([ids])[0]
    assert sp.decode([ids])[0] == "w5x6 w7x999 w10" == "".join(sp.IdToPiece(i) for i in ids).replace("▁", " ").strip()
    assert sp.encode("w10 w11x12") == [10, 11, 12]
    assert sp.decode([[2, 5, 1]])[0] == "w5"                                 # control pieces (<s>, </s>) vanish


def test_qwen_metadata_prompt_ids_and_output_parsing(tmp_path):
    from transformers import AutoTokenizer
    q = sub("qwen_asr")
    cfg = sub("config").qwen_asr_tiny()
    tok = AutoTokenizer.from_pretrained(qwen_tokenizer_dir(str(tmp_path / "qtok"), cfg.vocab))
    assert len(tok) == cfg.vocab
    sp = q.special_token_ids(tok)
    v = tok.get_vocab()
    assert sp["stop"] == [v["<|endoftext|>"], v["<|im_end|>"]] and sp["asr_text"] == [v["<asr_text>"]]
    assert tok.decode([sp["system"]]) == "system" and tok.decode([sp["assistant"]]) == "assistant" and tok.decode([sp["newline"]]) == "\n"
    assert tok.decode(sp["language_prefix"]) == q.LANG_PREFIX
    meta = q.build_metadata(tok, ["English", "Chinese"], cfg)
    langs = __import__("json").loads(meta["supported_languages"])
    assert tok.decode(langs["en"]["prompt_token_ids"]) == "English<asr_text>"
    head, suffix, tail = q.prompt_ids(sp)
    assert tok.decode(head + suffix) == "<|im_start|>system\n<|im_end|>\n<|im_start|>user\n<|audio_start|>"
    assert tok.decode(tail) == "<|audio_end|><|im_end|>\n<|im_start|>assistant\n" + q.LANG_PREFIX
    # the transcriber's text path (qwen_asr.py:187-192 = Inference_Qwen_ASR_ONNX.py:746-752) without a forced language: the model continues
    # "language " with the language name, the tag and the text; special tokens are skipped, <asr_text> is not
    gen = tok.encode(" English<asr_text> hello world", add_special_tokens=False) + [v["<|im_end|>"]]
    raw = tok.decode(gen, skip_special_tokens=True).strip()
    assert raw == "English<asr_text> hello world"
    assert q.parse_asr_output(q.LANG_PREFIX + raw) == ("English", "hello world")
    assert q.parse_asr_output("hello world", user_language="Chinese") == ("Chinese", "hello world")
    assert q.parse_asr_output("") == ("", "") and q.parse_asr_output("no tag here") == ("", "no tag here")
    assert q.resolve_language(langs, "mandarin")[0] == "zh" and q.resolve_language(langs, "EN")[0] == "en"

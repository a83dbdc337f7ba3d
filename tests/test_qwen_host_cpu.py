"""CPU: host logic of the Qwen3-ASR call surface (prompt assembly from the tokenizer, metadata map, language resolution, output
parsing) -- the parts of Inference_Qwen_ASR_ONNX.py / Export_Qwen_ASR.py that run outside the graphs."""
import json

import pytest

from conftest import sub


class FakeTokenizer:
    """Word-level stand-in with the special tokens the exporter looks up (the real tokenizer is a download)."""
    def __init__(self):
        self.vocab = {"<|endoftext|>": 1, "<|im_start|>": 2, "<|im_end|>": 3, "<|audio_start|>": 4, "<|audio_end|>": 5, "<|audio_pad|>": 6,
                      "<asr_text>": 7, "system": 10, "user": 11, "assistant": 12, "\n": 13, "language": 14, " ": 15, "English": 16, "Chinese": 17,
                      "hello": 18, "world": 19}
        self.inv = {v: k for k, v in self.vocab.items()}

    def get_vocab(self):
        return dict(self.vocab)

    def encode(self, text, add_special_tokens=False):
        if text == "\n":
            return [13]
        out = []
        for i, w in enumerate(text.split(" ")):
            if i:
                out.append(15)
            if w:
                out.append(self.vocab[w])
        return out

    def decode(self, ids, skip_special_tokens=True):
        special = {1, 2, 3, 4, 5, 6}
        return "".join(self.inv[i] for i in ids if not (skip_special_tokens and i in special))


def test_prompt_ids_and_metadata():
    q = sub("qwen_asr")
    tok = FakeTokenizer()
    sp = q.special_token_ids(tok)
    assert sp["stop"] == [1, 3] and sp["language_prefix"] == [14, 15] and sp["newline"] == 13
    head, suffix, tail = q.prompt_ids(sp)
    assert head == [2, 10, 13] and suffix == [3, 13, 2, 11, 13, 4] and tail == [5, 3, 13, 2, 12, 13, 14, 15]
    meta = q.build_metadata(tok, ["English", "Chinese"], sub("config").qwen_asr_tiny())
    assert all(isinstance(v, str) for v in meta.values()) and meta["audio_pcm_scale"] == "32768" and meta["max_seq_len"] == "512"
    langs = json.loads(meta["supported_languages"])
    assert langs["en"]["prompt_token_ids"] == [16, 7] and langs["zh"]["name"] == "Chinese"
    assert q.resolve_language(langs, "EN")[0] == "en" and q.resolve_language(langs, "mandarin")[0] == "zh" and q.resolve_language(langs, "中文")[0] == "zh"
    with pytest.raises(ValueError, match="unsupported language"):
        q.resolve_language(langs, "klingon")


@pytest.mark.parametrize("raw,lang,want", [
    ("language English<asr_text>hello world", None, ("English", "hello world")),
    ("language  chinese <asr_text> 你好 ", None, ("Chinese", "你好")),
    ("hello world", None, ("", "hello world")),
    ("hello", "English", ("English", "hello")),
    ("", None, ("", "")), ("   ", None, ("", "")),
    ("<asr_text>text only", None, ("", "text only")),
])
def test_parse_asr_output(raw, lang, want):
    assert sub("qwen_asr").parse_asr_output(raw, lang) == want

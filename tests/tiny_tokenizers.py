"""Tiny tokenizer assets built inside the tests (no tokenizer files exist offline): a hand-written SentencePiece model, a Whisper
tokenizer directory and a Qwen-style byte-level BPE tokenizer directory, each with as many ids as the synthetic model's vocabulary so
that whatever a random-weight model emits can be detokenised. They exercise the reference's detokenisation calls:
SentencePieceProcessor.decode (SenseVoice/Inference_SenseVoice_ONNX.py:305), tokenizer._decode_asr
(Whisper/Inference_Whisper_ONNX.py:702-715), tokenizer.decode(..., skip_special_tokens=True) (Qwen_ASR/Inference_Qwen_ASR_ONNX.py:746-752)."""
import os


def _bytes_to_unicode():
    """The GPT-2 byte <-> printable-character table byte-level BPE vocabularies are written in."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def sentencepiece_model(path, vocab_size):
    """A unigram SentencePiece model with `vocab_size` pieces: <unk>, <s>, </s>, then word pieces ("▁w<i>") and word-internal ones ("x<i>")."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()

    def add(piece, score, typ):
        p = m.pieces.add(); p.piece = piece; p.score = score; p.type = typ
    add("<unk>", 0.0, pb.ModelProto.SentencePiece.UNKNOWN)
    add("<s>", 0.0, pb.ModelProto.SentencePiece.CONTROL)
    add("</s>", 0.0, pb.ModelProto.SentencePiece.CONTROL)
    for i in range(3, vocab_size):
        add("▁w%d" % i if i % 3 else "x%d" % i, -0.01 * i, pb.ModelProto.SentencePiece.NORMAL)
    m.trainer_spec.model_type = pb.TrainerSpec.UNIGRAM
    m.trainer_spec.vocab_size = vocab_size
    m.normalizer_spec.name = "identity"
    m.normalizer_spec.add_dummy_prefix = True
    with open(path, "wb") as f:
        f.write(m.SerializeToString())
    return path


def whisper_tokenizer_dir(folder, cfg):
    """A WhisperTokenizer whose ids line up with a synthetic WhisperConfig: byte tokens + filler words below eot_id, then <|endoftext|>,
    <|startoftranscript|>, the language tokens, task / control tokens and timestamp tokens up to cfg.vocab."""
    from tokenizers import AddedToken
    from transformers.models.whisper.tokenization_whisper import LANGUAGES, WhisperTokenizer
    b2u = _bytes_to_unicode()
    vocab = {b2u[b]: b for b in range(256)}
    for i in range(256, cfg.eot_id):
        vocab["Ġtok%d" % i] = i
    spec = ["<|startoftranscript|>"] + ["<|%s|>" % l for l in list(LANGUAGES)[:cfg.n_languages]]
    assert cfg.sot_id == cfg.eot_id + 1 and cfg.first_language_id == cfg.sot_id + 1
    spec += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    tok = WhisperTokenizer(vocab=vocab, merges=[], additional_special_tokens=spec)
    n_ts = cfg.vocab - len(tok)
    tok.add_tokens([AddedToken("<|%.2f|>" % (0.02 * i), special=False, normalized=False) for i in range(n_ts)])
    assert len(tok) == cfg.vocab and tok.convert_tokens_to_ids("<|endoftext|>") == cfg.eot_id
    for name, want in (("<|transcribe|>", cfg.transcribe_id), ("<|translate|>", cfg.translate_id), ("<|notimestamps|>", cfg.no_timestamps_id),
                       ("<|nospeech|>", cfg.no_speech_id)):
        assert tok.convert_tokens_to_ids(name) == want, (name, tok.convert_tokens_to_ids(name), want)
    os.makedirs(folder, exist_ok=True)
    tok.save_pretrained(folder)
    return folder


QWEN_SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|audio_start|>", "<|audio_end|>", "<|audio_pad|>"]


def qwen_tokenizer_dir(folder, vocab_size):
    """A byte-level BPE tokenizer in the Qwen2 mould (PreTrainedTokenizerFast): 256 byte tokens, a few merged words, the chat / audio control
    tokens as special tokens and <asr_text> as an ordinary added token (it must survive skip_special_tokens=True: parse_asr_output splits on it)."""
    from tokenizers import AddedToken, Tokenizer, decoders, pre_tokenizers
    from tokenizers.models import BPE
    from transformers import PreTrainedTokenizerFast
    b2u = _bytes_to_unicode()
    vocab = {b2u[b]: b for b in range(256)}
    merges = []

    def word(w):                       # merge the characters of `w` left to right so that it becomes one token
        sym = [b2u[c] for c in w.encode("utf-8")]
        cur = sym[0]
        for s in sym[1:]:
            if (cur, s) not in merges:
                merges.append((cur, s))
            cur = cur + s
            if cur not in vocab:
                vocab[cur] = len(vocab)
    for w in ("system", "user", "assistant", "language", " English", " Chinese", " hello", " world"):
        word(w)
    n_added = len(QWEN_SPECIALS) + 1
    i = 0
    while len(vocab) < vocab_size - n_added:
        vocab["Ġq%d" % i] = len(vocab); i += 1
    tk = Tokenizer(BPE(vocab=vocab, merges=merges, fuse_unk=False))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tk.decoder = decoders.ByteLevel()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>", pad_token="<|endoftext|>", additional_special_tokens=QWEN_SPECIALS)
    tok.add_tokens([AddedToken("<asr_text>", special=False, normalized=False)])
    assert len(tok) == vocab_size, (len(tok), vocab_size)
    os.makedirs(folder, exist_ok=True)
    tok.save_pretrained(folder)
    return folder

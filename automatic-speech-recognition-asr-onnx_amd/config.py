"""Model geometry for the in-scope families.

Every number here is read from a checkpoint by the reference exporters; the
defaults are the public-checkpoint values quoted in SURVEY.md section 8
(SenseVoiceSmall: SenseVoice/Export_SenseVoice.py:20-32,165,178-183;
Whisper-large-v3: Whisper/Export_Whisper.py:678-683,721).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class SenseVoiceConfig:
    sample_rate: int = 16000
    n_mels: int = 80
    nfft: int = 512              # Kaldi pads the 400-sample frame to 512 before the DFT
    win_length: int = 400
    hop_length: int = 160
    pre_emphasis: float = 0.97
    lfr_m: int = 7
    lfr_n: int = 6
    d_model: int = 512
    n_heads: int = 4
    d_head: int = 128
    d_ffn: int = 2048
    n_enc0: int = 1              # encoders0: first block, 560 -> 512, no residual
    n_enc: int = 49              # encoders
    n_tp: int = 20               # tp_encoders (after after_norm)
    fsmn_kernel: int = 11
    vocab: int = 25055
    embed_rows: int = 16         # prompt embedding table rows (ids 0..15 used: 14 is the max)
    blank_id: int = 0
    use_emo: bool = True
    max_audio_len: int = 480000
    language_prompt_token_ids: tuple = (0, 3, 4, 7, 11, 12, 13)  # Export_SenseVoice.py:37-49

    @property
    def feat_dim(self) -> int:
        return self.n_mels * self.lfr_m

    @property
    def n_blocks(self) -> int:
        return self.n_enc0 + self.n_enc + self.n_tp

    @property
    def n_prompt(self) -> int:
        return 1 + (3 if self.use_emo else 2)

    def n_frames(self, audio_len: int) -> int:
        """Kaldi snip_edges framing (Export_SenseVoice.py:53)."""
        return (audio_len - self.win_length) // self.hop_length + 1

    def n_lfr(self, audio_len: int) -> int:
        return (self.n_frames(audio_len) + self.lfr_n - 1) // self.lfr_n

    def seq_len(self, audio_len: int) -> int:
        return self.n_lfr(audio_len) + self.n_prompt

    def to_dict(self):
        return asdict(self)


def sensevoice_small() -> SenseVoiceConfig:
    return SenseVoiceConfig()


def sensevoice_tiny() -> SenseVoiceConfig:
    """Reduced geometry for fast CPU goldens (same head_dim / feature width)."""
    return SenseVoiceConfig(d_model=256, n_heads=2, d_head=128, d_ffn=512,
                            n_enc0=1, n_enc=2, n_tp=1, vocab=1000)


@dataclass(frozen=True)
class WhisperConfig:
    sample_rate: int = 16000
    n_mels: int = 128
    nfft: int = 400
    hop_length: int = 160
    d_model: int = 1280
    n_heads: int = 20
    d_head: int = 64
    d_ffn: int = 5120
    n_enc_layers: int = 32
    n_dec_layers: int = 32
    vocab: int = 51866
    max_source_positions: int = 1500
    max_target_positions: int = 448
    max_audio_len: int = 480000
    # special ids of large-v3 (generation_config.json); only used by the host loop
    sot_id: int = 50258
    eot_id: int = 50257
    transcribe_id: int = 50360
    translate_id: int = 50359
    no_timestamps_id: int = 50364
    no_speech_id: int = 50363
    first_language_id: int = 50259
    n_languages: int = 100

    def n_frames(self, audio_len: int) -> int:
        return audio_len // self.hop_length

    def n_enc_pos(self, audio_len: int) -> int:
        return (self.n_frames(audio_len) + 1) // 2

    def to_dict(self):
        return asdict(self)


def whisper_large_v3() -> WhisperConfig:
    return WhisperConfig()


def whisper_tiny_test() -> WhisperConfig:
    """Reduced geometry (same head_dim=64) for fast CPU goldens."""
    return WhisperConfig(n_mels=128, d_model=128, n_heads=2, d_head=64, d_ffn=256,
                         n_enc_layers=2, n_dec_layers=2, vocab=600,
                         max_source_positions=1500, max_target_positions=64,
                         sot_id=500, eot_id=499, transcribe_id=560, translate_id=559,
                         no_timestamps_id=564, no_speech_id=563, first_language_id=501,
                         n_languages=58)


def whisper_mid_test() -> WhisperConfig:
    """Mid-size geometry: 128 mels like large-v3, 6 heads x 64, enough layers to exercise the fused cross-KV split."""
    return WhisperConfig(n_mels=128, d_model=384, n_heads=6, d_head=64, d_ffn=1536, n_enc_layers=4, n_dec_layers=4, vocab=5000,
                         max_source_positions=1500, max_target_positions=448,
                         sot_id=4900, eot_id=4899, transcribe_id=4990, translate_id=4989, no_timestamps_id=4994,
                         no_speech_id=4993, first_language_id=4901, n_languages=80)


def whisper_d256_test() -> WhisperConfig:
    """Smallest geometry the decode GEMM and the FP8 mode accept (d_model and d_ffn multiples of 256): 4 heads x 64, 2 + 3 layers."""
    return WhisperConfig(n_mels=128, d_model=256, n_heads=4, d_head=64, d_ffn=1024, n_enc_layers=2, n_dec_layers=3, vocab=3000,
                         max_source_positions=1500, max_target_positions=448,
                         sot_id=2900, eot_id=2899, transcribe_id=2990, translate_id=2989, no_timestamps_id=2994,
                         no_speech_id=2993, first_language_id=2901, n_languages=80)


@dataclass(frozen=True)
class ParaformerConfig:
    """Paraformer-large (non-streaming): Paraformer/Non-Streaming/Export_Paraformer.py:73-96,389-431."""
    sample_rate: int = 16000
    n_mels: int = 80
    nfft: int = 512
    win_length: int = 400
    hop_length: int = 160
    pre_emphasis: float = 0.97
    lfr_m: int = 7
    lfr_n: int = 6
    d_model: int = 512
    n_heads: int = 4
    d_head: int = 128
    d_ffn: int = 2048
    n_enc0: int = 1
    n_enc: int = 49
    fsmn_kernel: int = 11
    n_dec: int = 16              # decoders: FFN -> FSMN self-attention -> cross-attention
    n_dec3: int = 1              # decoders3: FFN only
    d_dec_ffn: int = 2048
    cif_kernel: int = 3
    tail_threshold: float = 0.45
    vocab: int = 8404
    max_audio_len: int = 480000

    @property
    def feat_dim(self) -> int:
        return self.n_mels * self.lfr_m

    def n_frames(self, audio_len: int) -> int:
        return (audio_len - self.win_length) // self.hop_length + 1

    def seq_len(self, audio_len: int) -> int:
        return (self.n_frames(audio_len) + self.lfr_n - 1) // self.lfr_n

    def to_dict(self):
        return asdict(self)


def paraformer_large() -> ParaformerConfig:
    return ParaformerConfig()


def paraformer_tiny() -> ParaformerConfig:
    return ParaformerConfig(d_model=256, n_heads=2, d_head=128, d_ffn=512, n_enc0=1, n_enc=2, n_dec=2, n_dec3=1, d_dec_ffn=512, vocab=500)


# =============================================================================== Qwen3-ASR
@dataclass(frozen=True)
class QwenAsrConfig:
    """Qwen3-ASR (Qwen_ASR/Export_Qwen_ASR.py): Whisper-style log-mel -> 3 x Conv2d(s2) chunk stem -> windowed-attention audio encoder
    -> Qwen3 decoder (RMSNorm, QK-norm, RoPE, GQA, SwiGLU). Defaults follow the 0.6B checkpoint's published geometry."""
    sample_rate: int = 16000
    n_mels: int = 128
    nfft: int = 400
    hop_length: int = 160
    # audio encoder
    enc_d: int = 896
    enc_heads: int = 14
    enc_ffn: int = 3584
    n_enc_layers: int = 18
    conv_channels: int = 480
    n_window: int = 50                 # chunk = 2 * n_window mel frames -> 13 tokens
    n_window_infer: int = 800          # attention window = n_window_infer / chunk chunks
    max_source_positions: int = 1500
    # text decoder
    d_model: int = 1024
    n_heads: int = 16
    n_kv_heads: int = 8
    d_head: int = 128
    d_ffn: int = 3072
    n_layers: int = 28
    vocab: int = 151936
    rms_eps: float = 1e-6
    rope_theta: float = 1000000.0
    max_seq_len: int = 1024
    max_audio_len: int = 480000

    @property
    def chunk(self) -> int:
        return 2 * self.n_window

    @property
    def chunks_per_window(self) -> int:
        return self.n_window_infer // self.chunk

    def n_frames(self, audio_len: int) -> int:
        return audio_len // self.hop_length

    def to_dict(self):
        return asdict(self)


def qwen_asr_0p6b() -> QwenAsrConfig:
    return QwenAsrConfig()


def qwen_asr_tiny() -> QwenAsrConfig:
    return QwenAsrConfig(enc_d=128, enc_heads=2, enc_ffn=256, n_enc_layers=2, conv_channels=32, n_window_infer=400, d_model=128, n_heads=2,
                         n_kv_heads=1, d_head=128, d_ffn=256, n_layers=2, vocab=600, max_seq_len=512)


def qwen_asr_mid() -> QwenAsrConfig:
    """Geometry that reaches the production kernels' paths in tests: 8-chunk attention windows, two kv heads with a GQA group of 2,
    widths the weight-streaming GEMM's fused RMSNorm accepts (d_model % 256 == 0)."""
    return QwenAsrConfig(enc_d=256, enc_heads=4, enc_ffn=512, n_enc_layers=2, conv_channels=48, n_window_infer=800, d_model=256, n_heads=4,
                         n_kv_heads=2, d_head=128, d_ffn=512, n_layers=3, vocab=1000, max_seq_len=512)

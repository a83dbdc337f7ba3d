"""Kaldi triangular mel filterbank -- restatement of a THIRD-PARTY dependency.

TEST INFRASTRUCTURE (oracle). Not imported by the product package.

The reference calls `torchaudio.compliance.kaldi.get_mel_banks(80, 512, 16000.,
20., 0., 100., -500., 1.)` at SenseVoice/Export_SenseVoice.py:159 (also
Paraformer/Non-Streaming/Export_Paraformer.py:352). torchaudio is NOT in
/root/reference, is not installed in this image and the reference pins no
version, so this follows the published algorithm (Kaldi `MelBanks::MelBanks`,
feat/mel-computations.cc, as transcribed in torchaudio's compliance module):

  mel(f) = 1127 * ln(1 + f / 700)
  num_fft_bins = padded_window / 2;  fft_bin_width = sr / padded_window
  delta = (mel(high) - mel(low)) / (num_bins + 1)
  bin b: left = mel_low + b*delta, center = left + delta, right = center + delta
  weight(b, i) = max(0, min((mel_i - left)/(center-left), (right - mel_i)/(right-center)))
  with mel_i = mel(fft_bin_width * i), i in [0, num_fft_bins)

(vtln_warp == 1.0 => no warping.)  PARITY UNPINNED against torchaudio itself:
pinned here only by analytic known-answers (tests/test_oracle_kaldi_mel.py).
Arithmetic is float32 like torchaudio's (torch default dtype).
"""
from __future__ import annotations

import math

import torch


def mel_scale_scalar(freq: float) -> float:
    return 1127.0 * math.log(1.0 + freq / 700.0)


def get_mel_banks(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float,
                  high_freq: float, vtln_low: float = 100.0, vtln_high: float = -500.0,
                  vtln_warp_factor: float = 1.0):
    assert vtln_warp_factor == 1.0, "VTLN warping is not used by the reference"
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / window_length_padded
    mel_low = mel_scale_scalar(low_freq)
    mel_high = mel_scale_scalar(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    center_freqs = 700.0 * ((center / 1127.0).exp() - 1.0)
    mel = (1127.0 * (1.0 + fft_bin_width * torch.arange(num_fft_bins) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return bins, center_freqs

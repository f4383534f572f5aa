#!/usr/bin/env python3
"""Whisper's log-mel front end from its PUBLISHED definition, as vectors (row N4's pin).

    python tests/golden/gen_golden_whisper.py        # writes whisper_logmel_r5.npz

openai-whisper is not in the image and not under /root/reference (the reference imports it: asr/asr.py).  Its front end is
small and published: whisper/audio.py `log_mel_spectrogram` is
    window = torch.hann_window(400); stft = torch.stft(audio, 400, 160, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2;  mel_spec = mel_filters(n_mels) @ magnitudes
    log_spec = clamp(mel_spec, min=1e-10).log10();  log_spec = maximum(log_spec, log_spec.max() - 8.0);  (log_spec + 4.0) / 4.0
and its `mel_filters` asset is, by that file's own docstring, `librosa.filters.mel(sr=16000, n_fft=400, n_mels=n)`: the
Slaney mel scale (linear below 1 kHz, logarithmic above: log-step ln(6.4) / 27) with Slaney area normalisation.  This script
evaluates exactly those two definitions with torch (torch.stft IS the transform whisper calls) -- it shares no code with
oracle/css_oracle.py::whisper_log_mel, with csrc/handoff.hip or with transformers.WhisperFeatureExtractor -- and writes
the filter banks (80 and 128 bands) and the features of two seeded signals.  tests/test_oracle_whisper_pin.py holds the
oracle to these vectors; the HIP kernels are held to the oracle (tests/test_hip_session.py)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    mel = f / (200.0 / 3.0)
    log_region = f >= 1000.0
    mel[log_region] = 15.0 + np.log(f[log_region] / 1000.0) / (np.log(6.4) / 27.0)
    return mel


def mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    f = m * (200.0 / 3.0)
    log_region = m >= 15.0
    f[log_region] = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m[log_region] - 15.0))
    return f


def librosa_mel(sr=16000, n_fft=400, n_mels=80):
    """librosa.filters.mel(sr, n_fft, n_mels) with its defaults: fmin 0, fmax sr / 2, htk False, norm 'slaney', float32"""
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney([0.0])[0], hz_to_mel_slaney([sr / 2.0])[0], n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, fft_f.size))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def log_mel_spectrogram(audio, n_mels):
    a = torch.from_numpy(np.asarray(audio, np.float32))
    stft = torch.stft(a, 400, 160, window=torch.hann_window(400), return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = torch.from_numpy(librosa_mel(n_mels=n_mels)) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).numpy()


def signals():
    rs = np.random.RandomState(4)
    t = np.arange(30 * 16000) / 16000.0
    chunk = (0.2 * np.sin(2 * np.pi * 440 * t) * (np.sin(2 * np.pi * 0.3 * t) > 0) + 0.02 * rs.randn(t.size)).astype(np.float32)
    clip = (0.1 * np.random.RandomState(5).randn(7 * 16000)).astype(np.float32)
    return {"chunk30": chunk, "clip7": clip}


def main():
    out = {"mel_filters_80": librosa_mel(n_mels=80), "mel_filters_128": librosa_mel(n_mels=128)}
    for name, x in signals().items():
        for n_mels in (80, 128):
            out[f"{name}_logmel_{n_mels}"] = log_mel_spectrogram(x, n_mels)[:, ::3]     # every third frame
            print(name, n_mels, out[f"{name}_logmel_{n_mels}"].shape)
    np.savez_compressed(os.path.join(HERE, "whisper_logmel_r5.npz"), **out)


if __name__ == "__main__":
    main()

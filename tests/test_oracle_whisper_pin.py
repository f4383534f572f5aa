"""Row N4's oracle (`oracle/css_oracle.py::whisper_log_mel`) pinned twice.  (1) To COMMITTED VECTORS of Whisper's published
definition (tests/golden/whisper_logmel_r5.npz from gen_golden_whisper.py: whisper/audio.py's `log_mel_spectrogram` evaluated
with torch.stft -- the call whisper itself makes -- and its `mel_filters` asset rebuilt from the librosa definition that
file's docstring names; no code shared with the oracle, the HIP kernels or transformers).  (2) To an independent
implementation the image does hold: `transformers.WhisperFeatureExtractor` (numpy).  openai-whisper itself is not in the
image and not under /root/reference (the reference imports it, asr/asr.py).  CPU only."""
import numpy as np
import pytest

import css_oracle as O

from conftest import GOLDEN

transformers = pytest.importorskip("transformers")


def _signals():
    rs = np.random.RandomState(4)
    t = np.arange(30 * 16000) / 16000.0
    chunk = (0.2 * np.sin(2 * np.pi * 440 * t) * (np.sin(2 * np.pi * 0.3 * t) > 0) + 0.02 * rs.randn(t.size)).astype(np.float32)
    clip = (0.1 * np.random.RandomState(5).randn(7 * 16000)).astype(np.float32)
    return {"chunk30": chunk, "clip7": clip}


@pytest.mark.parametrize("n_mels", [80, 128])
def test_oracle_vs_the_committed_vectors_of_the_published_definition(n_mels):
    import os
    g = np.load(os.path.join(GOLDEN, "whisper_logmel_r5.npz"))
    bank = g[f"mel_filters_{n_mels}"]
    assert bank.shape == (n_mels, 201) and np.abs(O._slaney_mel_bank(n_mels) - bank).max() < 1e-7
    for name, x in _signals().items():
        ref = g[f"{name}_logmel_{n_mels}"]
        got = O.whisper_log_mel(x, n_mels)[:, ::3]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert np.abs(got - ref).max() < 2e-4, (name, float(np.abs(got - ref).max()))


def _hf(audio, n_mels):
    fe = transformers.WhisperFeatureExtractor(feature_size=n_mels)
    return fe(audio, sampling_rate=16000, return_tensors="np")["input_features"][0]      # [n_mels, 3000]: 30 s chunk


@pytest.mark.parametrize("n_mels", [80, 128])
def test_log_mel_of_a_30_s_chunk(n_mels):
    rs = np.random.RandomState(4)
    t = np.arange(30 * 16000) / 16000.0
    audio = (0.2 * np.sin(2 * np.pi * 440 * t) * (np.sin(2 * np.pi * 0.3 * t) > 0) + 0.02 * rs.randn(t.size)).astype(np.float32)
    ref = _hf(audio, n_mels)
    got = O.whisper_log_mel(audio, n_mels)
    assert got.shape == ref.shape == (n_mels, 3000)
    assert np.abs(got - ref).max() < 2e-4, np.abs(got - ref).max()


def test_mel_filter_bank_is_whispers():
    fe = transformers.WhisperFeatureExtractor(feature_size=80)
    bank = np.asarray(fe.mel_filters, np.float64)           # [201, 80]
    assert np.abs(O._slaney_mel_bank(80).astype(np.float64) - bank.T).max() < 1e-7
    bank128 = np.asarray(transformers.WhisperFeatureExtractor(feature_size=128).mel_filters, np.float64)
    assert np.abs(O._slaney_mel_bank(128).astype(np.float64) - bank128.T).max() < 1e-7


def test_short_audio_matches_away_from_the_padding():
    """Whisper pads a short clip to 30 s with zeros BEFORE the transform; the hand-off transforms the clip itself.  The
    frames whose 400-sample window lies inside the clip agree as long as the dynamic-range clamp (max - 8) is the same,
    i.e. the clip holds the chunk's maximum -- always, the padding is silence."""
    rs = np.random.RandomState(5)
    audio = (0.1 * rs.randn(7 * 16000)).astype(np.float32)
    ref = _hf(audio, 80)
    got = O.whisper_log_mel(audio, 80)
    n = got.shape[1]
    assert n == 700
    assert np.abs(got[:, 2:n - 2] - ref[:, 2:n - 2]).max() < 2e-4

"""Synthetic multi-microphone meetings (the workloads BASELINE.json's configs name).

There is no dataset on the build container or the GPU box (the reference downloads NOTSOFAR data
from Azure, utils/azure_storage.py:109), so throughput and parity are measured on a seeded synthetic
meeting, regenerated identically on every box ("Config 2"):

* 3 sources = AR(1) noise ``lfilter([1], [1, -0.9], N(0,1)) * 0.1``, each gated by an on/off envelope
  ``sin(2 pi t / (3 + s) + 2.1 s) > -0.3``;
* every (source, mic) pair convolved with its own 32-tap decaying random FIR
  ``h = N(0,1) * exp(-n/8); h[0] += 1``;
* independent white noise (sigma 0.1) per mic.

Output layout is exactly ``css/helpers.py::load_audio``'s: float32 ``[1, n_samples, n_mics]``.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import lfilter

FS = 16000


def _fir(h: np.ndarray, x: np.ndarray) -> np.ndarray:
    """``lfilter(h, [1.0], x)``, bit for bit, 20x faster: numpy's direct convolution accumulates the taps in the same
    order as scipy's transposed direct form.  That is an observation about these library builds, not a contract, so
    the head of every call is checked against ``lfilter`` and the slow path taken on any difference (a 30-min, 7-mic
    meeting is 21 filters over 28.8 M samples: 25 s of lfilter against 1.3 s)."""
    y = np.convolve(x, h)[:x.shape[0]]
    k = min(x.shape[0], 1 << 14)
    if not np.array_equal(y[:k], lfilter(h, [1.0], x[:k])):
        return lfilter(h, [1.0], x)
    return y


def synth_meeting(seconds: float, n_mics: int = 7, seed: int = 1, fs: int = FS, n_src: int = 3,
                  noise_sigma: float = 0.1) -> np.ndarray:
    rs = np.random.RandomState(seed)
    n = int(round(seconds * fs))
    t = np.arange(n, dtype=np.float64) / fs
    mix = np.zeros((n, n_mics), dtype=np.float64)
    taps = np.arange(32, dtype=np.float64)
    for s in range(n_src):
        src = lfilter([1.0], [1.0, -0.9], rs.randn(n)) * 0.1
        env = (np.sin(2.0 * np.pi * t / (3.0 + s) + 2.1 * s) > -0.3).astype(np.float64)
        src = src * env
        for m in range(n_mics):
            h = rs.randn(32) * np.exp(-taps / 8.0)
            h[0] += 1.0
            mix[:, m] += _fir(h, src)
    mix += noise_sigma * rs.randn(n, n_mics)
    return mix.astype(np.float32)[None]

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
                  noise_sigma: float = 0.1, return_sources: bool = False):
    """the mixture float32 [1, n, n_mics]; with ``return_sources`` also the float64 images [n_src + 1, n, n_mics] it is
    the sum of (each source at every microphone, then the noise)"""
    rs = np.random.RandomState(seed)
    n = int(round(seconds * fs))
    t = np.arange(n, dtype=np.float64) / fs
    mix = np.zeros((n, n_mics), dtype=np.float64)
    images = np.zeros((n_src + 1, n, n_mics), dtype=np.float64) if return_sources else None
    taps = np.arange(32, dtype=np.float64)
    for s in range(n_src):
        src = lfilter([1.0], [1.0, -0.9], rs.randn(n)) * 0.1
        env = (np.sin(2.0 * np.pi * t / (3.0 + s) + 2.1 * s) > -0.3).astype(np.float64)
        src = src * env
        for m in range(n_mics):
            h = rs.randn(32) * np.exp(-taps / 8.0)
            h[0] += 1.0
            img = _fir(h, src)
            mix[:, m] += img
            if return_sources:
                images[s, :, m] = img
    noise = noise_sigma * rs.randn(n, n_mics)
    mix += noise
    if return_sources:
        images[n_src] = noise
        return mix.astype(np.float32)[None], images
    return mix.astype(np.float32)[None]


def synth_conversation(seconds: float, n_mics: int = 7, seed: int = 11, fs: int = FS, n_src: int = 3,
                       noise_sigma: float = 0.003, return_sources: bool = False):
    """A meeting in the regime a trained separator works in: speech-like talkers who take turns.

    ``synth_meeting``'s sources are stationary noises that are all "active" everywhere, which keeps every mask near the
    middle of its range.  Here each talker is a harmonic source (own pitch contour, own formant envelope, syllable-rate
    amplitude modulation, a little breath noise), so the talkers are spectrally sparse and an ideal ratio mask is sharp;
    and the talkers follow a seeded turn-taking script -- single talker, two- and three-talker overlap, pauses -- in
    which a talker stays silent for many seconds at a time (longer than a 3 s segment), so that whole segments see one
    or two silent speakers.  Every (talker, microphone) pair has its own 32-tap decaying random FIR, the microphones add
    independent white noise.

    Returns the mixture float32 ``[1, n, n_mics]`` (``load_audio``'s layout, css/helpers.py:40) and, with
    ``return_sources``, the images float64 ``[n_src + 1, n, n_mics]`` whose sum it is (the talkers at every microphone,
    then the noise): what an ideal-ratio-mask separator is computed from (tests/irm_separator.py)."""
    rs = np.random.RandomState(seed)
    n = int(round(seconds * fs))
    t = np.arange(n, dtype=np.float64) / fs
    # ---- the script: turns of 1.2 .. 5 s; who talks in a turn
    active = np.zeros((n_src, n), dtype=bool)
    pos, last = 0, -1
    while pos < n:
        ln = int(rs.uniform(1.2, 5.0) * fs)
        r = rs.rand()
        if r < 0.06:
            who = []
        elif r < 0.62:
            k = int(rs.randint(n_src))
            if k == last:
                k = (k + 1 + int(rs.randint(n_src - 1))) % n_src
            who = [k]
            last = k
        elif r < 0.90:
            who = list(rs.choice(n_src, 2, replace=False))
        else:
            who = list(range(n_src))
        for k in who:
            active[k, pos:pos + ln] = True
        pos += ln
    ramp = int(0.02 * fs)
    win = np.hanning(2 * ramp + 1)
    win /= win.sum()
    images = np.zeros((n_src + 1, n, n_mics), dtype=np.float64)
    taps = np.arange(32, dtype=np.float64)
    harm = np.arange(1, 41, dtype=np.float64)
    for s in range(n_src):
        f0 = (105.0, 150.0, 205.0, 260.0)[s % 4] * (1.0 + 0.08 * np.sin(2 * np.pi * (0.31 + 0.07 * s) * t + s)
                                                   + 0.03 * np.sin(2 * np.pi * (1.7 + 0.3 * s) * t))
        phase = 2.0 * np.pi * np.cumsum(f0) / fs
        formants = rs.uniform([300, 900, 2200], [800, 1800, 3200])
        src = np.zeros(n)
        for hk in harm:
            fk = hk * f0.mean()
            if fk > 0.45 * fs:
                break
            amp = sum(np.exp(-0.5 * ((fk - fc) / (120.0 + 0.08 * fc)) ** 2) for fc in formants) + 0.02
            src += amp * np.sin(hk * phase + rs.uniform(0, 2 * np.pi))
        syll = 0.55 + 0.45 * np.sin(2 * np.pi * (3.6 + 0.5 * s) * t + rs.uniform(0, 2 * np.pi))
        src = (src * syll + 0.05 * rs.randn(n)) * 0.02
        env = np.convolve(active[s].astype(np.float64), win, mode="same")
        env[env < 1e-9] = 0.0
        src = src * env
        for m in range(n_mics):
            h = rs.randn(32) * np.exp(-taps / 8.0)
            h[0] += 1.0
            images[s, :, m] = _fir(h, src)
    images[n_src] = noise_sigma * rs.randn(n, n_mics)
    mix = images.sum(axis=0).astype(np.float32)[None]
    return (mix, images) if return_sources else mix

"""An ideal-ratio-mask separator: the mask estimator of a *trained* model, simulated from known sources.

No pretrained checkpoint exists offline, and seeded weights only ever produce soft masks around 0.5 with
every speaker "active" everywhere.  The driver under test takes ANY object with the separator protocol (css/css.py:131,199:
``stft`` / ``separate`` / ``istft``), so this object supplies what a well-trained separator would: masks computed from the
talkers' own images at the reference microphone (synth.synth_conversation(..., return_sources=True)) --

* sharp and sparse (harmonic talkers), and **saturated to exactly 0 and 1** on a large fraction of the bins
  (``sharpen``: the ratio is stretched and clipped) -- exact zeros for a silent talker through whole segments;
* **exact ties between the winning masks** (mvdr_util.py:53-54 keeps every mask equal to the maximum): one bin in four
  is quantised to sixteenths, and frames where all sources are silent (the zero-padded tail of the last segment) tie at 0;
* **speaker order shuffled per segment** by a seeded rule, so that the stitching permutations (css.py:266-285) do real
  work -- including the exact cost ties of two talkers silent through a whole overlap (losses.py:43, scipy's rule);
* nothing about the activity gate is arranged: with the shipped threshold 0.3 it opens and closes with the turns.

The same class drives the REFERENCE in tests/golden/gen_golden_r4.py (numpy in, torch out, its own model's transforms) and
the HIP stages in tests/test_hip_realistic.py (device tensors), from the same float32 masks: the fixture stores their
SHA-256.  Only numpy here; torch is used through the tensors handed in."""
from __future__ import annotations

import hashlib

import numpy as np

FRAME, HOP, BINS = 512, 256, 257


def _power_stft(x: np.ndarray) -> np.ndarray:
    """|STFT|^2 of a mono float64 signal, [F, T]: 512-point periodic Hann, hop 256, no padding (feature.py:88-128)"""
    n = x.shape[0]
    t = (n - FRAME) // HOP + 1
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(FRAME) / FRAME)
    idx = np.arange(FRAME)[None, :] + HOP * np.arange(t)[:, None]
    spec = np.fft.rfft(x[idx] * w[None, :], axis=1)
    return (spec.real ** 2 + spec.imag ** 2).T


class IdealMasks:
    """The float32 masks of the whole recording, [S + 1, F, T_long] (talkers, then noise), and the per-segment view."""

    def __init__(self, images: np.ndarray, segment_frames: int = 186, hop_frames: int = 93, sharpen=(0.08, 0.92),
                 quant_every: int = 4, quant_levels: int = 16, shuffle_seed: int = 5):
        s1 = images.shape[0]
        p = np.stack([_power_stft(np.asarray(images[k, :, 0], np.float64)) for k in range(s1)])   # reference mic 0
        ratio = p / (p.sum(axis=0, keepdims=True) + 1e-30)
        lo, hi = sharpen
        m = np.clip((ratio - lo) / (hi - lo), 0.0, 1.0)
        q = np.arange(BINS) % quant_every == 1
        m[:, q, :] = np.round(m[:, q, :] * quant_levels) / quant_levels
        self.full = np.ascontiguousarray(m.astype(np.float32))
        self.S = s1 - 1
        self.T, self.hop = segment_frames, hop_frames
        self.shuffle_seed = shuffle_seed

    @property
    def frames(self) -> int:
        return self.full.shape[2]

    def order(self, i: int) -> np.ndarray:
        """the order in which call i returns the talkers"""
        return np.random.RandomState(self.shuffle_seed + 7919 * i).permutation(self.S)

    def segment(self, i: int):
        """(spk [F, T, S], noise [F, T, 1]) of segment i, zero beyond the recording (css.py:185-190 pads the spectra)"""
        st = i * self.hop
        t = max(min(self.T, self.frames - st), 0)
        m = np.zeros((self.S + 1, BINS, self.T), np.float32)
        if t > 0:
            m[:, :, :t] = self.full[:, :, st:st + t]
        spk = np.moveaxis(m[:self.S][self.order(i)], 0, 2)
        return np.ascontiguousarray(spk), np.ascontiguousarray(m[self.S:].transpose(1, 2, 0))

    def sha256(self, num_segments: int) -> str:
        h = hashlib.sha256()
        for i in range(num_segments):
            spk, noi = self.segment(i)
            h.update(spk.tobytes())
            h.update(noi.tobytes())
        return h.hexdigest()

    def statistics(self, num_segments: int) -> dict:
        """what the masks exercise (recorded in the fixture's report)"""
        sat0 = sat1 = ties = total = 0
        silent_segments = np.zeros(self.S, np.int64)
        for i in range(num_segments):
            spk, noi = self.segment(i)
            m = np.concatenate([spk, noi], axis=2)
            total += m[..., 0].size
            sat0 += int((m == 0).sum())
            sat1 += int((m == 1).sum())
            mx = m.max(axis=2, keepdims=True)
            ties += int(((m == mx).sum(axis=2) > 1).sum())
            inv = np.argsort(self.order(i))
            for k in range(self.S):
                silent_segments[k] += int(not spk[..., inv[k]].any())
        return {"tf_points": int(total), "mask_values_exactly_0": sat0, "mask_values_exactly_1": sat1,
                "tf_points_with_tied_winners": ties, "segments_with_talker_all_zero": silent_segments.tolist()}


class IdealMaskSeparator:
    """The separator protocol around ``IdealMasks``.  ``stft`` / ``istft`` are the host's transform (the reference
    model's in the generator, a torch.stft of the same definition in the GPU test); ``separate`` ignores the spectra it
    is given except for their device and counts its calls, as the reference calls it once per segment in order."""
    training = False

    def __init__(self, masks: IdealMasks, stft_fn, istft_fn=None):
        self.masks, self._stft, self._istft = masks, stft_fn, istft_fn
        self.calls = 0

    def stft(self, s):
        return self._stft(s)

    def istft(self, s):
        return self._istft(s)

    def separate(self, stft_seg):
        import torch
        spk, noi = self.masks.segment(self.calls)
        self.calls += 1
        dev = stft_seg.device
        return {"spk_masks": torch.from_numpy(spk)[None].to(dev), "noise_masks": torch.from_numpy(noi)[None].to(dev)}

    # nn.Module surface the drivers touch (css.py:141,176,178,318)
    def cpu(self):
        return self

    def to(self, device):
        return self

    def eval(self):
        return self


def unpack_bits(packed: np.ndarray, shape) -> np.ndarray:
    return np.unpackbits(packed)[:int(np.prod(shape))].reshape(tuple(int(x) for x in shape)).astype(bool)


def reproduced_samples(d: np.ndarray, perms: np.ndarray, tau: float, frames: int, n_out: int, dec: int,
                       segment_frames: int = 186, hop_frames: int = 93):
    """Which (stitched stream, decimated sample) the reference reproduces itself.

    ``d[i, k]``: distance between the reference's complex64 and complex128 evaluation of the MVDR response of segment i,
    RAW stream k (tests/golden/gen_golden_r4.py).  Stitched stream s of segment i is raw stream perms[i][s]
    (css.py:283-285).  A frame of stream s counts when every segment covering it is reproduced for that stream; a sample
    when both frames overlapping it do.  Returns (frame mask [S, frames], sample mask [S, ceil(n_out / dec)])."""
    nseg, S = d.shape
    full = np.vstack([np.arange(S)[None], np.asarray(perms).reshape(-1, S)])
    ok = np.ones((S, frames), bool)
    for i in range(nseg):
        for s in range(S):
            if not d[i, full[i, s]] <= tau:
                ok[s, i * hop_frames:min(i * hop_frames + segment_frames, frames)] = False
    q = np.arange(0, n_out, dec) // HOP
    ok_s = ok[:, np.clip(q, 0, frames - 1)] & ok[:, np.clip(q - 1, 0, frames - 1)]
    return ok, ok_s

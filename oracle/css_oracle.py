"""CPU restatement (oracle) of the NOTSOFAR CSS hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, the algorithm of the reference's continuous speech separation
front end (``css/css.py::separate_and_stitch`` and everything it drives).  It exists so that the
HIP path can be checked on the GPU box, where ``/root/reference`` does not exist.

Rules ("Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
    this module; the product package never does and fails loudly when its HIP library is missing;
  * parity is PINNED: ``tests/golden/gen_golden.py`` imports the real reference in the build
    container and writes fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
    function below against them (the reference itself ships only two inline known-answer tests on
    this path -- ``test_pit_wrapper`` and ``test_morphology`` -- both reproduced in the tests).

Every function cites the reference file:line it follows (paths relative to the reference root).
Arithmetic is float32/complex64 like the reference unless ``dtype=np.float64`` is requested
(``float64`` gives the "exact" answer both the reference and the HIP path should sit next to).
"""
from __future__ import annotations

import dataclasses
import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)  # feature.py:15  EPSILON


# ----------------------------------------------------------------------------------------------
# configuration (css/css.py:24-48)
# ----------------------------------------------------------------------------------------------
@dataclasses.dataclass
class OracleCssCfg:
    """Field-for-field restatement of the knobs of ``CssCfg`` that influence arithmetic."""
    segment_size_sec: float = 3.0
    hop_size_sec: float = 1.5
    normalize_segment_power: bool = False
    stitching_loss: str = "l1"
    stitching_input: str = "mask"
    seg_weight_m0_sec: float = 0.15
    seg_weight_m1_sec: float = 0.3
    activity_th: float = 0.4
    activity_dilation_sec: float = 0.4
    activity_erosion_sec: float = 0.2
    num_spks: int = 3
    mc_mvdr: bool = True
    mc_mask_floor_db: float = 0.0
    sc_mask_floor_db: float = -np.inf


@dataclasses.dataclass
class ModelDims:
    """Architecture of the mask estimator (conformer_wrapper.py:11-48)."""
    num_mics: int = 7
    in_features: int = 1799
    attention_dim: int = 512
    attention_heads: int = 8
    linear_units: int = 1024
    num_blocks: int = 18
    kernel_size: int = 33
    num_spks: int = 3
    num_nois: int = 1
    num_bins: int = 257
    frame_len: int = 512
    frame_hop: int = 256
    maxlen: int = 1000  # conformer.py:213


# ----------------------------------------------------------------------------------------------
# elementary functions.  The phase features have a branch cut (atan2 at +-pi) whose side is decided
# by quantities of the size of one float32 rounding error, e.g. in the purely real DC / Nyquist
# bins (see DESIGN.md "Numerical hazards").  numpy's float32 arctan2/sin/cos are not correctly
# rounded and land on the other side of the cut than ATen's (SLEEF) kernels do; evaluating in
# float64 and rounding once reproduces the correctly-rounded float32 result, which is what the
# reference's libm returns on those inputs (checked in tests/golden/gen_golden.py).
# ----------------------------------------------------------------------------------------------
def _ef(fn, dtype, *args):
    if dtype == np.float64:
        return fn(*args)
    return fn(*[np.asarray(a, dtype=np.float64) for a in args]).astype(dtype)


def _atan2(y, x, dtype):
    return _ef(np.arctan2, dtype, y, x)


def _sin(x, dtype):
    return _ef(np.sin, dtype, x)


def _cos(x, dtype):
    return _ef(np.cos, dtype, x)


def _angle(z, dtype):
    return _atan2(z.imag, z.real, dtype)


# ----------------------------------------------------------------------------------------------
# STFT / iSTFT  (feature.py:19-45, 88-128, 138-167; conformer_wrapper.py:106-146)
# ----------------------------------------------------------------------------------------------
def hann_periodic(n: int, dtype=np.float32) -> np.ndarray:
    """torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi k / n).  feature.py:29"""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def fft_points(frame_len: int, round_pow_of_two: bool = True) -> int:
    """feature.py:27: N = 2 ** ceil(log2(frame_len)) if round_pow_of_two else frame_len"""
    return 2 ** int(math.ceil(math.log2(frame_len))) if round_pow_of_two else frame_len


def _dft_angles(frame_len: int, n_fft: Optional[int] = None):
    """cos/sin of 2 pi f n / N (f <= N / 2, n < frame_len) with the argument reduced mod N in integers, so that the DC and
    Nyquist sine rows are EXACTLY zero -- as they are in the reference's kernel, which is the rfft of an identity matrix
    cut to its first frame_len rows (feature.py:40).  Exact zeros there make Im X[0] = Im X[N/2] = +0.0, which the phase
    features below depend on."""
    N = frame_len if n_fft is None else n_fft
    n = np.arange(frame_len, dtype=np.int64)
    f = np.arange(N // 2 + 1, dtype=np.int64)
    k = np.outer(f, n) % N
    ang = 2.0 * np.pi * k.astype(np.float64) / N
    c, s = np.cos(ang), np.sin(ang)
    s[(2 * k) % N == 0] = 0.0      # multiples of pi
    c[(4 * k) % N == 0] = np.round(c[(4 * k) % N == 0])  # multiples of pi/2
    return c, s


def stft_kernel(frame_len: int = 512, dtype=np.float32, window: str = "hann", frame_hop: int = 256, n_fft: Optional[int] = None) -> np.ndarray:
    """Analysis kernel K[2F, frame_len] (rows 0..F-1 = cos*w, rows F.. = -sin*w; F = N / 2 + 1, N = n_fft FFT points,
    default frame_len).  feature.py:19-45:
    window='hann' -> S = 1; window='sqrt_hann' -> W = hann ** 0.5 (on the float32 window, feature.py:29-31) and
    S = 0.5 sqrt(N N / hop) (init_kernel's `normalize` stays True: STFTBase does not pass its own, feature.py:63-66)."""
    if window not in ("hann", "sqrt_hann"):
        raise RuntimeError("Now only support sqrt hanning window or hann window")   # feature.py:24-25
    N = frame_len if n_fft is None else n_fft
    c, s = _dft_angles(frame_len, N)
    if window == "hann":
        w = hann_periodic(frame_len, np.float64)
    else:
        w = np.sqrt(hann_periodic(frame_len, np.float32)).astype(np.float64) / (0.5 * math.sqrt(N * N / frame_hop))
    k = np.concatenate([c * w, 0.0 - s * w], axis=0)
    return k.astype(dtype)


def istft_kernel(frame_len: int = 512, frame_hop: int = 256, dtype=np.float32, n_fft: Optional[int] = None) -> np.ndarray:
    """Synthesis kernel K[2F, frame_len]: sqrt-hann / S with S = 0.5*sqrt(N*N/hop) (16 as shipped).
    feature.py:30-36 (iSTFT is built without ``window`` -> 'sqrt_hann', feature.py:422-425)."""
    N = frame_len if n_fft is None else n_fft
    c, sn = _dft_angles(frame_len, N)
    w = np.sqrt(hann_periodic(frame_len, np.float64))
    s = 0.5 * math.sqrt(N * N / frame_hop)
    k = np.concatenate([c * w / s, 0.0 - sn * w / s], axis=0)
    return k.astype(dtype)


def num_frames(n_samples: int, frame_len: int = 512, frame_hop: int = 256) -> int:
    """conv1d with stride=hop and no padding (feature.py:116)."""
    if n_samples < frame_len:
        return 0
    return (n_samples - frame_len) // frame_hop + 1


def stft(x: np.ndarray, dtype=np.float32, frame_len: int = 512, frame_hop: int = 256, window: str = "hann",
         n_fft: Optional[int] = None) -> np.ndarray:
    """x [N, C] (or [N]) -> complex X [F, T, C] (or [F, T]).

    ConformerCssWrapper.stft (conformer_wrapper.py:106-129): conv1d with the Hann-DFT kernel
    (feature.py:116), magnitude/phase (feature.py:126-127), then polar() (conformer_wrapper.py:124).
    """
    squeeze = x.ndim == 1
    if squeeze:
        x = x[:, None]
    n, c = x.shape
    t = num_frames(n, frame_len, frame_hop)
    k = stft_kernel(frame_len, dtype, window, frame_hop, n_fft)
    nb = k.shape[0] // 2
    cdtype = np.complex64 if dtype == np.float32 else np.complex128
    out = np.zeros((nb, t, c), dtype=cdtype)
    xs = np.ascontiguousarray(x.T.astype(dtype))  # [C, N]
    for ch in range(c):
        frames = np.lib.stride_tricks.as_strided(
            xs[ch], shape=(t, frame_len), strides=(xs.strides[1] * frame_hop, xs.strides[1]))
        y = frames @ k.T  # [T, 2F]
        r, i = y[:, :nb], y[:, nb:]
        m = np.sqrt(r * r + i * i)
        p = _atan2(i, r, dtype)
        out[:, :, ch] = ((m * _cos(p, dtype)) + 1j * (m * _sin(p, dtype))).T.astype(cdtype)
    return out[:, :, 0] if squeeze else out


def istft(x: np.ndarray, dtype=np.float32, frame_len: int = 512, frame_hop: int = 256, n_fft: Optional[int] = None) -> np.ndarray:
    """complex X [B, F, T] -> real [B, (T-1)*hop + frame_len].

    ConformerCssWrapper.istft (conformer_wrapper.py:131-146): abs/angle, then m cos p / m sin p
    (feature.py:157-158) and conv_transpose1d with the sqrt-Hann / S kernel (feature.py:162).
    """
    b, nb, t = x.shape
    k = istft_kernel(frame_len, frame_hop, dtype, n_fft)  # [2F, L]
    m = np.abs(x).astype(dtype)
    p = _angle(x, dtype)
    r = m * _cos(p, dtype)
    i = m * _sin(p, dtype)
    c = np.concatenate([r, i], axis=1)  # [B, 2F, T]
    out = np.zeros((b, (t - 1) * frame_hop + frame_len), dtype=dtype)
    for bi in range(b):
        g = c[bi].T @ k  # [T, L]
        # overlap-add: frame t lands at [t*hop, t*hop+L); an output sample receives its frames oldest first
        for j in reversed(range((frame_len + frame_hop - 1) // frame_hop)):
            piece = g[:, j * frame_hop:min((j + 1) * frame_hop, frame_len)]     # offsets [j hop, (j+1) hop) of every frame
            w = piece.shape[1]
            for tt in range(t):
                out[bi, (tt + j) * frame_hop:(tt + j) * frame_hop + w] += piece[tt]
    return out


# ----------------------------------------------------------------------------------------------
# features (feature.py:478-508 compute_spectra, 198-249 IPDFeature, 543-569 forward)
# ----------------------------------------------------------------------------------------------
def features(stft_seg: np.ndarray, dtype=np.float32, log_spectrogram: bool = False, mvn_spectrogram: bool = True,
             ipd_mean_normalize: bool = True, ipd_mean_normalize_version: int = 1, ipd_cos: bool = False,
             pairs=None) -> np.ndarray:
    """stft_seg complex [F, T, C] (MC) or [F, T] (SC) -> feature [D, T] (D = F (1 + pairs); 1799 or 257 as shipped).

    Rows 0..F-1: clamped magnitude of channel 0 (feature.py:496-499), optionally its log (:500-501), mean / variance
    normalised over time with the unbiased std (:503-507).  Rows F + F*p + f: IPD of pair p (feature.py:212;
    default pairs (m, 0), ipd_index '1,0;...;6,0'): phase difference, time-mean removed per version 1 (:220-221:
    atan2 of the mean-removed unit phasor -- the shipped one), 2 (:222-224: minus atan2(mean sin, mean cos)) or
    3 (:225-227: minus the mean angle), as a raw angle (:245) or its cosine (:234-236).
    """
    if stft_seg.ndim == 2:
        mag = np.abs(stft_seg).astype(dtype)[None]
        pha = None
    else:
        xs = np.moveaxis(stft_seg, 2, 0)  # [C, F, T]   conformer_wrapper.py:91
        mag = np.abs(xs).astype(dtype)
        pha = _angle(xs, dtype)
    eps = dtype(EPS32)
    f = np.maximum(mag[0], eps)
    if log_spectrogram:
        f = _ef(np.log, dtype, f)
    if mvn_spectrogram:
        mean = f.mean(-1, keepdims=True, dtype=dtype)
        std = f.std(-1, keepdims=True, ddof=1, dtype=dtype)
        f = (f - mean) / (std + eps)
    feats = [f]
    if pha is not None and pha.shape[0] > 1:
        if pairs is None:
            pairs = [(m, 0) for m in range(1, pha.shape[0])]
        for l, r in pairs:
            d = pha[l] - pha[r]
            if ipd_mean_normalize:
                yr = _cos(d, dtype)
                yi = _sin(d, dtype)
                yrm = yr.mean(-1, keepdims=True, dtype=dtype)
                yim = yi.mean(-1, keepdims=True, dtype=dtype)
                if ipd_mean_normalize_version == 1:
                    d = _atan2(yi - yim, yr - yrm, dtype)
                elif ipd_mean_normalize_version == 2:
                    d = d - _atan2(yim, yrm, dtype)
                elif ipd_mean_normalize_version == 3:
                    d = d - d.mean(-1, keepdims=True, dtype=dtype)
                else:
                    raise RuntimeError("ipd_mean_normalize_version must be 1, 2 or 3")   # feature.py:228-231
            feats.append(_cos(d, dtype) if ipd_cos else d)
    return np.concatenate(feats, axis=0)


# ----------------------------------------------------------------------------------------------
# Conformer mask estimator (nnet/conformer.py)
# ----------------------------------------------------------------------------------------------
def _layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm over the last axis (biased variance)."""
    dt = x.dtype.type
    mu = x.mean(-1, keepdims=True, dtype=x.dtype)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=x.dtype)
    return xc / np.sqrt(var + dt(eps)) * w + b


def _sigmoid(x: np.ndarray) -> np.ndarray:
    dt = x.dtype.type
    return dt(1.0) / (dt(1.0) + np.exp(-x))


def _linear(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    return x @ w.T + b


class ConformerParams:
    """Thin accessor over a reference-style state dict (keys as in SURVEY.md A.4, with or without the
    leading ``module.`` of a DDP checkpoint, helpers.py:32-36)."""

    def __init__(self, state: Dict[str, np.ndarray], dtype=np.float32):
        st = {}
        for k, v in state.items():
            if k.startswith("module."):
                k = k[len("module."):]
            st[k] = np.asarray(v)
        self.st = st
        self.dtype = dtype

    def __call__(self, key: str) -> np.ndarray:
        return self.st["executor.nnet." + key].astype(self.dtype, copy=False)

    def dims(self) -> ModelDims:
        emb = self.st["executor.nnet.conformer.embed.0.weight"]
        d = emb.shape[0]
        in_features = emb.shape[1]
        blocks = 0
        while f"executor.nnet.conformer.encoders.{blocks}.layer_norm.weight" in self.st:
            blocks += 1
        dk = self.st["executor.nnet.conformer.pos_emb.pe_k.weight"].shape[1]
        ff = self.st["executor.nnet.conformer.encoders.0.feed_forward_in.net.0.weight"].shape[0]
        ks = self.st["executor.nnet.conformer.encoders.0.conv.dw_conv_1d.weight"].shape[2]
        nout = self.st["executor.nnet.linear.weight"].shape[0]
        nb = 257
        return ModelDims(num_mics=7 if in_features > nb else 1, in_features=in_features, attention_dim=d,
                         attention_heads=d // dk, linear_units=ff, num_blocks=blocks, kernel_size=ks,
                         num_spks=nout // nb - 1, num_nois=1, num_bins=nb,
                         maxlen=self.st["executor.nnet.conformer.pos_emb.pe_k.weight"].shape[0] // 2)


def conformer_forward(p: ConformerParams, feat: np.ndarray, taps: Optional[dict] = None,
                      affine_applied: bool = False) -> np.ndarray:
    """feat [D, T] -> masks [num_spks+num_nois, F, T]   (ConformerCSS.forward, conformer.py:287-310).

    Inference mode: every Dropout is the identity, BatchNorm uses running statistics.  ``affine_applied``: ``feat`` already
    carries the input bias and scale of conformer.py:298-299 (the rows the HIP feature kernel hands to the embedding).
    """
    dt = p.dtype
    dims = p.dims()
    h_, dk = dims.attention_heads, dims.attention_dim // dims.attention_heads
    x = feat.T.astype(dt)  # [T, D]                                           conformer.py:295
    if not affine_applied:
        x = (x + p("input_bias").reshape(-1)) * p("input_scale").reshape(-1)     # conformer.py:298-299
    # embed: Linear -> LayerNorm -> (Dropout) -> ReLU                           conformer.py:205-210
    x = _linear(x, p("conformer.embed.0.weight"), p("conformer.embed.0.bias"))
    x = _layer_norm(x, p("conformer.embed.1.weight"), p("conformer.embed.1.bias"))
    x = np.maximum(x, 0)
    t = x.shape[0]
    if taps is not None:
        taps["embed"] = x.copy()
    # relative positions pos_k[i, j] = pe_k[clamp(i-j) + maxlen]               conformer.py:229-233, 24-29
    pe = p("conformer.pos_emb.pe_k.weight")
    ii = np.arange(t)
    rel = np.clip(ii[:, None] - ii[None, :], -dims.maxlen, dims.maxlen - 1) + dims.maxlen  # [T, T]
    # B[h,i,j] = q[h,i] . pe[rel[i,j]] is evaluated as (q @ pe_slice^T)[h, i, rel[i,j]-lo]: the same
    # dot products as the reference's [T,T,dk] gather + batched matmul, without materialising it.
    lo, hi = int(rel.min()), int(rel.max()) + 1
    pe_slice_t = np.ascontiguousarray(pe[lo:hi].T)  # [dk, n_offsets]
    rel_idx = (rel - lo)[None]  # [1, T, T]
    half = dt(0.5)
    for l in range(dims.num_blocks):
        pre = f"conformer.encoders.{l}."

        def ff(name, x_):
            u = _layer_norm(x_, p(pre + name + ".layer_norm.weight"), p(pre + name + ".layer_norm.bias"))
            u = np.maximum(_linear(u, p(pre + name + ".net.0.weight"), p(pre + name + ".net.0.bias")), 0)
            return _linear(u, p(pre + name + ".net.3.weight"), p(pre + name + ".net.3.bias"))

        x = x + half * ff("feed_forward_in", x)                                # conformer.py:179
        # ---- self attention (conformer.py:65-92) ----
        u = _layer_norm(x, p(pre + "self_attn.layer_norm.weight"), p(pre + "self_attn.layer_norm.bias"))
        q = _linear(u, p(pre + "self_attn.linear_q.weight"), p(pre + "self_attn.linear_q.bias"))
        k = _linear(u, p(pre + "self_attn.linear_k.weight"), p(pre + "self_attn.linear_k.bias"))
        v = _linear(u, p(pre + "self_attn.linear_v.weight"), p(pre + "self_attn.linear_v.bias"))
        q = q.reshape(t, h_, dk).transpose(1, 0, 2)  # [H, T, dk]
        k = k.reshape(t, h_, dk).transpose(1, 0, 2)
        v = v.reshape(t, h_, dk).transpose(1, 0, 2)
        a = q @ k.transpose(0, 2, 1)  # [H, T, T]
        b = np.take_along_axis(q @ pe_slice_t, rel_idx, axis=2)
        scores = (a + b) / dt(math.sqrt(dk))
        scores = scores - scores.max(-1, keepdims=True)
        e = np.exp(scores)
        attn = e / e.sum(-1, keepdims=True, dtype=dt)
        ctx = (attn @ v).transpose(1, 0, 2).reshape(t, h_ * dk)
        x = x + _linear(ctx, p(pre + "self_attn.linear_out.weight"), p(pre + "self_attn.linear_out.bias"))
        # ---- conv module (conformer.py:113-127) ----
        u = _layer_norm(x, p(pre + "conv.layer_norm.weight"), p(pre + "conv.layer_norm.bias"))
        w1 = p(pre + "conv.pw_conv_1.weight").reshape(2)
        b1 = p(pre + "conv.pw_conv_1.bias").reshape(2)
        z = (u * w1[0] + b1[0]) * _sigmoid(u * w1[1] + b1[1])  # [T, D]
        wd = p(pre + "conv.dw_conv_1d.weight")[:, 0, :]  # [D, K]
        bd = p(pre + "conv.dw_conv_1d.bias")
        ksz = wd.shape[1]
        pad = (ksz - 1) // 2
        zp = np.zeros((t + 2 * pad, z.shape[1]), dtype=dt)
        zp[pad:pad + t] = z
        y = np.zeros_like(z)
        for j in range(ksz):
            y += zp[j:j + t] * wd[:, j]
        y = y + bd
        rm, rv = p(pre + "conv.BN.running_mean"), p(pre + "conv.BN.running_var")
        y = (y - rm) / np.sqrt(rv + dt(1e-5)) * p(pre + "conv.BN.weight") + p(pre + "conv.BN.bias")
        y = np.maximum(y, 0)
        w2 = p(pre + "conv.pw_conv_2.weight").reshape(())
        b2 = p(pre + "conv.pw_conv_2.bias").reshape(())
        x = x + (y * w2 + b2)
        x = x + half * ff("feed_forward_out", x)                               # conformer.py:182
        x = _layer_norm(x, p(pre + "layer_norm.weight"), p(pre + "layer_norm.bias"))  # conformer.py:184
        if taps is not None:
            taps[f"block{l}"] = x.copy()
    m = _sigmoid(_linear(x, p("linear.weight"), p("linear.bias")))           # conformer.py:302-304
    nmask = dims.num_spks + dims.num_nois
    return np.ascontiguousarray(m.T.reshape(nmask, dims.num_bins, t))          # chunk(4) conformer.py:307-310


def separate(p: ConformerParams, stft_seg: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """One segment: complex [F, T, C] / [F, T] -> (spk_masks [F, T, S], noise_masks [F, T, 1]).
    ConformerCssWrapper.separate (conformer_wrapper.py:79-104)."""
    feat = features(stft_seg, p.dtype)
    m = conformer_forward(p, feat)
    s = p.dims().num_spks
    return np.moveaxis(m[:s], 0, 2), np.moveaxis(m[s:], 0, 2)


# ----------------------------------------------------------------------------------------------
# WTA -> SCM -> MVDR  (utils/mvdr_util.py)
# ----------------------------------------------------------------------------------------------
def make_wta(spk_masks: np.ndarray, noise_masks: np.ndarray, wta_index: Optional[np.ndarray] = None) -> np.ndarray:
    """[S, F, T], [Nn, F, T] -> winner-take-all masks [S+1, F, T] (mvdr_util.py:50-55).

    ``wta_index`` [F, T] (test hook) replaces the ``mask == max`` decision by a
    given winner map, so that the chain downstream can be compared on identical decisions."""
    noise = noise_masks.sum(axis=0, keepdims=True)
    mask = np.vstack([spk_masks, noise])
    if wta_index is not None:
        win = np.arange(mask.shape[0])[:, None, None] == wta_index[None]
    else:
        win = mask == mask.max(axis=0, keepdims=True)
    return np.where(win, mask, 1e-10)  # float64 result, exactly like np.where in the reference


def mask_scm(mix: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """mix [C, F, T] complex, mask [F, T] -> [F, C, C] (mvdr_util.py:58-66)."""
    c = mix.shape[0]
    xt = mix.transpose(1, 2, 0)
    r = np.einsum("FT,FTM,FTm->FMm", mask, xt, xt.conj())
    r += 1e-15 * np.eye(c)[None]
    return r


def bf_coeffs(noi_scm: np.ndarray, tgt_scm: np.ndarray) -> np.ndarray:
    """Souden MVDR, reference mic 0 (mvdr_util.py:69-75).  Note ``den[0] += 1e-15`` touches bin 0 only."""
    num = np.linalg.solve(noi_scm, tgt_scm)
    den = np.trace(num, axis1=-2, axis2=-1)[..., None, None]
    den[0] += 1e-15
    return (num / den)[..., 0]


def apply_bf(mix: np.ndarray, w: np.ndarray) -> np.ndarray:
    """y[f, t] = sum_c conj(W[f, c]) x[c, f, t]  (mvdr_util.py:78-80)."""
    c, f, t = mix.shape
    return np.sum(w.reshape(f, c, 1).conj() * mix.transpose(1, 0, 2), axis=1)


def make_mvdr(spk_masks: np.ndarray, noise_masks: np.ndarray, mix_stft: np.ndarray,
              cplx=None, taps: Optional[dict] = None, wta_index: Optional[np.ndarray] = None) -> List[np.ndarray]:
    """spk [S, F, T], noise [Nn, F, T], mix [C, F, T] complex -> S x [F, T] complex (mvdr_util.py:5-47).

    ``cplx=np.complex128`` evaluates the same chain in double precision (the reference's complex64
    result is ~2e-5 away from it on well-conditioned inputs)."""
    if cplx is not None:
        mix_stft = mix_stft.astype(cplx)
    wta = make_wta(spk_masks, noise_masks, wta_index)
    scms = [mask_scm(mix_stft, m) for m in wta]
    spk_scms = np.stack(scms[:-1])
    noise_scm = scms[-1]
    out = []
    ws = []
    ns = spk_scms.shape[0]
    for i in range(ns):
        other = spk_scms[np.arange(ns) != i].sum(axis=0)
        w = bf_coeffs(noise_scm + other, spk_scms[i])
        ws.append(w)
        out.append(apply_bf(mix_stft, w))
    if taps is not None:
        taps["wta"] = wta
        taps["scm"] = np.stack(scms)
        taps["w"] = np.stack(ws)
    return out


# ----------------------------------------------------------------------------------------------
# stitching helpers (css/css.py:341-390, training/losses.py, utils/numpy_utils.py)
# ----------------------------------------------------------------------------------------------
def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """torch.linspace in float32 as ATen's vectorised CPU kernel evaluates it: float32 step, symmetric
    evaluation (start + i*step for the first half, end - (steps-1-i)*step for the second), each with a
    fused multiply-add (emulated here in float64: the product of two float32 is exact in float64)."""
    start, end = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((end - start) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = np.float32(np.float64(start) + np.float64(step) * i)
        else:
            out[i] = np.float32(np.float64(end) - np.float64(step) * (steps - i - 1))
    return out


def calc_segment_weight(seg_frames: int, m0: int, m1: int, is_first_seg=False, is_last_seg=False) -> np.ndarray:
    """Trapezoid overlap-add weight (css.py:341-390)."""
    assert seg_frames > 2 * m1, "not enough frames to fit weighting window"
    w = np.ones(seg_frames, dtype=np.float32)
    w[:m0] = 0
    w[seg_frames - m0:] = 0
    lin = linspace_f32(0.1, 1.0, m1 - m0)
    w[m0:m1] = lin
    w[seg_frames - m1:seg_frames - m0] = lin[::-1]
    if is_first_seg:
        w[:m0] = 0.1
    if is_last_seg:
        w[seg_frames - m0:] = 0.1
    return w


def lsap(cost: np.ndarray) -> Tuple[int, ...]:
    """``scipy.optimize.linear_sum_assignment`` for a square matrix, restated (the reference's permutation solver,
    losses.py:43; scipy pinned 1.11.4, requirements.txt): the shortest-augmenting-path algorithm of scipy's
    ``rectangular_lsap`` (Crouse 2016), INCLUDING its tie rules -- columns are scanned in descending order, among equally
    cheap columns an unassigned one wins, later ones win otherwise.  The optimum is what a brute-force search finds; the
    rules decide WHICH optimum on exact ties, and exact ties are the normal case where two speakers are silent through a
    whole overlap (all four costs between them are 0).  Returns col4row: row a is assigned column col4row[a].
    Pinned against scipy itself on tie-laden matrices in tests/test_oracle_lsap.py."""
    c = np.asarray(cost, dtype=np.float64)
    n = c.shape[0]
    assert c.shape == (n, n)
    inf = float("inf")
    u, v = [0.0] * n, [0.0] * n
    path = [-1] * n
    col4row, row4col = [-1] * n, [-1] * n
    for cur in range(n):
        remaining = [n - it - 1 for it in range(n)]
        num_remaining = n
        sr, sc = [False] * n, [False] * n
        spc = [inf] * n
        min_val, i, sink = 0.0, cur, -1
        while sink == -1:
            index, lowest = -1, inf
            sr[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + c[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            assert min_val != inf, "infeasible cost matrix"
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            sc[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for i in range(n):
            if sr[i] and i != cur:
                u[i] += min_val - spc[col4row[i]]
        for j in range(n):
            if sc[j]:
                v[j] -= min_val - spc[j]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    return tuple(col4row)


def pit_perm(pred: np.ndarray, target: np.ndarray, loss: str = "l1") -> Tuple[float, Tuple[int, ...], np.ndarray]:
    """pred, target [..., S] -> (min loss, target_perm, cost[S, S]).

    cost[a, b] = mean |pred[..., a] - target[..., b]|  (losses.py:50-71); the reference minimises with
    scipy's linear_sum_assignment (losses.py:43), restated in ``lsap`` with its tie rules."""
    s = pred.shape[-1]
    cost = np.zeros((s, s), dtype=np.float64)
    for a in range(s):
        for b in range(s):
            d = pred[..., a].astype(np.float64) - target[..., b].astype(np.float64)
            cost[a, b] = np.mean(np.abs(d)) if loss == "l1" else np.mean(d * d)
    perm = lsap(cost)
    best = sum(cost[a, perm[a]] for a in range(s)) / s
    return float(best), tuple(perm), cost


def dilate(arr: np.ndarray, iters: int) -> np.ndarray:
    """numpy_utils.py:10-13 (zero padding, window 2*iters+1, max)."""
    n = arr.shape[0]
    pad = np.concatenate([np.zeros(iters, arr.dtype), arr, np.zeros(iters, arr.dtype)])
    return np.lib.stride_tricks.sliding_window_view(pad, 2 * iters + 1).max(1) if n else arr


def erode(arr: np.ndarray, iters: int) -> np.ndarray:
    """numpy_utils.py:4-7 (one padding, window 2*iters+1, min)."""
    n = arr.shape[0]
    pad = np.concatenate([np.ones(iters, arr.dtype), arr, np.ones(iters, arr.dtype)])
    return np.lib.stride_tricks.sliding_window_view(pad, 2 * iters + 1).min(1) if n else arr


@dataclasses.dataclass
class SegmentPlan:
    """Index arithmetic of css.py:144-171, reproduced with the same float expressions."""
    segment_frames: int
    hop_frames: int
    m0_frames: int
    m1_frames: int
    dilation_frames: int
    erosion_frames: int
    mix_frames: int       # T_long after the optional padding (css.py:159-164)
    stft_frames: int      # T_long before padding
    num_segments: int

    @property
    def overlap_frames(self) -> int:
        return self.segment_frames - self.hop_frames

    def seg_range(self, i: int) -> Tuple[int, int, int]:
        """(st, en, t_valid) of segment i (css.py:183-193; note ``>=``)."""
        st = i * self.hop_frames
        en = st + self.segment_frames
        if en >= self.mix_frames:
            en = self.mix_frames
        return st, en, en - st


def make_plan(n_samples: int, fs: int, cfg, frame_len: int = 512, frame_hop: int = 256) -> SegmentPlan:
    seg_frames = num_frames(int(cfg.segment_size_sec * fs), frame_len, frame_hop)       # css.py:145-147
    hop = int(seg_frames * cfg.hop_size_sec / cfg.segment_size_sec)                      # css.py:148
    m0 = int(seg_frames * cfg.seg_weight_m0_sec / cfg.segment_size_sec)
    m1 = int(seg_frames * cfg.seg_weight_m1_sec / cfg.segment_size_sec)
    dil = int(seg_frames * cfg.activity_dilation_sec / cfg.segment_size_sec)
    ero = int(seg_frames * cfg.activity_erosion_sec / cfg.segment_size_sec)
    t_stft = num_frames(n_samples, frame_len, frame_hop)
    t_long = max(t_stft, seg_frames)                                                       # css.py:159-164
    overlap = seg_frames - hop
    nseg = int(np.ceil((t_long - overlap) / hop))                                          # css.py:166-169
    return SegmentPlan(seg_frames, hop, m0, m1, dil, ero, t_long, t_stft, nseg)


# ----------------------------------------------------------------------------------------------
# the driver (css/css.py:110-338)
# ----------------------------------------------------------------------------------------------
def separate_and_stitch(speech_mix: np.ndarray, params: ConformerParams, fs: int, cfg,
                        mvdr_cplx=None, separate_fn=None, taps: Optional[dict] = None,
                        wta_override: Optional[np.ndarray] = None, frame: Optional[dict] = None):
    """speech_mix [1, N, C] float32 -> (list of S float32 [N_out], side_info).

    ``separate_fn(i, stft_seg[F, T, C]) -> (spk [F,T,S], noise [F,T,1])`` overrides the mask estimator
    (used by tests to inject e.g. rotated speaker orders); ``wta_override``
    [num_segments, F, T] injects winner-take-all decisions (see ``make_wta``).  ``frame``: the extractor's geometry,
    dict(frame_len=, frame_hop=, n_fft=, window=) (ExtractorCfg, feature.py:19-45); default 512 / 256 / 512 / 'hann'."""
    fr = dict(frame_len=512, frame_hop=256, n_fft=None, window="hann")
    fr.update(frame or {})
    assert speech_mix.ndim == 3, f"expecting 3 dimensions, got {speech_mix.shape}"          # css.py:139
    assert speech_mix.shape[0] == 1, "assuming 1 example in batch"                          # css.py:196
    dt = params.dtype if params is not None else np.float32
    x = speech_mix[0]
    n, c = x.shape
    plan = make_plan(n, fs, cfg, fr["frame_len"], fr["frame_hop"])
    tseg, hop = plan.segment_frames, plan.hop_frames
    stft_mix = stft(x, dt, fr["frame_len"], fr["frame_hop"], fr["window"], fr["n_fft"])  # [F, T_long, C]                                               css.py:155
    nb = stft_mix.shape[0]
    if stft_mix.shape[1] < tseg:                                                            # css.py:159-164
        pad = np.zeros((nb, tseg - stft_mix.shape[1], c), dtype=stft_mix.dtype)
        stft_mix = np.concatenate([stft_mix, pad], axis=1)
    t_long = stft_mix.shape[1]
    s = cfg.num_spks
    sep_list, mask_list, t_valid = [], [], []
    for i in range(plan.num_segments):                                                      # css.py:182-250
        st, en, t = plan.seg_range(i)
        seg = np.zeros((nb, tseg, c), dtype=stft_mix.dtype)
        seg[:, :t] = stft_mix[:, st:en]
        seg_in = seg if c > 1 else seg[:, :, 0]
        if separate_fn is not None:
            spk, noi = separate_fn(i, seg_in)
        else:
            spk, noi = separate(params, seg_in)
        assert spk.shape == (nb, tseg, s)                                                   # css.py:202-203
        ref = seg[:, :, 0]
        if c > 1 and cfg.mc_mvdr:                                                           # css.py:211-218
            mv = make_mvdr(np.moveaxis(spk, 2, 0), np.moveaxis(noi, 2, 0), np.moveaxis(seg, 2, 0),
                           cplx=mvdr_cplx, taps=(taps.setdefault(f"mvdr{i}", {}) if taps is not None else None),
                           wta_index=(wta_override[i] if wta_override is not None else None))
            seg_for_masking = np.stack(mv, axis=-1).astype(stft_mix.dtype)
            floor_db = cfg.mc_mask_floor_db
        else:
            seg_for_masking = ref[:, :, None]
            floor_db = cfg.mc_mask_floor_db if c > 1 else cfg.sc_mask_floor_db             # css.py:223
        assert floor_db <= 0                                                                # css.py:224
        floor = 10.0 ** (floor_db / 20.0)
        clipped = np.maximum(spk, np.asarray(floor, dtype=spk.dtype))                       # css.py:226
        sep = seg_for_masking * clipped                                                     # css.py:227
        if cfg.normalize_segment_power:                                                     # css.py:233-247
            mix_e = np.sqrt(np.mean(np.abs(ref[:, :t]) ** 2))
            sep_e = np.sqrt(np.mean(np.abs(sep[:, :t].sum(-1)) ** 2))
            sep = (mix_e / sep_e) * sep
        sep_list.append(sep.astype(stft_mix.dtype))
        mask_list.append(spk.astype(dt))
        t_valid.append(t)
    # ---- stitch (css.py:254-299) ----
    cd = stft_mix.dtype
    stft_st = np.zeros((nb, t_long, s), dtype=cd)
    mask_st = np.zeros((nb, t_long, s), dtype=np.float32)
    wg_st = np.zeros(t_long, dtype=np.float32)
    wg = calc_segment_weight(tseg, plan.m0_frames, plan.m1_frames, is_first_seg=True)
    wg_st[:tseg] += wg
    stft_st[:, :tseg] += wg[None, :, None] * sep_list[0]
    mask_st[:, :tseg] += wg[None, :, None] * mask_list[0]
    perms, costs = [tuple(range(s))], []
    ov = plan.overlap_frames
    for i in range(1, plan.num_segments):
        if cfg.stitching_input == "mask":
            left, right = mask_list[i - 1], mask_list[i]
        elif cfg.stitching_input == "separation_result":
            left, right = np.abs(sep_list[i - 1]), np.abs(sep_list[i])
        else:
            raise AssertionError(f"unexpected stitching_input: {cfg.stitching_input}")
        _, perm, cost = pit_perm(left[:, -ov:], right[:, :ov], cfg.stitching_loss)        # css.py:276
        perms.append(perm)
        costs.append(cost)
        mask_list[i] = mask_list[i][..., list(perm)]                                        # css.py:283-285
        sep_list[i] = sep_list[i][..., list(perm)]
        st = i * hop
        en = min(st + tseg, t_long)
        wg = calc_segment_weight(tseg, plan.m0_frames, plan.m1_frames,
                                 is_last_seg=(i == plan.num_segments - 1))[:en - st]
        wg_st[st:en] += wg
        stft_st[:, st:en] += wg[None, :, None] * sep_list[i][:, :en - st]
        mask_st[:, st:en] += wg[None, :, None] * mask_list[i][:, :en - st]
    assert (wg_st > 1e-5).all(), "zero weights found. check hop_size, segment_size or m0, m1"  # css.py:297
    stft_st /= wg_st[None, :, None]
    mask_st /= wg_st[None, :, None]
    # ---- activity gating (css.py:301-312) ----
    activity = mask_st.mean(axis=0, dtype=np.float32)  # [T, S]
    activity_b = activity >= np.float32(cfg.activity_th)
    act_final = np.stack([erode(dilate(activity_b[:, k], plan.dilation_frames), plan.erosion_frames)
                          for k in range(s)], axis=1)
    stft_st = stft_st * act_final[None].astype(np.float32)
    wavs = istft(np.ascontiguousarray(np.moveaxis(stft_st, 2, 0)), dt, fr["frame_len"], fr["frame_hop"], fr["n_fft"])   # css.py:316-319
    side = {
        "mask_stitched": mask_st[None],            # [1, F, T_long, S]
        "activity_b": activity_b,                  # [T_long, S]
        "activity_final": act_final[None],         # [1, T_long, S]
        "segment_frames": tseg,
        # extras (not in the reference's dict): decisions the parity tests compare exactly
        "perms": perms, "pit_costs": costs, "activity": activity, "plan": plan,
    }
    if taps is not None:
        taps["stft_stitched"] = stft_st
    return [wavs[k].astype(np.float32) for k in range(s)], side


# ----------------------------------------------------------------------------------------------
# validation loss of the training loop (css/training/train.py:411-470 _calc_loss, :529 eval_model)
# ----------------------------------------------------------------------------------------------
def validation_loss(params: ConformerParams, mix: np.ndarray, gt_spk0: np.ndarray, gt_noise0: np.ndarray,
                    loss_name: str = "masked_mag", base_loss: str = "mse", clip_gt_to_mixture: bool = False,
                    noise_weight: float = 1.0):
    """mix [B, n, M]; gt_spk0 [B, S, n] and gt_noise0 [B, n]: the ground truths at the reference microphone
    (train.py:421-425 takes mic 0 of the batch's ground-truth tensors).  -> (loss, spk_loss [B], noise_loss [B], perms).

    train.py:413-420  forward and |STFT| of the mixture's mic 0; :427-438 optional clipping of the targets to the mixture;
    :449-462 'masked_mag' (mask * |mix| against |gt|) or :464-476 'mask' (mask against |gt| / (|mix| + eps));
    PIT over the speaker outputs (losses.py:50-97), plain loss for the noise output; :481 the weighted mean."""
    eps = np.float32(np.finfo(np.float32).eps)
    b_, n, m = mix.shape
    s = gt_spk0.shape[1]
    spk_loss, noise_loss, perms = [], [], []
    for b in range(b_):
        x = stft(mix[b])                                                       # [F, T, M]
        masks = conformer_forward(params, features(x if m > 1 else x[:, :, 0]))   # [S + 1, F, T]
        mixmag = np.abs(x[:, :, 0]).astype(np.float32)                          # [F, T]
        gt = np.stack([np.abs(stft(gt_spk0[b, k][:, None])[:, :, 0]) for k in range(s)], axis=-1).astype(np.float32)   # [F, T, S]
        gn = np.abs(stft(gt_noise0[b][:, None])[:, :, 0]).astype(np.float32)
        if clip_gt_to_mixture:
            gt = np.minimum(gt, mixmag[..., None])
            gn = np.minimum(gn, mixmag)
        pred = np.moveaxis(masks[:s], 0, 2)                                     # [F, T, S]
        pn = masks[s:].sum(axis=0) if masks.shape[0] - s > 1 else masks[s]
        if loss_name == "masked_mag":
            pred, pn = pred * mixmag[..., None], pn * mixmag
            tgt, tn = gt, gn
        elif loss_name == "mask":
            tgt, tn = gt / (mixmag[..., None] + eps), gn / (mixmag + eps)
        else:
            raise ValueError(f"Unknown loss name: {loss_name}!")
        best, perm, _ = pit_perm(pred, tgt, "l1" if base_loss == "l1" else "mse")
        d = pn.astype(np.float64) - tn.astype(np.float64)
        spk_loss.append(best)
        noise_loss.append(float(np.mean(np.abs(d)) if base_loss == "l1" else np.mean(d * d)))
        perms.append(perm)
    spk_loss, noise_loss = np.array(spk_loss), np.array(noise_loss)
    return float(np.mean(spk_loss + noise_weight * noise_loss)), spk_loss, noise_loss, perms


# ----------------------------------------------------------------------------------------------
# downstream hand-off: active regions + Whisper's log-mel front end.
# whisper (openai-whisper, requirements.txt of the reference; asr/asr.py:58,73-74 calls it) is NOT under the reference
# tree and not installed here: its published algorithm (whisper/audio.py log_mel_spectrogram; mel bank =
# librosa.filters.mel(sr=16000, n_fft=400, n_mels), slaney scale / normalisation) is restated.  Pinned to a second
# party's implementation of the same front end that the image holds (transformers.WhisperFeatureExtractor:
# tests/test_oracle_whisper_pin.py), NOT to openai-whisper itself.
# ----------------------------------------------------------------------------------------------
def active_regions(act_final: np.ndarray, pad_frames: int, n_out: int, frame_hop: int = 256, frame_len: int = 512):
    """act_final [T_long] bool (css.py:303-312) -> [n, 2] sample ranges: maximal runs of active frames, widened by
    pad_frames on both sides, overlapping / touching runs merged; frame t spans samples [t hop, t hop + frame_len)."""
    t, tl, out = 0, len(act_final), []
    while t < tl:
        if not act_final[t]:
            t += 1
            continue
        e = t
        while e < tl and act_final[e]:
            e += 1
        a = max(t - pad_frames, 0) * frame_hop
        b = min((e - 1 + pad_frames) * frame_hop + frame_len, n_out)
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
        t = e
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def _slaney_mel_bank(n_mels: int, sr: int = 16000, n_fft: int = 400) -> np.ndarray:
    f_sp, min_log_hz, logstep = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    hz2mel = lambda f: np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)
    mel2hz = lambda m: np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    mel_f = mel2hz(np.linspace(hz2mel(np.float64(0.0)), hz2mel(np.float64(sr / 2)), n_mels + 2))
    fft_f = np.linspace(0, sr / 2, 1 + n_fft // 2)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def whisper_log_mel(audio: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """whisper/audio.py log_mel_spectrogram(audio, n_mels) without its 30 s padding: [n_mels, len(audio) // 160]."""
    audio = np.asarray(audio, dtype=np.float32)
    n = len(audio)
    if n // 160 == 0:
        return np.zeros((n_mels, 0), np.float32)
    pad = np.pad(audio, 200, mode="reflect")
    nfr = 1 + n // 160
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)).astype(np.float32)     # torch.hann_window(400)
    frames = np.lib.stride_tricks.sliding_window_view(pad, 400)[::160][:nfr] * win
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)[:-1]                             # stft[..., :-1]
    power = (np.abs(spec) ** 2).astype(np.float32).T                                       # [201, frames]
    mel = _slaney_mel_bank(n_mels) @ power
    log_spec = np.log10(np.maximum(mel, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).astype(np.float32)

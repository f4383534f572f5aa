"""Segment-level separator on the MI355X: the counterpart of ``ConformerCssWrapper``.

Mirrors the plug-in surface of ``css/training/conformer_wrapper.py`` (the interface the reference's
README names as THE customisation point): the config dataclasses ``ExtractorCfg`` / ``ConformerCfg`` /
``NnetCfg`` / ``ConformerCssCfg`` with the same field names and defaults, and a separator object with
``stft`` / ``separate`` / ``istft`` / ``forward`` plus the handful of ``nn.Module`` methods the driver
calls (``cpu``, ``to``, ``eval``, ``training``).  All arithmetic runs in libcss_mi355.so (hand-written
gfx950 kernels); torch appears only as the tensor type at this boundary.
"""
from __future__ import annotations

import dataclasses
import glob
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _lib
from .weights import ModelDesc, pack_blob, strip_module_prefix


# The default values mirror conformer_wrapper.py:11-48 (conformer_base MC model).
@dataclass
class ExtractorCfg:
    ang_index: str = ''
    frame_hop: int = 256
    frame_len: int = 512
    ipd_cos: bool = False
    ipd_index: str = '1,0;2,0;3,0;4,0;5,0;6,0'
    ipd_mean_normalize: bool = True
    ipd_mean_normalize_version: int = 1
    log_spectrogram: bool = False
    mvn_spectrogram: bool = True
    num_spks: int = 2
    round_pow_of_two: bool = True
    window: str = 'hann'


@dataclass
class ConformerCfg:
    attention_dim: int = 256
    attention_heads: int = 4
    dropout_rate: float = 0.1
    kernel_size: int = 33
    linear_units: int = 1024
    num_blocks: int = 16


@dataclass
class NnetCfg:
    conformer_conf: ConformerCfg = field(default_factory=ConformerCfg)
    in_features: int = 1799
    num_nois: int = 1
    num_spks: int = 3


@dataclass
class ConformerCssCfg:
    extractor_conf: ExtractorCfg = field(default_factory=ExtractorCfg)
    nnet_conf: NnetCfg = field(default_factory=NnetCfg)


_SUPPORTED_EXTRACTOR = ExtractorCfg()


def ipd_pairs(ipd_index: str):
    """'1,0;2,0;...' -> [(l, r), ...] as IPDFeature parses it (feature.py:182-188)"""
    return [tuple(int(x) for x in p.split(",")) for p in ipd_index.split(";")] if ipd_index else []


def feature_options(e: ExtractorCfg) -> dict:
    """ExtractorCfg -> keyword arguments of Handle.set_feature_options (css_set_feature_options)"""
    return dict(log_spectrogram=e.log_spectrogram, mvn_spectrogram=e.mvn_spectrogram,
                ipd_mean_normalize=e.ipd_mean_normalize, ipd_mean_normalize_version=e.ipd_mean_normalize_version,
                ipd_cos=e.ipd_cos, pairs=ipd_pairs(e.ipd_index))


def desc_from_cfg(cfg: ConformerCssCfg) -> ModelDesc:
    """ConformerCssCfg -> the C ABI's model descriptor.  The spectral and IPD options of ExtractorCfg
    (`log_spectrogram`, `mvn_spectrogram`, `ipd_index`, `ipd_mean_normalize`, `ipd_mean_normalize_version`, `ipd_cos`) are
    all implemented (css_set_feature_options), and so are both analysis windows init_kernel builds, 'hann' and 'sqrt_hann'
    (css_set_analysis_window; feature.py:24-36).  `frame_len` / `frame_hop` / `round_pow_of_two` (init_kernel,
    feature.py:19-45): N = frame_len rounded up to a power of two (or frame_len itself) FFT points, N / 2 + 1 bins, a window
    of frame_len samples, any hop that is a multiple of 4; 512 / 256 takes the FFT kernel and the pipelined schedules, other
    sizes the DFT-matrix product and the plain stage sequence.  (The reference's own wrapper builds its network with 257
    mask bins whatever the extractor says -- NnetCfg has no num_bins, conformer.py:260 -- so through it only N = 512 can
    run: frame_len in (256, 512] with round_pow_of_two, e.g. 400 / 160.)  Rejected: `ang_index` -- the reference's own wrapper never passes the direction of arrival its AngleFeature needs
    (conformer_wrapper.py:96-100 calls the executor without `doa`), so no model can use it there either."""
    e, n = cfg.extractor_conf, cfg.nnet_conf
    if e.ang_index != _SUPPORTED_EXTRACTOR.ang_index:
        raise NotImplementedError(f"extractor_conf.ang_index={e.ang_index!r} is not supported by the HIP front end "
                                  f"(supported: {_SUPPORTED_EXTRACTOR.ang_index!r})")
    if e.window not in _lib.ANALYSIS_WINDOWS:
        raise RuntimeError("Now only support sqrt hanning window or hann window")   # feature.py:24-25
    n_fft = 2 ** int(np.ceil(np.log2(e.frame_len))) if e.round_pow_of_two else e.frame_len     # feature.py:27
    if e.frame_len % 4 or e.frame_hop % 4 or not (4 <= e.frame_hop <= e.frame_len) or e.frame_len < 32 or n_fft % 2:
        raise NotImplementedError("frame_len / frame_hop must be multiples of 4, 4 <= frame_hop <= frame_len, frame_len >= 32")
    pairs = ipd_pairs(e.ipd_index)
    if len(pairs) > 16:
        raise NotImplementedError("at most 16 IPD pairs")
    if e.ipd_mean_normalize and e.ipd_mean_normalize_version not in (1, 2, 3):
        raise RuntimeError(f"expect ipd_mean_normalization version 1, 2 or 3, got {e.ipd_mean_normalize_version}")  # feature.py:228-231
    num_mics = (max(max(p) for p in pairs) + 1) if pairs else 1
    num_mics = 7 if pairs and num_mics <= 7 else num_mics        # the NOTSOFAR array (mic_array_model.py:4)
    bins = n_fft // 2 + 1
    assert n.in_features == bins * (1 + len(pairs)), \
        f"in_features={n.in_features} does not match {bins} bins x (1 + {len(pairs)} IPD pairs)"
    c = n.conformer_conf
    return ModelDesc(num_mics=num_mics, num_bins=bins, in_features=n.in_features,
                     attention_dim=c.attention_dim, attention_heads=c.attention_heads,
                     linear_units=c.linear_units, num_blocks=c.num_blocks, kernel_size=c.kernel_size,
                     num_spks=n.num_spks, num_nois=n.num_nois, frame_len=e.frame_len, frame_hop=e.frame_hop)


def _torch():
    import torch
    return torch


def _like(out, ref):
    """Return `out` on the device of the tensor the caller passed in (the reference driver mixes the
    separator's outputs with tensors it moved to `device` itself, css.py:198,216,227)."""
    dev = getattr(ref, "device", None)
    return out.to(dev) if dev is not None and str(dev) != "cpu" else out


def _device_index(device) -> int:
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s.startswith("cpu"):
        return 0  # the separator never computes on the host; `.cpu()` is a no-op (see HipSeparator.cpu)
    if ":" in s:
        return int(s.split(":")[1])
    return 0


class HipSeparator:
    """ConformerCssWrapper counterpart (conformer_wrapper.py:51-146) backed by the HIP library.

    ``linear_mode``: arithmetic of the Conformer's Linear layers and matrix products.  "exact_f32" (default) keeps the
    reference's own operand precision (torch.nn.Linear in float32, conformer.py:137-150) on the float32 matrix instruction;
    "split_f16" is the explicit opt-in to 22-bit operands on the f16 matrix cores (about twice the throughput)."""

    def __init__(self, state_dict: Dict[str, "np.ndarray"], cfg: Optional[ConformerCssCfg] = None,
                 device=0, max_batch_segments: int = 64, stream: int = 0, linear_mode: str = "exact_f32"):
        if linear_mode not in ("exact_f32", "split_f16"):
            raise ValueError(f"linear_mode must be 'exact_f32' or 'split_f16', got {linear_mode!r}")
        st = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))
              for k, v in strip_module_prefix(state_dict).items()}
        desc = desc_from_cfg(cfg) if cfg is not None else None
        self.blob, self.desc = pack_blob(st, desc)
        self.cfg = cfg
        self.training = False
        self._device = _device_index(device)
        self._max_batch = max_batch_segments
        self._stream = stream
        self._linear_mode = linear_mode
        self._handle: Optional[_lib.Handle] = None

    # ---- nn.Module surface used by the driver (css.py:88,141,176,178,318)
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("the HIP separator is inference-only")
        return self.eval()

    def cpu(self):
        """The reference moves the module to the host for the whole-meeting STFT / iSTFT "to avoid
        potential GPU memory overflow" (css.py:141,154,318); with 288 GB of HBM nothing moves."""
        return self

    def to(self, device):
        idx = _device_index(device)
        if self._handle is not None and idx != self._device:
            self._handle.close()
            self._handle = None
        self._device = idx
        return self

    @property
    def handle(self) -> _lib.Handle:
        if self._handle is None:
            self._handle = _lib.Handle(self.desc, self.blob, self._device, self._stream, self._max_batch)
            if self._linear_mode != self._handle.linear_mode():   # (a new handle is in "exact_f32")
                self._handle.set_linear_mode(self._linear_mode)
            if self.cfg is not None and self.cfg.extractor_conf != _SUPPORTED_EXTRACTOR:
                self._handle.set_feature_options(**feature_options(self.cfg.extractor_conf))
                self._handle.set_analysis_window(self.cfg.extractor_conf.window)
        return self._handle

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    # ---- separator protocol
    def stft(self, s):
        """[Batch, T, Mics] or [Batch, T] -> complex64 [Batch, F, T', Mics] / [Batch, F, T']
        (conformer_wrapper.py:106-129)."""
        torch = _torch()
        x = s.detach().cpu().numpy() if hasattr(s, "detach") else np.asarray(s)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[:, :, None]
        outs = []
        f = self.desc.num_bins
        for b in range(x.shape[0]):
            planes = self.handle.stft_host(x[b])  # [C, 2F, T']
            outs.append(np.moveaxis(planes[:, :f] + 1j * planes[:, f:], 0, 2).astype(np.complex64))
        out = np.stack(outs)  # [B, F, T', C]
        if squeeze:
            out = out[..., 0]
        return _like(torch.from_numpy(out), s)

    def separate(self, stft):
        """complex [Batch, F, T, Mics] / [Batch, F, T] -> {'spk_masks': [B,F,T,S], 'noise_masks': [B,F,T,1]}
        (conformer_wrapper.py:79-104)."""
        torch = _torch()
        x = stft.detach().cpu().numpy() if hasattr(stft, "detach") else np.asarray(stft)
        assert np.iscomplexobj(x)
        if x.ndim == 3:
            x = x[..., None]
        b, f, t, c = x.shape
        if c != self.desc.num_mics:
            raise _lib.CssError(_lib.CSS_ERR_SHAPE, f"stft has {c} channels, the model expects {self.desc.num_mics}")
        # [B, F, T, C] -> planes [C, 2F, B*T] (segments back to back along time)
        xt = np.moveaxis(x, 3, 0).transpose(0, 2, 1, 3).reshape(c, f, b * t)  # [C, F, B*T]
        planes = np.concatenate([xt.real, xt.imag], axis=1).astype(np.float32)
        m = self.handle.separate_host(planes, b)  # [(S+1)F, B*T]
        nm = self.desc.num_spks + self.desc.num_nois
        m = m.reshape(nm, f, b, t).transpose(2, 1, 3, 0)  # [B, F, T, S+1]
        m = _like(torch.from_numpy(np.ascontiguousarray(m)), stft)
        return {'spk_masks': m[..., :self.desc.num_spks], 'noise_masks': m[..., self.desc.num_spks:]}

    def istft(self, stft):
        """complex [Batch, F, T] -> [Batch, NSamples] (conformer_wrapper.py:131-146)."""
        torch = _torch()
        x = stft.detach().cpu().numpy() if hasattr(stft, "detach") else np.asarray(stft)
        assert np.iscomplexobj(x) and x.ndim == 3
        planes = np.concatenate([x.real, x.imag], axis=1).astype(np.float32)
        return _like(torch.from_numpy(self.handle.istft_host(planes)), stft)

    def forward(self, mix):
        """[Batch, T, Mics] time-domain mixture -> masks (conformer_wrapper.py:58-77)."""
        assert (mix.shape[2] == 1) == (self.desc.num_mics == 1), \
            "IPD extractor is expected iff the number of microphones is greater than 1"
        # fused on the device (css_forward_host): batched analysis transform, features, mask estimator
        torch = _torch()
        x = mix.detach().cpu().numpy() if hasattr(mix, "detach") else np.asarray(mix)
        b = x.shape[0]
        m = self.handle.forward_host(x)  # [(S+1)F, B*T']
        nm, f = self.desc.num_spks + self.desc.num_nois, self.desc.num_bins
        m = m.reshape(nm, f, b, -1).transpose(2, 1, 3, 0)  # [B, F, T', S+1]
        m = _like(torch.from_numpy(np.ascontiguousarray(m)), mix)
        return {'spk_masks': m[..., :self.desc.num_spks], 'noise_masks': m[..., self.desc.num_spks:]}

    __call__ = forward


def _cfg_from_yaml(path: str) -> Optional[ConformerCssCfg]:
    """Reads ``conformer_css_cfg`` out of the TrainCfg yaml shipped beside a checkpoint
    (css/helpers.py:24-28; the reference parses it with OmegaConf into TrainCfg, train.py:48-80)."""
    import yaml
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    node = y.get("conformer_css_cfg")
    if node is None:
        return ConformerCssCfg()

    nested = {(ConformerCssCfg, "extractor_conf"): ExtractorCfg, (ConformerCssCfg, "nnet_conf"): NnetCfg,
              (NnetCfg, "conformer_conf"): ConformerCfg}

    def build(cls, data):
        data = data or {}
        unknown = set(data) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError(f"unknown keys in {cls.__name__}: {sorted(unknown)}")  # OmegaConf would reject too
        kw = {}
        for name, v in data.items():
            sub = nested.get((cls, name))
            kw[name] = build(sub, v) if sub is not None else v
        return cls(**kw)

    return build(ConformerCssCfg, node)


def load_css_model(model_dir, device=0, max_batch_segments: int = 64, linear_mode: str = "exact_f32"):
    """Counterpart of css/helpers.py:14-37: a directory with exactly one ``*.yaml`` (TrainCfg) and one
    ``*.pt`` (``ckpt['model']``, keys prefixed ``module.``) -> (separator, cfg).  ``linear_mode``: see HipSeparator."""
    def one(suffix):
        files = glob.glob(os.path.join(str(model_dir), suffix))
        if len(files) == 0:
            raise FileNotFoundError(f'expecting at least one {suffix} file in {model_dir}')
        assert len(files) == 1, f'expecting exactly one {suffix} file in {model_dir}'
        return files[0]

    yaml_path, ckpt_path = one('*.yaml'), one('*.pt')
    cfg = _cfg_from_yaml(yaml_path)
    torch = _torch()
    ckpt = torch.load(ckpt_path, map_location="cpu")
    state = {k[len("module."):]: v for k, v in ckpt["model"].items() if k.startswith("module.")}
    return HipSeparator(state, cfg, device=device, max_batch_segments=max_batch_segments, linear_mode=linear_mode), cfg

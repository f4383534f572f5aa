"""Long-form continuous speech separation on the MI355X: the counterpart of the reference's ``css/css.py``.

Same plug-in surface -- ``CssCfg``, ``css_inference``, ``separate_and_stitch``, ``calc_segment_weight``
with the reference's argument names, return types and assertion behaviour -- so that the rest of the
NOTSOFAR pipeline (Whisper ASR, diarization) consumes the ``sep_stream{i}.wav`` files unchanged.  The
arithmetic of ``separate_and_stitch`` (css/css.py:110-338) runs in one fused pass of hand-written
gfx950 kernels behind the C ABI of ``include/css_mi355.h``: one upload of the PCM, every segment of
the meeting batched through the mask estimator, MVDR / stitching / gating / inverse transform on the
device, one download of the separated waveforms.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .separator import HipSeparator, load_css_model
from .wavio import NUM_MICS_MC, load_audio, read_wav_pcm16, write_pcm16_samples, write_wav

_LOG = logging.getLogger('css')


# CSS inference configuration: the reference's CssCfg (css/css.py:24-48) field for field -- names, types and defaults are
# the drop-in contract (configs/inference/*.yaml are merged into it); the notes say where each one acts on the HIP path.
@dataclass
class CssCfg:
    segment_size_sec: float = 3.           # sliding window: length ...
    hop_size_sec: float = 1.5              # ... and stride (css.py:144-171 -> CssRunCfg.segment_frames / hop_frames)
    normalize_segment_power: bool = False  # rescale every separated segment to the mixture's power (css.py:233-247)
    stitching_loss: str = 'l1'             # cost that aligns adjacent segments: 'l1' | 'mse' (css.py:263)
    stitching_input: str = 'mask'          # what the cost compares: 'mask' | 'separation_result' (css.py:267-271)
    seg_weight_m0_sec: float = 0.15        # overlap-add window: zero up to m0 ...
    seg_weight_m1_sec: float = 0.3         # ... ramp up to m1 (calc_segment_weight)
    activity_th: float = 0.4               # gate: mean mask over frequency >= this (css.py:303-304; shipped yaml: 0.3)
    activity_dilation_sec: float = 0.4     # gate smoothing: dilate ...
    activity_erosion_sec: float = 0.2      # ... then erode (css.py:305-308)
    device: Optional[str] = None           # (ignored by the reference too: css.py:87)
    show_progressbar: bool = True          # (no per-segment loop to show here)
    checkpoint_sc: str = 'notsofar/conformer1.0/sc'   # model directory under models_dir, one microphone
    checkpoint_mc: str = 'notsofar/conformer1.0/mc'   # ... seven microphones
    device_id: int = 0                     # GPU the session runs on
    num_spks: int = 3                      # output streams of the model
    mc_mvdr: bool = True                   # multi-channel: MVDR beamformer instead of masking channel 0 (css.py:211-221)
    mc_mask_floor_db: float = 0.           # floor of the mask applied after it, dB <= 0; 0 dB: the mask does nothing
    sc_mask_floor_db: float = -np.inf      # ... single channel; -inf: plain mask multiplication (css.py:222-227)
    pass_through_ch0: bool = False         # skip separation, hand channel 0 on (css.py:73-75)
    slice_audio_for_debug: bool = False    # separate only seconds 20 .. 30 (css.py:91-92)


def _linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """float32 linspace with ATen's evaluation order (symmetric halves, fused multiply-add), so that the
    taper of calc_segment_weight is bit-identical to the reference's torch.linspace(0.1, 1, n)."""
    start, end = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((end - start) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    for i in range(steps):
        if i < steps // 2:
            out[i] = np.float32(np.float64(start) + np.float64(step) * i)
        else:
            out[i] = np.float32(np.float64(end) - np.float64(step) * (steps - i - 1))
    return out


def calc_segment_weight(seg_frames: int, m0_frames: int, m1_frames: int,
                        is_first_seg: bool = False, is_last_seg: bool = False) -> np.ndarray:
    """Trapezoid weighting window of the overlap-add (css/css.py:341-390): 0 on the outer m0 frames,
    linear 0.1 -> 1 between m0 and m1, 1 in the middle, mirrored on the right; the first (last) segment of
    the recording keeps 0.1 instead of 0 on its left (right) edge because nothing else covers it."""
    assert seg_frames > 2 * m1_frames, \
        'not enough frames to fit weighting window. try modifying hop_size, segment_size or m0, m1'
    w = np.ones(seg_frames, dtype=np.float32)
    w[:m0_frames] = 0
    w[seg_frames - m0_frames:] = 0
    linear = _linspace_f32(0.1, 1, m1_frames - m0_frames)
    w[m0_frames:m1_frames] = linear
    w[seg_frames - m1_frames:seg_frames - m0_frames] = linear[::-1]
    if is_first_seg:
        w[:m0_frames] = 0.1
    if is_last_seg:
        w[seg_frames - m0_frames:] = 0.1
    return w


def make_run_cfg(cfg: CssCfg, fs: int, num_channels: int, frame_len: int = 512, frame_hop: int = 256) -> _lib.RunCfg:
    """Seconds -> frames with the reference's own float expressions (css/css.py:144-152) and the
    knobs the C ABI needs."""
    seg_samples = int(cfg.segment_size_sec * fs)
    segment_frames = (seg_samples - frame_len) // frame_hop + 1  # length of the dummy STFT, css.py:145-147
    hop_frames = int(segment_frames * cfg.hop_size_sec / cfg.segment_size_sec)
    # what the kernels take: segments of 2 frames and more (up to 512 frames = 8 s on the tuned kernels, beyond that on
    # their any-length forms; the bound is a sanity limit of 262 s) and any hop with at least one frame of overlap for the
    # stitching cost (css.py:276 compares the overlapping frames of adjacent segments; hop == segment has none)
    if not (2 <= segment_frames <= _lib.MAX_SEGMENT_FRAMES and 0 < hop_frames < segment_frames):
        raise NotImplementedError(
            f"segment_size_sec={cfg.segment_size_sec} / hop_size_sec={cfg.hop_size_sec} give {segment_frames}-frame segments "
            f"every {hop_frames} frames; the HIP path covers segments of 2..{_lib.MAX_SEGMENT_FRAMES} frames with 1 <= hop < segment")
    m0_frames = int(segment_frames * cfg.seg_weight_m0_sec / cfg.segment_size_sec)
    m1_frames = int(segment_frames * cfg.seg_weight_m1_sec / cfg.segment_size_sec)
    dilation_frames = int(segment_frames * cfg.activity_dilation_sec / cfg.segment_size_sec)
    erosion_frames = int(segment_frames * cfg.activity_erosion_sec / cfg.segment_size_sec)
    mask_floor_db = cfg.mc_mask_floor_db if num_channels > 1 else cfg.sc_mask_floor_db   # css.py:223
    assert mask_floor_db <= 0                                                            # css.py:224
    mask_floor = 10. ** (mask_floor_db / 20.)
    assert cfg.stitching_loss in ('l1', 'mse'), f'unexpected stitching_loss: {cfg.stitching_loss}'
    assert cfg.stitching_input in ('mask', 'separation_result'), f'unexpected stitching_input: {cfg.stitching_input}'
    return _lib.RunCfg(
        segment_frames, hop_frames, dilation_frames, erosion_frames, cfg.mc_mvdr,
        {'l1': 0, 'mse': 1}[cfg.stitching_loss], {'mask': 0, 'separation_result': 1}[cfg.stitching_input],
        cfg.normalize_segment_power, mask_floor, cfg.activity_th,
        calc_segment_weight(segment_frames, m0_frames, m1_frames, is_first_seg=True),
        calc_segment_weight(segment_frames, m0_frames, m1_frames),
        calc_segment_weight(segment_frames, m0_frames, m1_frames, is_last_seg=True))


def _maybe_torch(arr: np.ndarray):
    try:
        import torch
        return torch.from_numpy(arr)
    except Exception:  # pragma: no cover
        return arr


def separate_and_stitch(speech_mix: np.ndarray, separator, fs: int, device, cfg: CssCfg,
                        return_side_info: bool = True) -> (List[np.ndarray], Dict):
    """Applies speech separation in block-online fashion (css/css.py:110-338).

    Args:
        speech_mix: long-form input [Batch, Nsamples, Channels] float32 (Channels == 1 or 7).
        separator: a ``HipSeparator`` (see separator.py).
        fs: sample rate.
        device: GPU to run on (torch.device / 'cuda:N' / int).
        cfg: CSS configuration.
        return_side_info: False skips the three device-to-host reads behind ``side_info`` (the stitched masks alone
            are 347 MB for a 30-min meeting) and returns an empty dict; the reference always builds it.
    Returns:
        separated_wavs: list of ``cfg.num_spks`` float32 arrays [Nsamples_out].
        side_info: dict with 'mask_stitched' [1, F, T_long, S], 'activity_b' [T_long, S],
            'activity_final' [1, T_long, S] and 'segment_frames', as in the reference.
    """
    assert speech_mix.ndim == 3, f'expecting 3 dimensions, got {speech_mix.shape}'
    assert speech_mix.shape[0] == 1, 'assuming 1 example in batch. easy to support more.'
    if not isinstance(separator, HipSeparator):
        # the reference's customisation point (README: "implement stft / separate / istft"): any object with the
        # separator protocol supplies the masks, everything around it runs on the HIP stages
        return _separate_and_stitch_protocol(speech_mix, separator, fs, device, cfg, return_side_info)
    assert not separator.training
    separator.to(device)
    desc = separator.desc
    assert cfg.num_spks == desc.num_spks, f"cfg.num_spks={cfg.num_spks} but the model separates {desc.num_spks}"
    n, c = speech_mix.shape[1], speech_mix.shape[2]
    run_cfg = make_run_cfg(cfg, fs, c, desc.frame_len, desc.frame_hop)
    h = separator.handle
    wav = h.run(speech_mix[0], run_cfg)  # [S, n_out]
    separated_wavs = [wav[k] for k in range(desc.num_spks)]
    if not return_side_info:
        return separated_wavs, {}

    mask_st = h.read(_lib.BUF_MASK_ST)                       # [S, F, T_long]
    act_b = h.read(_lib.BUF_ACT_B).astype(bool)              # [S, T_long]
    act_final = h.read(_lib.BUF_ACT_FINAL).astype(bool)
    side_info = {
        'mask_stitched': _maybe_torch(np.ascontiguousarray(np.transpose(mask_st, (1, 2, 0)))[None]),
        'activity_b': _maybe_torch(np.ascontiguousarray(act_b.T)),
        'activity_final': _maybe_torch(np.ascontiguousarray(act_final.T)[None]),
        'segment_frames': int(run_cfg.c.segment_frames),
    }
    return separated_wavs, side_info


_STAGE_SEPARATORS: Dict[tuple, HipSeparator] = {}


def _stage_separator(num_mics: int, num_spks: int, device) -> HipSeparator:
    """A handle for the HIP stages around a foreign mask estimator: the library wants a model at css_create, so this is a
    one-block stand-in (2 MB of seeded weights) whose estimator is never launched.  One per (microphones, speakers, GPU)."""
    from .separator import _device_index
    from .weights import ModelDesc, portable_state_dict
    key = (num_mics, num_spks, _device_index(device))
    sep = _STAGE_SEPARATORS.get(key)
    if sep is None:
        bins = 257
        desc = ModelDesc(num_mics=num_mics, num_bins=bins, in_features=bins * (1 + (num_mics - 1 if num_mics > 1 else 0)),
                         attention_dim=256, attention_heads=4, linear_units=256, num_blocks=1, num_spks=num_spks)
        sep = _STAGE_SEPARATORS[key] = HipSeparator(portable_state_dict(desc, 0), None, device=device)
        assert (sep.desc.num_mics, sep.desc.num_spks) == (num_mics, num_spks)
    return sep


def _separate_and_stitch_protocol(speech_mix: np.ndarray, separator, fs: int, device, cfg: CssCfg, return_side_info: bool):
    """css/css.py:110-338 for ANY object that honours the separator protocol (css.py:131: `stft`, `separate`, `istft`;
    conformer_wrapper.py:79-146).  As in the reference, `separator.separate(stft_seg)` is called once per segment with a
    complex tensor [1, F, T, C] on `device` (css.py:183-199; zero-padded last segment; C == 1 keeps its axis) and returns
    {'spk_masks': [1, F, T, S], 'noise_masks': [1, F, T, 1]}.  Everything else -- analysis transform of the whole
    recording, winner-take-all masks / covariances / MVDR, mask floor, stitching, gate, synthesis -- runs on the HIP
    stages of the C ABI (css_begin, css_stage_stft, css_stage_mvdr, ..., css_stage_istft); the masks go straight into the
    library's device buffer through a zero-copy torch view.  Nothing is computed on the host.

    The stages implement the transform of the reference's ConformerCssWrapper in the stand-in's geometry (the shipped one:
    512-sample Hann analysis window, hop 256, sqrt-Hann synthesis, feature.py:19-45,88-167); a separator whose own `stft` is a
    different transform is rejected: its masks would not belong to these spectra."""
    import torch
    from .parallel import HipShardBackend
    assert not getattr(separator, "training", False)
    n, c = speech_mix.shape[1], speech_mix.shape[2]
    S = cfg.num_spks
    stage = _stage_separator(c, S, device)
    desc = stage.desc
    run_cfg = make_run_cfg(cfg, fs, c, desc.frame_len, desc.frame_hop)
    h = stage.handle
    dev = torch.device("cuda", stage._device)
    if hasattr(separator, "to"):
        separator.to(dev)
    be = HipShardBackend(h, dev)
    F, T, hop = desc.num_bins, int(run_cfg.c.segment_frames), int(run_cfg.c.hop_frames)
    try:
        h.begin(np.ascontiguousarray(speech_mix[0], dtype=np.float32), n, c, run_cfg)
        h.stage_stft()
        plan = h.get_plan()
        nseg, TL, frames = int(plan.num_segments), int(plan.mix_frames), int(plan.stft_frames)
        with be.on_stream(), torch.no_grad():
            X = be._view(_lib.BUF_X, "<f4")                      # [C, 2F, T_ld]: rows 0..F-1 real, F..2F-1 imaginary
            masks = be._view(_lib.BUF_MASKS, "<f4")              # [(S + 1) F, nseg * T]
            assert tuple(masks.shape) == ((S + 1) * F, nseg * T), (masks.shape, S, F, nseg, T)
            # the separator's own transform must be the stages' transform (checked on the first frames of the recording)
            if hasattr(separator, "stft"):
                # [B, N, C] in, [B, F, T, C] out, also for C == 1: what css.py:155 hands over
                probe = torch.from_numpy(np.ascontiguousarray(speech_mix[:, :desc.frame_len + 3 * desc.frame_hop])).to(dev)
                theirs = separator.stft(probe)
                theirs = theirs.reshape(1, F, -1, c) if theirs.ndim == 3 else theirs
                k = min(theirs.shape[2], frames)
                ours = torch.complex(X[:, :F, :k], X[:, F:2 * F, :k]).permute(1, 2, 0)[None]
                scale = float(ours.abs().max()) + 1e-30
                if tuple(theirs.shape[:2]) != (1, F) or float((theirs[:, :, :k].to(dev) - ours).abs().max()) > 1e-3 * scale:
                    raise ValueError(f"the separator's stft() is not the transform of the HIP stages ({desc.frame_len}-sample Hann window, hop "
                                     f"{desc.frame_hop}, conformer_wrapper.py:106-129): its masks cannot be applied to these spectra")
            for i in range(nseg):                                # css.py:182-250, one call per segment as in the reference
                t = max(min(T, frames - i * hop), 0)
                seg = torch.zeros((1, F, T, c), dtype=torch.complex64, device=dev)
                if t > 0:
                    blk = X[:, :, i * hop:i * hop + t]
                    seg[0, :, :t, :] = torch.complex(blk[:, :F], blk[:, F:2 * F]).permute(1, 2, 0)
                out = separator.separate(seg)                    # always [1, F, T, C], C == 1 included (css.py:199)
                spk, noi = out['spk_masks'], out['noise_masks']
                assert tuple(spk.shape) == (1, F, T, S), f'spk_masks {tuple(spk.shape)}, expected {(1, F, T, S)}'   # css.py:202
                assert tuple(noi.shape) == (1, F, T, 1), f'noise_masks {tuple(noi.shape)}, expected {(1, F, T, 1)}'  # css.py:203
                both = torch.cat([spk, noi], dim=3)[0].to(device=dev, dtype=torch.float32)      # [F, T, S + 1]
                masks[:, i * T:(i + 1) * T] = both.permute(2, 0, 1).reshape((S + 1) * F, T)
        h.stage_mvdr(0, nseg)
        h.stage_pit_costs(0, nseg - 1)
        h.stage_pit_scan()
        h.stage_stitch(0, TL)
        h.stage_istft(0, TL)
        h.check_range()
        wav = h.read(_lib.BUF_WAV)
        separated_wavs = [wav[k] for k in range(S)]
        if not return_side_info:
            return separated_wavs, {}
        mask_st = h.read(_lib.BUF_MASK_ST)
        act_b = h.read(_lib.BUF_ACT_B).astype(bool)
        act_final = h.read(_lib.BUF_ACT_FINAL).astype(bool)
        return separated_wavs, {
            'mask_stitched': _maybe_torch(np.ascontiguousarray(np.transpose(mask_st, (1, 2, 0)))[None]),
            'activity_b': _maybe_torch(np.ascontiguousarray(act_b.T)),
            'activity_final': _maybe_torch(np.ascontiguousarray(act_final.T)[None]),
            'segment_frames': T,
        }
    finally:
        be.close()
        if hasattr(separator, "cpu"):
            separator.cpu()                                      # css.py:318: the separator goes back to the host


def close_stage_separators() -> None:
    """Releases the stand-in handles `_separate_and_stitch_protocol` keeps per (microphones, speakers, GPU)."""
    while _STAGE_SEPARATORS:
        _, sep = _STAGE_SEPARATORS.popitem()
        sep.close()


@dataclass
class _SessionOutput:
    """Where one session's CSS result lives and which shortcut, if any, answers the call without separating
    (the rules of css/css.py:70-82: pass-through of channel 0, cache hit on an existing output directory)."""
    directory: Path
    shortcut: Optional[list]

    @classmethod
    def plan(cls, out_dir, session, cfg: CssCfg, fetch_from_cache: bool) -> "_SessionOutput":
        assert isinstance(session.wav_file_names, list)
        directory = Path(out_dir) / "css_inference" / session.session_id
        if cfg.pass_through_ch0:
            return cls(directory, session.wav_file_names[0:1])
        if fetch_from_cache and directory.exists():
            return cls(directory, sorted(directory.glob('sep*.wav')))
        return cls(directory, None)

    def stream_path(self, i: int) -> Path:
        return self.directory / f"sep_stream{i}.wav"


def _separate_pcm16_session(separator, raw, cfg: CssCfg, device):
    """The device-side wav edges: raw int16 planes up, peak-normalised PCM16 streams down."""
    sr = raw[0][1]
    separator.to(device)
    desc = separator.desc
    run_cfg = make_run_cfg(cfg, sr, len(raw), desc.frame_len, desc.frame_hop)
    pcm16, _ = separator.handle.run_pcm16([r[0] for r in raw], run_cfg)
    mixture = raw[0][0].astype(np.float32) / np.float32(32768.0)
    return sr, mixture, [pcm16[i] for i in range(pcm16.shape[0])], True


def _separate_float_session(separator, session, cfg: CssCfg, device):
    mixwav, sr = load_audio(session.wav_file_names, is_mc=session.is_mc)
    if cfg.slice_audio_for_debug:
        mixwav = mixwav[:, sr * 20:sr * 30, :]
    wavs, _ = separate_and_stitch(mixwav, separator, sr, device, cfg, return_side_info=False)
    return sr, mixwav[0, :, 0], wavs, False


def css_inference(out_dir: str, models_dir: str, session, cfg: CssCfg, fetch_from_cache: bool, separator=None):
    """Applies CSS to one session row: the counterpart of css/css.py:51-107, same arguments, same files
    (``out_dir/css_inference/<session_id>/{input_mixture,sep_stream0..}.wav``), same returned Series (a copy of the
    session with ``sep_wav_file_names``).

    ``separator``: a model already resident on the GPU (the session loop of pipeline.py keeps one per model kind); by
    default the checkpoint under ``models_dir`` is loaded for this call and released afterwards, as the reference does.
    Mono 16-bit PCM inputs -- what the NOTSOFAR recordings are -- take the device-side wav edges: int16 over PCIe,
    scaling, peak normalisation and PCM16 encoding on the GPU."""
    _LOG.info("Running CSS (Continuous Speech Separation)")
    result = session.copy()
    where = _SessionOutput.plan(out_dir, session, cfg, fetch_from_cache)
    if where.shortcut is not None:
        result['sep_wav_file_names'] = where.shortcut
        return result

    owned = separator is None
    if owned:
        separator, _ = load_css_model(Path(models_dir) / (cfg.checkpoint_mc if session.is_mc else cfg.checkpoint_sc))
    separator.eval()
    device = f"cuda:{cfg.device_id}"
    try:
        raw = None if cfg.slice_audio_for_debug else [read_wav_pcm16(p) for p in session.wav_file_names]
        same_shape = raw and all(r is not None for r in raw) and len({(r[0].shape[0], r[1]) for r in raw}) == 1
        if same_shape:
            assert len(raw) == (NUM_MICS_MC if session.is_mc else 1), f'expecting {NUM_MICS_MC} microphones'
            sr, mixture, streams, encoded = _separate_pcm16_session(separator, raw, cfg, device)
        else:
            sr, mixture, streams, encoded = _separate_float_session(separator, session, cfg, device)
    finally:
        if owned:
            separator.close()

    write_wav(where.directory / 'input_mixture.wav', samps=mixture, sr=sr)
    names = []
    for i, samples in enumerate(streams):
        path = where.stream_path(i)
        _LOG.info(f"CSS: saving separated wav to {path}")
        if encoded:
            write_pcm16_samples(path, samples, sr)
        else:
            write_wav(path, samps=samples, sr=sr)
        names.append(str(path))
    result['sep_wav_file_names'] = names
    return result

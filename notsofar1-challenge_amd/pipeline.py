"""Session loop around the CSS stage: the thin counterpart of the CSS leg of
``inference_pipeline/inference.py:37-107``.

The reference iterates the sessions of ``all_session_df`` one by one in a single process
(``inference.py:59``: "sessions are independent by challenge rule") and reloads the checkpoint for every
session (``css.py:85``).  Here

* a rank takes every ``world``-th session (the scheme of the reference's unused ``DDPRowIterator``,
  ``utils/torch_utils.py:48-99``) -- dev-set-1 has 106 multi-channel sessions of ~6 min, which shard better
  by session than by segment;
* the separator of each model kind (multi-/single-channel) is loaded once per process and stays resident
  in HBM across sessions;
* everything else -- directory layout, wav naming, cache rule, ``pass_through_ch0`` -- is ``css_inference``.

ASR, diarization and scoring are not part of this package: their inputs are the wav files and the
``sep_wav_file_names`` column this loop produces.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Dict, Optional

from .css import CssCfg, separate_and_stitch
from .separator import load_css_model
from .wavio import load_audio, write_wav

_LOG = logging.getLogger('css')


def _rank_world(rank: Optional[int], world: Optional[int]):
    # utils/torch_utils.py:10-11: WORLD_SIZE / RANK from the environment (torchrun)
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    return rank, world


def css_sessions(out_dir: str, models_dir: str, sessions_df, cfg: CssCfg, fetch_from_cache: bool = False,
                 rank: Optional[int] = None, world: Optional[int] = None, device=None):
    """Run CSS on the sessions this rank owns; returns a DataFrame with those rows plus
    ``sep_wav_file_names`` (same per-row contract as ``css_inference``, css/css.py:51-107)."""
    import pandas as pd
    rank, world = _rank_world(rank, world)
    if device is None:
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', cfg.device_id))}"
    separators: Dict[bool, object] = {}
    rows = []
    for pos in range(rank, len(sessions_df), world):
        session = sessions_df.iloc[pos]
        session_css = session.copy()
        assert isinstance(session.wav_file_names, list)
        if cfg.pass_through_ch0:                                                   # css.py:73-75
            session_css['sep_wav_file_names'] = session.wav_file_names[0:1]
            rows.append(session_css)
            continue
        css_out_dir = Path(out_dir) / "css_inference" / session.session_id
        if fetch_from_cache and css_out_dir.exists():                              # css.py:78-82
            session_css['sep_wav_file_names'] = sorted(css_out_dir.glob('sep*.wav'))
            rows.append(session_css)
            continue
        is_mc = bool(session.is_mc)
        if is_mc not in separators:
            sep, _ = load_css_model(Path(models_dir) / (cfg.checkpoint_mc if is_mc else cfg.checkpoint_sc),
                                    device=device)
            separators[is_mc] = sep.eval()
        mixwav, sr = load_audio(session.wav_file_names, is_mc=is_mc)
        if cfg.slice_audio_for_debug:
            mixwav = mixwav[:, sr * 20:sr * 30, :]
        _LOG.info(f"CSS [{rank}/{world}] session {session.session_id}: {mixwav.shape[1] / sr:.1f} s")
        separated_wavs, _ = separate_and_stitch(mixwav, separators[is_mc], sr, device, cfg)
        write_wav(css_out_dir / 'input_mixture.wav', samps=mixwav[0, :, 0], sr=sr)
        names = []
        for i, w in enumerate(separated_wavs):
            filename = css_out_dir / f"sep_stream{i}.wav"
            write_wav(filename, samps=w, sr=sr)
            names.append(str(filename))
        session_css['sep_wav_file_names'] = names
        rows.append(session_css)
    for sep in separators.values():
        sep.close()
    return pd.DataFrame(rows)

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

from .css import CssCfg, css_inference
from .separator import _device_index, load_css_model

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
    ``sep_wav_file_names`` -- every row is exactly what ``css_inference`` returns for it (css/css.py:51-107)."""
    import dataclasses
    import pandas as pd
    rank, world = _rank_world(rank, world)
    if device is None:
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', cfg.device_id))}"
    # one index for the resident model AND for css_inference's `separator.to(cuda:<device_id>)`: an int, "cuda:1" or a
    # torch.device all name the same GPU (a mismatch would rebuild the resident handle on another device per session)
    if isinstance(device, int) or ":" in str(device):
        dev_index = _device_index(device)
    else:   # "cuda" / torch.device("cuda") without an index: the configured device
        dev_index = int(cfg.device_id)
    device = f"cuda:{dev_index}"
    cfg = dataclasses.replace(cfg, device_id=dev_index)
    resident: Dict[bool, object] = {}
    rows = []
    try:
        for pos in range(rank, len(sessions_df), world):
            session = sessions_df.iloc[pos]
            is_mc = bool(session.is_mc)
            shortcut = cfg.pass_through_ch0 or (fetch_from_cache and (Path(out_dir) / "css_inference" / session.session_id).exists())
            if not shortcut and is_mc not in resident:   # one resident model per kind, loaded on first use
                resident[is_mc], _ = load_css_model(Path(models_dir) / (cfg.checkpoint_mc if is_mc else cfg.checkpoint_sc),
                                                    device=device)
            _LOG.info(f"CSS [{rank}/{world}] session {session.session_id}")
            rows.append(css_inference(out_dir, models_dir, session, cfg, fetch_from_cache, separator=resident.get(is_mc)))
    finally:
        for sep in resident.values():
            sep.close()
    return pd.DataFrame(rows)

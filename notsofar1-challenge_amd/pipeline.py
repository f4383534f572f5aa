"""Session loop around the CSS stage: the counterpart of the CSS leg of
``inference_pipeline/inference.py:37-107``.

The reference iterates the sessions of ``all_session_df`` one by one in a single process
(``inference.py:59-63``: "sessions are independent by challenge rule"), reloads the checkpoint for every
session (``css.py:85``) and, per session, reads 7 wav files, separates, writes 4 (``css.py:51-107``).  Here

* a rank takes every ``world``-th session (the scheme of the reference's unused ``DDPRowIterator``,
  ``utils/torch_utils.py:48-99``) -- dev-set-1 has 106 multi-channel sessions of ~6 min, which shard better
  by session than by segment;
* the separator of each model kind (multi-/single-channel) is loaded once per process and stays resident
  in HBM across sessions;
* the sessions of a rank go through the library's QUEUE (round 6): ``css_run_enqueue_pcm16`` / ``css_wait``,
  i.e. the schedule ``bench.py`` times -- queued sessions share mask-estimator batches, a session's PCIe legs
  hide under its neighbours' kernels, both wav edges (int16 -> float scaling; peak normalisation and PCM16
  encoding, ``utils/audio_utils.py:37-49``) run on the device -- while worker threads read the NEXT
  sessions' wav payloads straight into page-locked memory (no intermediate copy) and write the PREVIOUS
  sessions' four files (``input_mixture.wav`` as a task of its own as soon as the session is queued).  Every
  file holds what ``css_inference`` writes for that session, bit for bit (``tests/test_hip_session.py``);
* everything else -- directory layout, wav naming, cache rule, ``pass_through_ch0``, sessions whose files are
  not mono 16-bit PCM -- is ``css_inference`` itself, one synchronous call per such session.

ASR, diarization and scoring are not part of this package: their inputs are the wav files and the
``sep_wav_file_names`` column this loop produces.
"""
from __future__ import annotations

import logging
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .css import CssCfg, _SessionOutput, css_inference, make_run_cfg
from .separator import _device_index, load_css_model
from .wavio import NUM_MICS_MC, probe_wav_pcm16, read_pcm16_payload_into, write_pcm16_samples, write_wav

_LOG = logging.getLogger('css')


def _rank_world(rank: Optional[int], world: Optional[int]):
    # utils/torch_utils.py:10-11: WORLD_SIZE / RANK from the environment (torchrun)
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    return rank, world


class _PinnedPool:
    """Page-locked blocks (css_host_alloc) re-used across sessions: allocating 13 MB of page-locked memory takes
    milliseconds, a session of the queue about nine."""

    def __init__(self):
        self._free: List[np.ndarray] = []
        self._lock = threading.Lock()

    def take(self, nbytes: int) -> np.ndarray:
        with self._lock:
            fit = [b for b in self._free if b.nbytes >= nbytes]
            if fit:
                best = min(fit, key=lambda b: b.nbytes)
                self._free = [b for b in self._free if b is not best]
                return best
        return _lib.pinned_empty((max(int(nbytes * 1.1), 1 << 16),), np.uint8)   # (10 % head room: sessions of similar lengths re-use it)

    def give(self, block: np.ndarray):
        with self._lock:
            self._free.append(block)
            while len(self._free) > 96 or sum(b.nbytes for b in self._free) > (4 << 30):
                self._free.pop(0)

    def clear(self):
        with self._lock:
            self._free = []


_POOL = _PinnedPool()   # process-wide: a second css_sessions call (or a long one) finds the blocks of the sessions before


class _Loaded:
    """One session decoded for the queue: its mono PCM16 planes in ONE page-locked block ([C][n] int16)."""
    __slots__ = ("pos", "session", "where", "sr", "block", "planes", "out_block", "out16", "peaks", "fallback", "mixture_written")


def _decode_session(pos, session, where, cfg: CssCfg, pool: _PinnedPool) -> _Loaded:
    ld = _Loaded()
    ld.pos, ld.session, ld.where, ld.fallback = pos, session, where, False
    ld.block = ld.planes = ld.out_block = ld.out16 = ld.peaks = ld.mixture_written = None
    probes = None if cfg.slice_audio_for_debug else [probe_wav_pcm16(p) for p in session.wav_file_names]
    same = probes and all(r is not None for r in probes) and len({(r[0], r[1]) for r in probes}) == 1
    if not same:                       # float / multi-channel / 24-bit files, the debug slice: css_inference's own float path
        ld.fallback = True
        return ld
    assert len(probes) == (NUM_MICS_MC if session.is_mc else 1), f'expecting {NUM_MICS_MC} microphones'
    n, c = probes[0][0], len(probes)
    ld.sr = probes[0][1]
    ld.block = pool.take(n * c * 2)
    planes = ld.block[:n * c * 2].view(np.int16).reshape(c, n)
    for k in range(c):                 # straight from the file into the page-locked plane
        read_pcm16_payload_into(session.wav_file_names[k], probes[k][2], planes[k])
    ld.planes = [planes[k] for k in range(c)]
    return ld


def _write_mixture(ld: _Loaded):
    """input_mixture.wav (css.py:96-97: channel 0, peak-normalised) needs nothing from the GPU: its own task, off the decode's
    critical path (the loop enqueues the session as soon as its planes are in page-locked memory) and off the loop's tail."""
    mixture = ld.planes[0].astype(np.float32) / np.float32(32768.0)
    write_wav(ld.where.directory / 'input_mixture.wav', samps=mixture, sr=ld.sr)


def _write_session(ld: _Loaded, n_out: int, pool: _PinnedPool) -> List[str]:
    """sep_stream{i}.wav of one finished session (css.py:98-106; input_mixture.wav left with the decode), then its buffers go
    back to the pool."""
    names = []
    for i in range(ld.out16.shape[0]):
        path = ld.where.stream_path(i)
        _LOG.info(f"CSS: saving separated wav to {path}")
        write_pcm16_samples(path, ld.out16[i, :n_out], ld.sr)
        names.append(str(path))
    if ld.mixture_written is not None:
        ld.mixture_written.result()          # (it reads channel 0 from the block that goes back to the pool here)
    pool.give(ld.block)
    pool.give(ld.out_block)
    ld.block = ld.planes = ld.out_block = ld.out16 = None
    return names


def css_sessions(out_dir: str, models_dir: str, sessions_df, cfg: CssCfg, fetch_from_cache: bool = False,
                 rank: Optional[int] = None, world: Optional[int] = None, device=None,
                 queue_depth: int = 12, io_threads: int = 4, max_batch_segments: int = 256, linear_mode: str = "exact_f32",
                 separators: Optional[Dict[bool, object]] = None, stats: Optional[dict] = None):
    """Run CSS on the sessions this rank owns; returns a DataFrame with those rows plus
    ``sep_wav_file_names`` -- every row is exactly what ``css_inference`` returns for it (css/css.py:51-107).

    ``queue_depth``: sessions the loop waits for at a time; up to twice as many are in flight (their page-locked buffers
    are alive together);
    ``io_threads``: worker threads that decode the next sessions' wav files and write the finished sessions' files (4: more
    of them only contend with the enqueueing thread for the interpreter -- 8 and 16 measured slower, tools/session_loop_probe.py);
    ``max_batch_segments``: segments per mask-estimator batch of the resident handles (sessions of a queue share batches up
    to it); ``linear_mode``: see ``HipSeparator`` (default: the reference's float32 operand precision); ``separators``:
    models already resident on the GPU, ``{is_mc: HipSeparator}`` -- they are used instead of ``models_dir`` and NOT closed
    (a caller that runs the loop repeatedly, e.g. ``bench.py``'s ``sessions_from_files`` leg); ``stats``: a dict that receives
    where the loop's wall time went (seconds: until the first session was on the queue, inside css_wait, after the last css_wait)."""
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
    resident: Dict[bool, object] = dict(separators or {})
    borrowed = set(resident)
    positions = list(range(rank, len(sessions_df), world))
    rows: Dict[int, object] = {}
    pool = _POOL

    def model_for(is_mc: bool):
        if is_mc not in resident:   # one resident model per kind, loaded on first use
            resident[is_mc], _ = load_css_model(Path(models_dir) / (cfg.checkpoint_mc if is_mc else cfg.checkpoint_sc),
                                                device=device, max_batch_segments=max_batch_segments, linear_mode=linear_mode)
            resident[is_mc].eval()
        return resident[is_mc]

    import time
    t_start = time.perf_counter()
    t_first = t_wait = 0.0
    t_last_wait = t_start
    io = ThreadPoolExecutor(max_workers=max(int(io_threads), 1), thread_name_prefix="css-io")
    try:
        # ---- shortcuts (pass-through, cache hit) are answered at once; the rest is decoded ahead by the worker threads
        work = []
        for pos in positions:
            session = sessions_df.iloc[pos]
            where = _SessionOutput.plan(out_dir, session, cfg, fetch_from_cache)
            if where.shortcut is not None:
                rows[pos] = css_inference(out_dir, models_dir, session, cfg, fetch_from_cache)
            else:
                work.append((pos, session, where))
        ahead = max(2 * int(queue_depth), 2)
        loads: List = []      # loads[k]: the decode of work[k], submitted in order

        def prefetch(upto):
            while len(loads) < min(upto, len(work)):
                j = len(loads)
                loads.append(io.submit(_decode_session, work[j][0], work[j][1], work[j][2], cfg, pool))

        writes = []
        # ---- the queue as a rolling window: sessions are enqueued as their decodes arrive; once 2 x queue_depth are in flight the
        # loop waits for the OLDEST queue_depth (css_wait_sessions: the younger ones keep the device busy), hands them to the file
        # writers and goes on enqueueing.  A full css_wait closes a window when the model kind changes (another handle), a session
        # takes the float path, every 16 x queue_depth sessions (the handle's bookkeeping) and at the end.  In "split_f16" mode a
        # css_wait may repeat a session in float32 from its input buffers, so there nothing is released before a full css_wait.
        rolling = linear_mode == "exact_f32"
        inflight: List[_Loaded] = []      # queued on `cur` since its last full css_wait, oldest first
        released = 0                      # how many of them have been handed to the writers
        cur = None                        # the separator whose queue is open

        def hand_over(upto):
            nonlocal released
            for ld in inflight[released:upto]:
                writes.append((ld.pos, ld.session, io.submit(_write_session, ld, ld.out16.shape[1], pool)))
            released = max(released, upto)

        def close_window():
            nonlocal inflight, released, cur, t_wait, t_last_wait
            if cur is not None and inflight:
                t0 = time.perf_counter()
                if rolling:
                    # drain in small steps: the writers start on the oldest sessions while the youngest are still on the device,
                    # so that what is left to write after the last wait is a few sessions, not the whole window
                    step = max(int(queue_depth) // 3, 1)
                    while len(inflight) - released > step:
                        cur.handle.wait_sessions(released + step)
                        hand_over(released + step)
                cur.handle.wait()
                t_last_wait = time.perf_counter()
                t_wait += t_last_wait - t0
                hand_over(len(inflight))
            inflight, released, cur = [], 0, None

        k = 0
        while k < len(work):
            prefetch(k + ahead)
            ld = loads[k].result()
            prefetch(k + 1 + ahead)
            kind = bool(work[k][1].is_mc)
            if ld.fallback:
                close_window()
                _LOG.info(f"CSS [{rank}/{world}] session {ld.session.session_id} (float path)")
                rows[ld.pos] = css_inference(out_dir, models_dir, ld.session, cfg, fetch_from_cache, separator=model_for(kind))
                k += 1
                continue
            sep = model_for(kind)
            if cur is not sep or len(inflight) >= 16 * queue_depth:
                close_window()
                cur = sep
            sep.to(device)
            h, desc = sep.handle, sep.desc
            run_cfg = make_run_cfg(cfg, ld.sr, len(ld.planes), desc.frame_len, desc.frame_hop)
            n_out = int(_lib.plan(desc, run_cfg, ld.planes[0].shape[0]).n_out)
            S = int(desc.num_spks)
            ld.out_block = pool.take(S * n_out * 2 + 64)
            ld.out16 = ld.out_block[64:64 + S * n_out * 2].view(np.int16).reshape(S, n_out)
            ld.peaks = ld.out_block[:4 * S].view(np.float32)
            _LOG.info(f"CSS [{rank}/{world}] session {ld.session.session_id}")
            h.run_enqueue_pcm16(ld.planes, run_cfg, ld.out16, ld.peaks)
            ld.mixture_written = io.submit(_write_mixture, ld)
            if not t_first:
                t_first = time.perf_counter() - t_start
            inflight.append(ld)
            k += 1
            if rolling and len(inflight) - released >= 2 * queue_depth:
                t0 = time.perf_counter()
                h.wait_sessions(released + queue_depth)
                t_last_wait = time.perf_counter()
                t_wait += t_last_wait - t0
                hand_over(released + queue_depth)
            elif not rolling and len(inflight) >= queue_depth:
                close_window()
        close_window()
        for pos, session, fut in writes:
            result = session.copy()
            result['sep_wav_file_names'] = fut.result()
            rows[pos] = result
    finally:
        io.shutdown(wait=True)
        if stats is not None:
            t_end = time.perf_counter()
            stats.update({"total_s": t_end - t_start, "until_first_enqueue_s": t_first, "in_css_wait_s": t_wait,
                          "after_last_css_wait_s": t_end - t_last_wait, "sessions": len(positions)})
        for kind, sep in resident.items():
            if kind not in borrowed:
                sep.close()
    return pd.DataFrame([rows[p] for p in positions])

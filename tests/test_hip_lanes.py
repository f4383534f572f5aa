"""The mask estimator's schedule must not change a bit of the result: one, two or three lanes (kernel chains on
separate streams), any batch size, odd splits.  Needs an MI355X."""
import os

import numpy as np
import pytest

from conftest import MODES, pkg

pytestmark = pytest.mark.gpu


def _separator(mc_state, lanes, max_batch, mode="exact_f32"):
    sep = pkg("separator").HipSeparator(mc_state[0], None, device=0, max_batch_segments=max_batch, linear_mode=mode)
    sep.handle.set_lanes(lanes)
    assert sep.handle.lanes() == lanes and sep.handle.linear_mode() == mode
    if mode == "exact_f32":   # (by default the exact mode takes a second lane only from 14 000 token rows per lane: 1 = as many as set_lanes gives)
        sep.handle.set_tuning("f32_lane_rows", 1)
    return sep


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seconds", [15.0, 21.2])     # 9 and 13 segments (plus a ragged last one)
def test_lanes_and_batching_are_bit_invariant(mc_state, mix60, seconds, mode):
    if pkg("_lib").load().css_device_count() < 1:
        pytest.fail("no HIP device visible")
    CSS, L = pkg("css"), pkg("_lib")
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    mix = np.ascontiguousarray(mix60[0, :int(seconds * 16000)])
    ref = None
    for lanes, mb in [(1, 64), (2, 64), (3, 64), (2, 6), (2, 7), (4, 5), (2, 12)]:
        sep = _separator(mc_state, lanes, mb, mode)
        try:
            h = sep.handle
            wav = h.run(mix, run_cfg)
            masks = h.read(L.BUF_MASKS)
            perms = h.read(L.BUF_PERMS)
            nseg = h.get_plan().num_segments
        finally:
            sep.close()
        assert np.isfinite(wav).all()
        if ref is None:
            ref = (wav, masks, perms, nseg)
            assert nseg >= 9
        else:
            assert nseg == ref[3]
            assert np.array_equal(masks, ref[1]), (lanes, mb)
            assert np.array_equal(perms, ref[2]), (lanes, mb)
            assert np.array_equal(wav, ref[0]), (lanes, mb)


def test_lanes_in_exact_mode(mc_state, mix60):
    """Same invariance with the exact float32 Linear layers."""
    CSS, L = pkg("css"), pkg("_lib")
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    mix = np.ascontiguousarray(mix60[0, :int(15.0 * 16000)])
    out = []
    for lanes, lane_rows in ((1, 14000), (2, 1), (3, 1), (2, 14000)):
        sep = _separator(mc_state, lanes, 64)
        try:
            sep.handle.set_linear_mode("exact_f32")
            # (by default the exact mode takes a second lane only from 14 000 token rows per lane: 1 = as many as set_lanes gives)
            sep.handle.set_tuning("f32_lane_rows", lane_rows)
            out.append(sep.handle.run(mix, run_cfg))
        finally:
            sep.close()
    for o in out[1:]:
        assert np.array_equal(out[0], o)


def test_split_mode_batch_cap_is_bit_invariant(mc_state, mix60):
    """css_set_tuning split_batch_rows (the split-f16 mode's own bound on an estimator batch) changes the batches, never a bit:
    one pass and a queue of sessions, capped at 4 / 9 segments and uncapped."""
    CSS, L = pkg("css"), pkg("_lib")
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    T = int(run_cfg.c.segment_frames)
    mix = np.ascontiguousarray(mix60[0, :int(21.2 * 16000)])
    pcm = L.pinned_copy(mix)
    ref = None
    for rows in (0, 4 * T, 9 * T + 5, 24576):
        sep = _separator(mc_state, 2, 64, "split_f16")
        try:
            h = sep.handle
            h.set_tuning("split_batch_rows", rows)
            wav = h.run(mix, run_cfg)
            n_out = int(h.get_plan().n_out)
            outs = [L.pinned_empty((3, n_out), np.float32) for _ in range(3)]
            for o in outs:
                h.run_enqueue(pcm, run_cfg, o)
            h.wait()
        finally:
            sep.close()
        if ref is None:
            ref = wav
        assert np.array_equal(wav, ref), rows
        for o in outs:
            assert np.array_equal(o[:, :wav.shape[1]], ref), rows


def test_long_meeting_multi_batch_is_lane_invariant(mc_state):
    """5 minutes (201 segments, four batches of 64): the lane schedule and the batch boundaries leave no trace."""
    CSS, L = pkg("css"), pkg("_lib")
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    mix = np.ascontiguousarray(pkg("synth").synth_meeting(302.0, 7, seed=3)[0])
    out = []
    for lanes, mb in [(1, 256), (2, 64), (3, 50)]:
        sep = _separator(mc_state, lanes, mb)
        try:
            wav = sep.handle.run(mix, run_cfg)
            nseg = sep.handle.get_plan().num_segments
            perms = sep.handle.read(L.BUF_PERMS)
        finally:
            sep.close()
        out.append((wav, perms))
        assert nseg >= 200 and np.isfinite(wav).all()
    for wav, perms in out[1:]:
        assert np.array_equal(perms, out[0][1]) and np.array_equal(wav, out[0][0])


@pytest.mark.parametrize("seg_hop", [(4.0, 2.0), (2.0, 1.0)])
def test_lanes_with_other_segment_lengths(mc_state, mix60, seg_hop):
    """249- and 124-frame segments (8 and 4 attention key tiles, other position-operand tables and q/k fragment
    layouts, conv runs with a ragged tail) through one, three and four lanes: bit for bit the same."""
    CSS, L = pkg("css"), pkg("_lib")
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False, segment_size_sec=seg_hop[0], hop_size_sec=seg_hop[1])
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    mix = np.ascontiguousarray(mix60[0, :int(44.3 * 16000)])
    out = []
    for lanes, mb in [(1, 64), (3, 64), (4, 17)]:
        sep = _separator(mc_state, lanes, mb)
        try:
            wav = sep.handle.run(mix, run_cfg)
            nseg = sep.handle.get_plan().num_segments
            masks = sep.handle.read(L.BUF_MASKS)
        finally:
            sep.close()
        assert nseg >= 16 and np.isfinite(wav).all()
        out.append((wav, masks))
    for wav, masks in out[1:]:
        assert np.array_equal(masks, out[0][1]) and np.array_equal(wav, out[0][0])

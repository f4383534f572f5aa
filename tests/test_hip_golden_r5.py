"""The HIP path (through the C ABI) against the round-5 reference fixtures (tests/golden/gen_golden_r5.py):

  * BASELINE.json configs[1] on a SCREENED recording -- no segment whose IPD features land on the other side of the atan2
    branch cut, a (nearly) full last segment -- so that the WHOLE meeting is compared with the reference: on the reference's
    decisions on 100 % of the frames, free-running on every frame outside the segments with a differing decision AND on the
    whole meeting against what the reference's own rounding-level flips cost it (e2e60_r6_self.npz: the reference at 8 / 4 /
    2 / 1 threads), in both arithmetic modes; every mask value within the survey's 5e-6;
  * a state dict that behaves like a trained model (peaky attention, saturated masks with exact winner-take-all ties, a
    feed-forward operand at ~1e3) through the reference, both modes held to it.

Needs an MI355X."""
import numpy as np
import pytest

from conftest import margins_at, pkg, reference_noise, rel_rms, take_windows
from test_hip_golden_r2 import staged_run
from test_hip_long import _report
from test_oracle_golden_r2 import unpack2, unpack_bits

pytestmark = pytest.mark.gpu

F, S, T = 257, 3, 186
MODES = ("exact_f32", "split_f16")


@pytest.fixture(scope="module")
def L():
    lib = pkg("_lib")
    if lib.load().css_device_count() < 1:
        pytest.fail("no HIP device visible: the parity tests must run on the GPU box")
    return lib


@pytest.fixture(scope="module")
def sep_mc(L, mc_state):
    s = pkg("separator").HipSeparator(mc_state[0], None, device=0, max_batch_segments=64)
    yield s
    s.close()


@pytest.mark.parametrize("mode", MODES)
def test_config2_screened_60s_whole_meeting_vs_reference(L, sep_mc, golden, mode):
    """Every frame of a 61 s / 40-segment meeting against the reference's own run: permutations and both activity maps bit
    for bit, winner-take-all maps, stitched masks, and the waveforms on 100 % of the frames -- on the reference's decisions
    and free-running."""
    CSS, SYN = pkg("css"), pkg("synth")
    import os
    g = golden(os.environ.get("R5_FIXTURE", "e2e60_r5.npz"))   # (gen_golden_r5.py writes candidates; the GPU run picked this one)
    n = int(g["mix_samples"])
    mix = SYN.synth_meeting(n / 16000.0, 7, seed=int(g["mix_seed"]))[:, :n]
    pcm = np.ascontiguousarray(mix[0])
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep_mc.handle
    h.set_linear_mode(mode)
    try:
        free = h.run(pcm, run_cfg)
        nseg = int(g["num_segments"])
        assert free.shape == (S, int(g["wav_len"])) and h.get_plan().num_segments == nseg == 40
        TL = int(h.get_plan().mix_frames)
        assert TL == 39 * 93 + 185        # the 40th segment holds 185 of its 186 frames: as good as full (hazard 8 is about FEW frames)
        m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
        perms = h.read(L.BUF_PERMS)
        act_b = h.read(L.BUF_ACT_B).astype(bool).T
        act_f = h.read(L.BUF_ACT_FINAL).astype(bool).T
        mask_st = h.read(L.BUF_MASK_ST)
        shape = tuple(g["activity_shape"])
        # decisions: exact, everywhere
        assert [tuple(p) for p in perms[1:]] == [tuple(p) for p in g["pit_perm"]]
        assert np.array_equal(act_b, unpack_bits(g["activity_b"], shape))
        assert np.array_equal(act_f, unpack_bits(g["activity_final"], shape))
        # winner-take-all maps and mask values
        wta = unpack2(g["wta_packed"], g["wta_shape"])
        # (a decision = the SET of masks that equal the maximum, mvdr_util.py:53-54: an exact tie in our masks where the
        # reference has a single winner counts as a difference even when argmax agrees)
        ours_win = m == m.max(axis=0, keepdims=True)                                  # [4, F, nseg, T]
        ref_win = np.arange(4)[:, None, None, None] == np.moveaxis(wta, 0, 1)[None]   # wta [nseg, F, T]
        per_seg = [int(np.any(ours_win[:, :, i] != ref_win[:, :, i], axis=0).sum()) for i in range(nseg)]
        md = np.abs(np.stack([np.moveaxis(m[:S, :, i], 0, 2)[::8, ::6] for i in range(nseg)]) - g["masks_spk_dec"])
        ms = np.abs(mask_st.transpose(1, 2, 0)[::32, ::16] - g["mask_stitched"])
        # waveforms, every 64th sample, ALL frames: on the reference's decisions ...
        forced, fperms, _, fact = staged_run(h, L, pcm, run_cfg, wta)
        forced_err = [rel_rms(forced[k, ::64], g["wav_dec64"][k]) for k in range(S)]
        # ... and free-running
        free_err = [rel_rms(free[k, ::64], g["wav_dec64"][k]) for k in range(S)]
        ww = take_windows(free, 4)
        win_err = [rel_rms(ww[k], g["wav_windows"][k]) for k in range(S)]
        _report(f"config2_screened_60s_{mode}" + os.environ.get("R5_TAG", ""), {
            "frames": TL, "segments": nseg, "mix_seed": int(g["mix_seed"]),
            "min_ipd_distance_from_the_branch_cut": float(np.min(g["cut_distance_per_segment"])),
            "reference_wta_margins_below_2e-5": int(g["wta_margin_below_2e-5"]),
            "wta_flips_per_segment": per_seg, "wta_flips": int(sum(per_seg)), "wta_decisions": int(wta.size),
            "fraction_of_frames_compared": 1.0,
            "waveform_rel_rms_on_the_reference_decisions": forced_err, "waveform_rel_rms_free_running": free_err,
            "waveform_rel_rms_free_running_windows": win_err,
            "segment_masks_max_abs": float(md.max()), "stitched_masks_max_abs": float(ms.max()),
            "segment_masks_within_5e-6": round(float((md <= 5e-6).mean()), 6), "stitched_masks_within_5e-6": round(float((ms <= 5e-6).mean()), 6)})
        assert md.max() < 5e-6 and ms.max() < 5e-6, (md.max(), ms.max())      # SURVEY.md 8(d)'s bar as it stands (measured 4.6e-6 / 4.4e-6)
        for k in range(S):
            assert forced_err[k] < 1e-4, (mode, k, forced_err)
        # free-running: one (split_f16) or two (exact_f32) winner-take-all decisions of 1.9 million differ -- segment 4, bin 132,
        # frame 145 and (exact_f32) an exact tie of two of OUR masks at segment 32, bin 41, frame 20.  Round 6 looked at why
        # (tests/golden/gen_golden_r6.py, e2e60_r6_self.npz): the REFERENCE's own top-2 margins there are 3.6e-7 and 4.2e-7,
        # its masks move by 2.1e-6 between its own thread counts, at the first point its 2-thread run ties EXACTLY, its
        # 8-thread winner there is not the float64 network's, and that one flip costs the reference 1.0e-4 whole-meeting
        # against itself.  So: a differing decision is accepted only where the reference's margin is inside twice its own mask
        # noise, and the whole-meeting bar is 1e-4 plus what the reference's own flip costs per differing decision.  A
        # differing decision re-solves its bin's beamformers for its whole segment: the frames of those segments are compared
        # on the reference's decisions above, every other frame free-running at 1e-4.
        g6 = golden("e2e60_r6_self.npz")
        mask_noise, self_dist = reference_noise(g6)
        differ = np.argwhere(np.moveaxis(np.any(ours_win != ref_win, axis=0), 1, 0))     # (segment, bin, frame)
        mg = margins_at(g6, differ) if int(g6["mix_seed"]) == int(g["mix_seed"]) else []
        assert sum(per_seg) <= 2 and all(x <= 2 * mask_noise for x in mg), (per_seg, differ.tolist(), mg)
        clean = np.ones(TL, bool)
        for i, nflip in enumerate(per_seg):
            if nflip:
                clean[max(i * 93 - 2, 0):i * 93 + T + 2] = False
        idx = np.flatnonzero(np.repeat(clean, 4))            # wav_dec64 holds four samples per frame
        idx = idx[idx < g["wav_dec64"].shape[1]]
        clean_err = [rel_rms(free[k, ::64][idx], g["wav_dec64"][k][idx]) for k in range(S)]
        bar = [1e-4 + len(differ) * float(self_dist[k]) for k in range(S)]
        _report(f"config2_screened_60s_{mode}_free_running", {"fraction_of_frames_outside_flipped_segments": round(float(clean.mean()), 4),
                                                              "waveform_rel_rms_there": clean_err, "waveform_rel_rms_whole_meeting": free_err,
                                                              "whole_meeting_bar": bar, "winner_sets_that_differ_at": differ.tolist(),
                                                              "reference_top2_margin_there": mg, "reference_mask_noise": mask_noise,
                                                              "reference_free_running_self_distance": [float(x) for x in self_dist]})
        for k in range(S):
            assert clean_err[k] < 1e-4 and free_err[k] < bar[k], (mode, k, clean_err, free_err, bar)
    finally:
        h.set_linear_mode("exact_f32")


@pytest.fixture(scope="module")
def sep_trained(L, mc_state):
    W = pkg("weights")
    s = pkg("separator").HipSeparator(W.apply_trained_like_recipe(mc_state[0]), None, device=0, max_batch_segments=64)
    yield s
    s.close()


@pytest.mark.parametrize("mode", MODES)
def test_trained_like_weights_vs_reference(L, sep_trained, mix60, golden, mc_state, mode):
    """conformer.py:65-92,302-310 in the regime a trained estimator works in: attention rows with a mean peak of 0.6,
    72 % of the mask values outside [0.01, 0.99] and 10 % at exactly 1 (35 178 exactly tied winner-take-all points), a
    feed-forward hidden operand up to 735 -- both arithmetic modes against the reference's run."""
    CSS = pkg("css")
    g = golden("trained_like_r5.npz")
    n = int(g["mix_samples"])
    pcm = np.ascontiguousarray(mix60[0, :n])
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep_trained.handle
    h.set_linear_mode(mode)
    try:
        wav = h.run(pcm, run_cfg)
        assert h.range_status() == (0, False)          # the ~1e3 operand stays inside the split-f16 range: no float32 repeat
        nseg = int(g["num_segments"])
        assert wav.shape == (S, int(g["wav_len"])) and h.get_plan().num_segments == nseg
        m = np.moveaxis(h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T), (0, 1, 2, 3), (3, 1, 0, 2))   # [nseg, F, T, 4]
        md = np.abs(m[:, ::4, ::3] - g["masks_dec"])
        # the reference's winners (every mask that equals the maximum, mvdr_util.py:53-54) against ours
        ref_win = np.unpackbits(g["wta_all_winners"])[:int(np.prod(g["wta_all_shape"]))].reshape(tuple(g["wta_all_shape"])).astype(bool)
        win = m == m.max(axis=-1, keepdims=True)
        differ = np.any(win != ref_win, axis=-1)
        perms = h.read(L.BUF_PERMS)
        act_b = h.read(L.BUF_ACT_B).astype(bool).T
        act_f = h.read(L.BUF_ACT_FINAL).astype(bool).T
        shape = tuple(g["activity_shape"])
        err = [rel_rms(wav[k, ::64], g["wav_dec64"][k]) for k in range(S)]
        ww = take_windows(wav, 4)
        win_err = [rel_rms(ww[k][:3], g["wav_windows"][k][:3]) for k in range(S)]
        ms = np.abs(h.read(L.BUF_MASK_ST).transpose(1, 2, 0)[::16, ::8] - g["mask_stitched"])
        _report(f"trained_like_{mode}", {
            "segments": nseg, "masks_max_abs": float(md.max()), "masks_within_5e-6": round(float((md <= 5e-6).mean()), 6),
            "stitched_masks_max_abs": float(ms.max()),
            "exact_ones_reference": int((g["masks_dec"] == 1).sum()), "exact_ones_ours": int((m[:, ::4, ::3] == 1).sum()),
            "winner_sets_that_differ": int(differ.sum()), "winner_sets_that_differ_per_segment": differ.sum(axis=(1, 2)).tolist(),
            "exactly_tied_points_reference": int((ref_win.sum(-1) > 1).sum()), "wta_decisions": int(differ.size),
            "waveform_rel_rms_free_running": err, "waveform_rel_rms_windows": win_err})
        assert [tuple(p) for p in perms[1:]] == [tuple(p) for p in g["pit_perm"]]
        assert np.array_equal(act_b, unpack_bits(g["activity_b"], shape))
        assert np.array_equal(act_f, unpack_bits(g["activity_final"], shape))
        # Mask values.  In this regime two correct float32 evaluations are far apart: the reference itself is 2.6 - 2.8e-4
        # (max) / 1.8 - 2.2e-5 (rms) from the same network in float64 (fixture, segments 0 and 6; its distance to the oracle's
        # float32 evaluation is 3e-4) -- the head's gain and the peaky softmax amplify rounding.  So: (a) our distance to
        # float64 must not exceed the reference's own, (b) our distance to the reference is bounded by the sum of the two.
        import css_oracle as O
        p64 = O.ConformerParams(pkg("weights").apply_trained_like_recipe(mc_state[0]), dtype=np.float64)
        for i in (0, 6):
            h.begin(pcm, pcm.shape[0], pcm.shape[1], run_cfg)
            h.stage_stft()
            h.stage_masknet(i, i + 1)
            feat = h.read(L.BUF_FEATURES)[:T, :1799].astype(np.float64)          # the rows OUR embedding consumed
            m64 = np.moveaxis(O.conformer_forward(p64, feat.T, affine_applied=True), 0, 2)           # [F, T, 4]
            d64 = np.abs(m[i].astype(np.float64) - m64)
            ours = float(np.sqrt((d64 ** 2).mean()))
            _report(f"trained_like_{mode}_seg{i}_vs_float64", {"ours_rms": ours, "ours_max": float(d64.max()), "reference_rms": float(g[f"ref_vs_f64_rms_seg{i}"]),
                                                               "reference_max": float(g[f"ref_vs_f64_max_seg{i}"])})
            # measured: exact_f32 1.19 / 0.93 x the reference's own rms distance to float64, split_f16 2.06 / 1.9 x -- in THIS regime
            # (attention logits x 16) the 22-bit operands of the split mode show, where on benign weights they do not
            # (tests/test_hip_precision.py: 0.4 - 0.5 x the exact mode's error)
            assert ours <= (1.5 if mode == "exact_f32" else 2.5) * float(g[f"ref_vs_f64_rms_seg{i}"]), (i, ours)
            assert d64.max() <= 2.5 * float(g[f"ref_vs_f64_max_seg{i}"]), (i, d64.max())
        assert md.max() < 1.5e-3 and float(np.sqrt((md.astype(np.float64) ** 2).mean())) < 5e-5, md.max()
        assert (m[:, ::4, ::3] == 1).sum() >= 0.999 * (g["masks_dec"] == 1).sum()
        assert differ.sum() <= 1e-4 * differ.size, int(differ.sum())      # measured: 22 of 621 426 winner sets
        # waveforms on the REFERENCE's winner sets (ties included: 16 + bitmask, css_mi355.h CSS_BUF_WTA_OVERRIDE): the
        # unconditional statement -- everything behind the estimator, with exactly tied frames in several covariances
        ov = (16 + (ref_win * (1 << np.arange(4))).sum(-1)).astype(np.uint8)          # [nseg, F, T]
        forced, fperms, fb, ff = staged_run(h, L, pcm, run_cfg, ov)
        ferr = [rel_rms(forced[k, ::64], g["wav_dec64"][k]) for k in range(S)]
        _report(f"trained_like_{mode}_forced", {"waveform_rel_rms_on_the_reference_winner_sets": ferr})
        assert [tuple(p) for p in fperms[1:]] == [tuple(p) for p in g["pit_perm"]] and np.array_equal(ff, unpack_bits(g["activity_final"], shape))
        for k in range(S):
            assert ferr[k] < 1e-4, (mode, ferr)
            assert err[k] < 3e-3, (mode, err)   # free-running: 22 differing winner sets move their bins' beamformers (hazard 1)
    finally:
        h.set_linear_mode("exact_f32")


def test_config2_nontrivial_decisions_from_the_estimators_own_masks(L, sep_mc, golden):
    """On the golden weights configs[1] decides nothing: identity permutations, a gate that never closes.  Here the reference
    ran with a separator-protocol wrapper that hands segment i's speaker masks back in a seeded order (all six occur) and
    with the activity threshold inside the range the stitched masks' activity covers (gen_golden_r5c.py).  The same wrapper
    around HipSeparator goes through the drop-in driver's protocol path (css.py:131,199): the masks are the HIP
    estimator's, the shuffle is undone by the HIP permutation solver at every one of the 39 boundaries, the gate opens and
    closes by itself -- permutations, both activity maps and the stitched masks against the reference, waveforms on the
    reference's winner-take-all decisions <= 1e-4 over the whole meeting."""
    import torch
    CSS, SYN = pkg("css"), pkg("synth")
    g = golden("e2e60_r5_decisions.npz")
    n = int(g["mix_samples"])
    mix = SYN.synth_meeting(n / 16000.0, 7, seed=int(g["mix_seed"]))[:, :n]
    orders = g["orders"]
    inner = sep_mc

    class Shuffled:
        """segment i's speaker masks in the order orders[i] (the reference side: gen_golden_r5c.ShuffledOutputs)"""
        training = False

        def __init__(self):
            self.calls = 0

        def to(self, dev):
            return self

        def cpu(self):
            return self

        def stft(self, s):
            return inner.stft(s)

        def istft(self, s):
            return inner.istft(s)

        def separate(self, stft_seg):
            out = inner.separate(stft_seg)
            o = [int(v) for v in orders[self.calls]]
            self.calls += 1
            return {"spk_masks": out["spk_masks"][..., o].contiguous(), "noise_masks": out["noise_masks"]}

    cfg = CSS.CssCfg(activity_th=float(g["activity_th"]), show_progressbar=False)
    sh = Shuffled()
    wavs, side = CSS.separate_and_stitch(mix, sh, 16000, "cuda:0", cfg)
    nseg = int(g["num_segments"])
    assert sh.calls == nseg and len(wavs) == S and len(wavs[0]) == int(g["wav_len"])
    shape = tuple(g["activity_shape"])
    act_b = np.asarray(side["activity_b"].cpu() if hasattr(side["activity_b"], "cpu") else side["activity_b"]).astype(bool)
    act_f = np.asarray(side["activity_final"].cpu() if hasattr(side["activity_final"], "cpu") else side["activity_final"]).astype(bool)[0]
    ref_b, ref_f = unpack_bits(g["activity_b"], shape), unpack_bits(g["activity_final"], shape)
    assert 0.2 < ref_b.mean() < 0.8 and (~ref_f).any()                      # the fixture's gate really works
    assert np.array_equal(act_b, ref_b) and np.array_equal(act_f, ref_f)
    stage = CSS._stage_separator(7, S, "cuda:0")
    perms = stage.handle.read(L.BUF_PERMS)
    assert [tuple(p) for p in perms[1:]] == [tuple(p) for p in g["pit_perm"]]
    assert len({tuple(p) for p in perms[1:]}) == 6
    ms = np.asarray(side["mask_stitched"].cpu() if hasattr(side["mask_stitched"], "cpu") else side["mask_stitched"])[0]   # [F, T, S]
    assert np.abs(ms[::32, ::16] - g["mask_stitched"]).max() < 6e-6
    # waveforms: the protocol path's masks sit in the stage handle; re-run everything behind them on the reference's decisions
    h = stage.handle
    wta = unpack2(g["wta_packed"], g["wta_shape"])
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
    flips = int(sum((np.argmax(m[:, :, i], axis=0) != wta[i]).sum() for i in range(nseg)))
    TL = int(h.get_plan().mix_frames)
    h.write(L.BUF_WTA_OVERRIDE, wta)
    h.stage_mvdr(0, nseg); h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
    forced = h.read(L.BUF_WAV)
    ferr = [rel_rms(forced[k, ::64], g["wav_dec64"][k]) for k in range(S)]
    free_err = [rel_rms(np.asarray(wavs[k])[::64], g["wav_dec64"][k]) for k in range(S)]
    _report("config2_nontrivial_decisions", {"permutations_non_identity": int((np.array(g["pit_perm"]) != np.arange(3)).any(axis=1).sum()),
                                             "distinct_permutations": 6, "activity_b_set_fraction": float(ref_b.mean()),
                                             "gate_closed_frames": int((~ref_f).sum()), "wta_flips": flips,
                                             "waveform_rel_rms_on_the_reference_decisions": ferr, "waveform_rel_rms_free_running": free_err})
    for k in range(S):
        assert ferr[k] < 1e-4, ferr
    assert flips <= 2

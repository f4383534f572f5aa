"""Pins the oracle (oracle/css_oracle.py) against fixtures produced by the REAL reference
(tests/golden/gen_golden.py) and against the reference's own two known-answer tests on this path."""
import numpy as np
import pytest

import css_oracle as O
from conftest import rel_rms, take_windows


@pytest.fixture(scope="module")
def cfg():
    return O.OracleCssCfg(activity_th=0.3)  # configs/inference/inference_v1.yaml:17


# ---------------------------------------------------------------- reference's inline known-answer tests
def test_morphology_known_answer():
    """utils/numpy_utils.py:16-22 (test_morphology) -- the vectors are the reference test's data."""
    arr = np.array([1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0], dtype=bool)
    assert np.all(O.erode(arr, 1) == [1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0])
    assert np.all(O.dilate(arr, 1) == [1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0])


def test_pit_known_answer():
    """css/training/losses.py:109-123 (test_pit_wrapper): a known permutation is recovered with zero loss."""
    rs = np.random.RandomState(43236)
    for _ in range(5):
        targets = rs.rand(100, 257, 4).astype(np.float32)
        p = (3, 0, 2, 1)
        preds = targets[..., p]
        loss, perm, _ = O.pit_perm(preds, targets, "mse")
        assert loss == 0.0
        assert perm == p
        assert np.array_equal(preds, targets[..., list(perm)])


def test_segment_weight_bit_exact(golden):
    g = golden("segment_weight.npz")
    assert np.array_equal(O.calc_segment_weight(186, 9, 18, is_first_seg=True), g["first"])
    assert np.array_equal(O.calc_segment_weight(186, 9, 18), g["mid"])
    assert np.array_equal(O.calc_segment_weight(186, 9, 18, is_last_seg=True), g["last"])


def test_plan_counts():
    """SURVEY.md App. B segment counts measured on the reference."""
    cfg = O.OracleCssCfg()
    for secs, t_long, nseg, last in ((10, 624, 6, 159), (60, 3749, 40, 122), (1800, 112499, 1209, 155)):
        plan = O.make_plan(secs * 16000, 16000, cfg)
        assert (plan.mix_frames, plan.num_segments) == (t_long, nseg)
        assert plan.seg_range(nseg - 1)[2] == last
    plan = O.make_plan(60 * 16000, 16000, cfg)
    assert (plan.segment_frames, plan.hop_frames, plan.m0_frames, plan.m1_frames,
            plan.dilation_frames, plan.erosion_frames) == (186, 93, 9, 18, 24, 12)


# ---------------------------------------------------------------- stage-by-stage (2 segments)
def test_stage_mc(golden, mc_state, mix_stage, cfg):
    g = golden("stage_mc.npz")
    st, _ = mc_state
    params = O.ConformerParams(st)
    fd, td = int(g["fdec"]), int(g["tdec"])
    x = O.stft(mix_stage[0])
    assert rel_rms(x[::16, ::4], g["stft"]) < 2e-6
    taps = {}
    wavs, side = O.separate_and_stitch(mix_stage, params, 16000, cfg, taps=taps)
    plan = side["plan"]
    assert plan.num_segments == 2 and len(wavs[0]) == int(g["wav_len"])
    # features / network taps of segment 0
    seg0 = x[:, :186]
    f0 = O.features(seg0)
    # IPD rows are angles of mean-removed unit phasors: where that vector is short the angle amplifies
    # float32 rounding, so the maximum is loose and the bulk is tight
    fdiff = np.abs(f0[::16, ::4] - g["features_seg0"])
    assert fdiff.max() < 1e-3 and np.percentile(fdiff, 99) < 2e-5
    nt = {}
    m0 = O.conformer_forward(params, f0, taps=nt)
    assert np.max(np.abs(nt["embed"][::4, ::8] - g["embed_seg0"])) < 2e-5
    for l in (0, 8, 17):
        assert np.max(np.abs(nt[f"block{l}"][::4, ::8] - g[f"block{l}_seg0"])) < 5e-5
    assert np.max(np.abs(np.moveaxis(m0[:3], 0, 2)[::fd, ::td] - g["masks_spk"][0])) < 1e-5
    assert np.max(np.abs(np.moveaxis(m0[3:], 0, 2)[::fd, ::td] - g["masks_noise"][0])) < 1e-5
    # WTA decisions, SCM, W, beamformer output of segment 0
    mv = taps["mvdr0"]
    wta_o = np.argmax(mv["wta"], axis=0)
    assert np.mean(wta_o != g["wta_index"][0]) < 1e-4
    assert rel_rms(mv["scm"][:, ::fd], g["scm_seg0"]) < 2e-5
    assert rel_rms(mv["w"], g["w_seg0"]) < 5e-4
    # decisions + stitched outputs
    assert [tuple(p) for p in side["perms"][1:]] == [tuple(p) for p in g["pit_perm"]]
    assert np.max(np.abs(side["mask_stitched"][0, ::fd, ::td] - g["mask_stitched"])) < 1e-5
    assert np.array_equal(side["activity_b"], g["activity_b"])
    assert np.array_equal(side["activity_final"][0], g["activity_final"])
    ww = take_windows(np.stack(wavs), 4)
    for k in range(3):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4


# ---------------------------------------------------------------- end to end, 20 s, 13 segments
@pytest.fixture(scope="module")
def e2e_run(golden, mc_state, mix60, cfg):
    st, _ = mc_state
    params = O.ConformerParams(st)
    mix = mix60[:, :20 * 16000]
    g = golden("e2e_mc.npz")
    store = {}

    def sep(i, seg):
        store[i] = O.separate(params, seg)
        return store[i]

    wavs, side = O.separate_and_stitch(mix, params, 16000, cfg, separate_fn=sep)
    return mix, params, g, store, wavs, side


def _unpack(bits, shape):
    return np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape).astype(bool)


def test_e2e_mc_decisions(e2e_run):
    mix, params, g, store, wavs, side = e2e_run
    assert side["plan"].num_segments == int(g["num_segments"])
    assert len(wavs[0]) == int(g["wav_len"])
    assert [tuple(p) for p in side["perms"][1:]] == [tuple(p) for p in g["pit_perm"]]
    shape = tuple(g["activity_shape"])
    assert np.array_equal(side["activity_b"], _unpack(g["activity_b"], shape))
    assert np.array_equal(side["activity_final"][0], _unpack(g["activity_final"], shape))
    assert np.max(np.abs(side["activity"] - g["activity_values"])) < 1e-5
    assert np.max(np.abs(side["mask_stitched"][0, ::16, ::8] - g["mask_stitched"])) < 2e-5
    # winner-take-all maps agree except at float32-rounding-level ties
    flips = sum(int((np.argmax(np.concatenate(store[i], -1), -1) != g["wta_index"][i]).sum()) for i in store)
    assert flips <= 1e-5 * g["wta_index"].size + 3


def test_e2e_mc_waveform_on_reference_decisions(e2e_run, cfg):
    """<= 1e-4 relative RMS against the reference's waveforms when the (discontinuous) WTA decisions are
    taken from the reference; without the override one flipped TF point moves a stream by ~1e-4."""
    mix, params, g, store, wavs, side = e2e_run
    w2, _ = O.separate_and_stitch(mix, params, 16000, cfg, separate_fn=lambda i, seg: store[i],
                                  wta_override=g["wta_index"])
    ww = take_windows(np.stack(w2))
    for k in range(3):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4
        assert rel_rms(np.stack(w2)[k, ::64], g["wav_dec"][k]) < 1e-4
    # free-running decisions: bounded by the handful of flips
    ww = take_windows(np.stack(wavs))
    for k in range(3):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-3


def test_e2e_mc_forced_permutations(e2e_run, cfg):
    """Variant (a) of SURVEY.md App. C.6: call i rotates its speaker channels by i mod 3."""
    mix, params, g, store, wavs, side = e2e_run

    def rot(i, seg):
        spk, noi = store[i]
        return np.roll(spk, i % 3, axis=-1), noi

    idx = g["wta_index"].copy()
    for i in range(idx.shape[0]):
        spk_sel = idx[i] < 3
        idx[i][spk_sel] = (idx[i][spk_sel] + (i % 3)) % 3
    w2, s2 = O.separate_and_stitch(mix, params, 16000, cfg, separate_fn=rot, wta_override=idx)
    assert [tuple(p) for p in s2["perms"][1:]] == [tuple(p) for p in g["rot_pit_perm"]]
    assert len(set(tuple(p) for p in g["rot_pit_perm"])) == 3
    ww = take_windows(np.stack(w2))
    for k in range(3):
        assert rel_rms(ww[k], g["rot_wav_windows"][k]) < 1e-4


def test_e2e_mc_activity_gating(e2e_run):
    """Variant (b): a threshold placed in a gap of the activity values so that gating toggles."""
    mix, params, g, store, wavs, side = e2e_run
    cfg_b = O.OracleCssCfg(activity_th=float(g["gate_th"]))
    w2, s2 = O.separate_and_stitch(mix, params, 16000, cfg_b, separate_fn=lambda i, seg: store[i],
                                   wta_override=g["wta_index"])
    shape = tuple(g["activity_shape"])
    assert np.array_equal(s2["activity_b"], _unpack(g["gate_activity_b"], shape))
    assert np.array_equal(s2["activity_final"][0], _unpack(g["gate_activity_final"], shape))
    assert 0.0 < s2["activity_final"].mean() < 1.0
    ww = take_windows(np.stack(w2))
    for k in range(3):
        assert rel_rms(ww[k], g["gate_wav_windows"][k]) < 1e-4


def test_e2e_sc(golden, sc_state, mix60, cfg):
    g = golden("e2e_sc.npz")
    st, _ = sc_state
    params = O.ConformerParams(st)
    mix = mix60[:, :12 * 16000, :1].copy()
    wavs, side = O.separate_and_stitch(mix, params, 16000, cfg)
    assert [tuple(p) for p in side["perms"][1:]] == [tuple(p) for p in g["pit_perm"]]
    assert np.array_equal(side["activity_final"][0], _unpack(g["activity_final"], tuple(g["activity_shape"])))
    ww = take_windows(np.stack(wavs))
    for k in range(3):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-5


def test_single_segment_input_raises():
    """Inputs <= 3.0 s give one segment whose right edge has zero weight: the reference asserts
    (css.py:297) and so must every implementation."""
    mix = np.zeros((1, 48000, 1), np.float32)
    with pytest.raises(AssertionError, match="zero weights"):
        O.separate_and_stitch(mix, None, 16000, O.OracleCssCfg(),
                              separate_fn=lambda i, seg: (np.full((257, 186, 3), 0.5, np.float32),
                                                          np.full((257, 186, 1), 0.5, np.float32)))

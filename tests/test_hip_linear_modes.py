"""The two arithmetic modes of the Conformer's Linear layers (include/css_mi355.h css_set_linear_mode):

  exact_f32 (default)  float32 operands on the float32 matrix instruction: the reference's own operand precision
  split_f16 (opt-in)   operands as hi + 2^-11 lo float16 pairs, three f16 MFMAs per product, float32 accumulation

Both must meet the SAME parity bars against the oracle and the reference fixtures (the rest of the `-m gpu` suite
runs in the default mode unless a test names the other); this module runs the key checks in both modes on one handle and
bounds the distance between the two.  Needs an MI355X.
"""
import numpy as np
import pytest

import css_oracle as O
from conftest import pkg, rel_rms, take_windows

pytestmark = pytest.mark.gpu

F, T, S = 257, 186, 3


@pytest.fixture(scope="module")
def L():
    lib = pkg("_lib")
    if lib.load().css_device_count() < 1:
        pytest.fail("no HIP device visible: the parity tests must run on the GPU box")
    return lib


@pytest.fixture(scope="module")
def sep(L, mc_state):
    st, _ = mc_state
    s = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=64)
    yield s
    s.close()


def _masks(h, L, nseg):
    return h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)


def test_default_mode_and_switching(L, sep, mc_state):
    h = sep.handle
    # a new handle computes in the reference's own operand precision (css_create; conformer.py:137-150 is float32)
    assert h.linear_mode() == "exact_f32"
    h.set_linear_mode("split_f16")
    assert h.linear_mode() == "split_f16"
    h.set_linear_mode("exact_f32")
    assert h.linear_mode() == "exact_f32"
    with pytest.raises(Exception):
        L.check(h.h, h.lib.css_set_linear_mode(h.h, 7))
    # the opt-in at the separator level (what css_inference / separate_and_stitch users pass)
    SEP = pkg("separator")
    s2 = SEP.HipSeparator(mc_state[0], None, device=0, linear_mode="split_f16")
    try:
        assert s2.handle.linear_mode() == "split_f16"
    finally:
        s2.close()
    with pytest.raises(ValueError):
        SEP.HipSeparator(mc_state[0], None, device=0, linear_mode="bf16")


def test_masks_and_hidden_in_both_modes(L, sep, mc_state, mix_stage, golden):
    """Encoder output and masks of segment 0 vs the oracle and vs the reference's masks, same tolerances in both
    modes; split vs exact differ by float32-rounding-level noise only."""
    CSS = pkg("css")
    st, _ = mc_state
    params = O.ConformerParams(st)
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    h = sep.handle
    xo = O.stft(mix_stage[0])
    taps = {}
    om = O.conformer_forward(params, O.features(xo[:, :T]), taps=taps)
    g = golden("stage_mc.npz")
    fd, td = int(g["fdec"]), int(g["tdec"])
    got = {}
    for mode in ("split_f16", "exact_f32", "split_f16", "exact_f32"):   # switching back and forth is part of the test (ends in the default)
        h.set_linear_mode(mode)
        wav = h.run(mix_stage[0], run_cfg)
        nseg = h.get_plan().num_segments
        m = _masks(h, L, nseg)
        hid = h.read(L.BUF_HIDDEN)
        feat = h.read(L.BUF_FEATURES)
        assert (feat[:, 1799:] == 0).all()
        assert np.abs(hid[:T] - taps["block17"]).max() < 1e-4, mode
        assert np.abs(m[:, :, 0, :] - om).max() < 1.5e-5, mode
        for i in range(2):
            spk = np.moveaxis(m[:S, :, i], 0, 2)
            assert np.abs(spk[::fd, ::td] - g["masks_spk"][i]).max() < 1.5e-5, mode
        ww = take_windows(wav, 4)
        for k in range(S):
            assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4, mode
        if mode in got:   # a mode reproduces itself bit for bit after a round trip through the other one
            assert np.array_equal(got[mode][0], m) and np.array_equal(got[mode][1], wav), mode
        got[mode] = (m.copy(), wav.copy(), feat.copy())
    d = np.abs(got["split_f16"][0] - got["exact_f32"][0])
    assert d.max() < 2e-5 and np.sqrt((d ** 2).mean()) < 2e-6
    # the feature rows are read back from the split operand buffer as hi + lo * 2^-11; the two modes also run the
    # analysis transform in their own arithmetic, so the features differ by float32 rounding of the STFT, which the
    # IPD angles of short phasors amplify (DESIGN.md "Numerical hazards" 3): tight bulk, loose maximum
    fs, fe = got["split_f16"][2], got["exact_f32"][2]
    df = np.abs(fs - fe)
    assert df.max() < 1e-2 and np.percentile(df, 99) < 1e-4 and (df > 1).sum() == 0
    for k in range(S):
        assert rel_rms(got["split_f16"][1][k], got["exact_f32"][1][k]) < 1e-4


def test_e2e_waveform_vs_reference_split_mode(L, sep, mix60, golden):
    """The 20 s end-to-end fixture in split_f16 mode (test_hip_parity.py runs it in the default exact_f32 mode)."""
    CSS = pkg("css")
    g = golden("e2e_mc.npz")
    mix = mix60[:, :20 * 16000]
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    h = sep.handle
    h.set_linear_mode("split_f16")
    try:
        nseg = int(g["num_segments"])
        h.begin(mix[0], mix.shape[1], 7, run_cfg)
        TL = h.get_plan().mix_frames
        h.write(L.BUF_WTA_OVERRIDE, g["wta_index"])
        h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
        h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
        w2 = h.read(L.BUF_WAV)
        ww = take_windows(w2)
        for k in range(S):
            assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4
            assert rel_rms(w2[k, ::64], g["wav_dec"][k]) < 1e-4
        assert [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["pit_perm"]]
        m = _masks(h, L, nseg)
        flips = int((np.argmax(m, axis=0).transpose(1, 0, 2) != g["wta_index"]).sum())
        assert flips <= 1e-5 * g["wta_index"].size + 3
    finally:
        h.set_linear_mode("exact_f32")


def test_exact_mode_is_bit_identical_on_either_float32_gemm(L, sep, mix_stage):
    """Round 5's exact float32 GEMM (gemm_f32.hip: four independent blocks per CU, persistent, balanced tile heights) must
    leave every bit of the exact mode where round 4's kernel (gemm.hip) put it: masks, hidden states and waveforms of a
    whole pass, for the balanced plan and for every forced tile height (css_set_tuning CSS_TUNE_F32_GEMM)."""
    CSS = pkg("css")
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep.handle
    h.set_linear_mode("exact_f32")
    try:
        ref = None
        for tune in (1, 0, 2, 3, 4, 5, 6):   # 1: gemm.hip; 0: gemm_f32.hip; 2..5: tiles of at most 32 .. 128 rows; 6: weights through LDS
            h.set_tuning("f32_gemm", tune)
            wav = h.run(mix_stage[0], run_cfg)
            got = (_masks(h, L, h.get_plan().num_segments).copy(), h.read(L.BUF_HIDDEN).copy(), wav.copy())
            if ref is None:
                ref = got
            else:
                for a, b in zip(ref, got):
                    assert np.array_equal(a, b), tune
    finally:
        h.set_tuning("f32_gemm", 0)

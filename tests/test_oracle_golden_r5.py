"""The oracle against the round-5 fixtures (tests/golden/gen_golden_r5.py): the trained-like state dict (peaky attention,
saturated masks with exact ties, a feed-forward operand at ~1e3) and the first segments of the screened 61 s meeting.
CPU only; sized for the few-minute CPU suite (the full meetings are the GPU suite's, tests/test_hip_golden_r5.py)."""
import numpy as np
import pytest

import css_oracle as O
from conftest import pkg, rel_rms
from test_oracle_golden_r2 import unpack2

F, T, S = 257, 186, 3


def test_trained_like_recipe_is_what_the_fixture_says(mc_state, golden):
    """the recipe is deterministic arithmetic on the golden weights; its feed-forward rescaling is exact (a power of two)"""
    W = pkg("weights")
    st = W.apply_trained_like_recipe(mc_state[0])
    p = W.PREFIX + f"conformer.encoders.{W.TRAINED_LIKE_FF_BLOCK}.feed_forward_in.net."
    assert np.array_equal(st[p + "0.weight"], mc_state[0][p + "0.weight"] * np.float32(256))
    assert np.array_equal(st[p + "3.weight"] * np.float32(256), mc_state[0][p + "3.weight"])
    q = W.PREFIX + "conformer.encoders.3.self_attn.linear_q.weight"
    assert np.array_equal(st[q], mc_state[0][q] * np.float32(4))
    assert np.array_equal(st[W.PREFIX + "linear.weight"], mc_state[0][W.PREFIX + "linear.weight"] * np.float32(8))
    g = golden("trained_like_r5.npz")
    assert int(g["num_segments"]) == 13 and float((g["masks_dec"] == 1).mean()) > 0.05


def test_trained_like_segment_oracle_vs_reference(mc_state, mix60, golden):
    """Segment 0 of the fixture through the oracle's float32 network: in this regime two correct float32 evaluations are
    ~3e-4 apart (the fixture holds the reference's own distance to float64: 2.6e-4 max / 1.8e-5 rms), so the oracle is held
    to the reference at that level, and to float64 at the reference's own distance."""
    W = pkg("weights")
    g = golden("trained_like_r5.npz")
    st = W.apply_trained_like_recipe(mc_state[0])
    x = O.stft(np.ascontiguousarray(mix60[0, :int(g["mix_samples"])]))
    feat = O.features(x[:, :T])
    m32 = np.moveaxis(O.conformer_forward(O.ConformerParams(st), feat), 0, 2)                       # [F, T, 4]
    ref = g["masks_dec"][0]
    d = np.abs(m32[::4, ::3] - ref)
    assert d.max() < 1e-3 and float(np.sqrt((d.astype(np.float64) ** 2).mean())) < 5e-5, d.max()
    d64 = np.abs(m32[::4, ::3].astype(np.float64) - g["masks_f64_seg0"])
    assert float(np.sqrt((d64 ** 2).mean())) < 1.5 * float(g["ref_vs_f64_rms_seg0"])
    # exact ones and exactly tied winners are where the reference has them (saturated sigmoids round to 1.0f alike)
    ones_ref, ones = (ref == 1), (m32[::4, ::3] == 1)
    assert (ones != ones_ref).mean() < 1e-3
    win = np.unpackbits(g["wta_all_winners"])[:int(np.prod(g["wta_all_shape"]))].reshape(tuple(g["wta_all_shape"])).astype(bool)[0]
    mine = m32 == m32.max(axis=-1, keepdims=True)
    assert np.any(mine != win, axis=-1).mean() < 2e-4


def test_screened_meeting_first_segments_oracle_vs_reference(mc_state, golden):
    g = golden("e2e60_r5.npz")
    n = int(g["mix_samples"])
    mix = pkg("synth").synth_meeting(n / 16000.0, 7, seed=int(g["mix_seed"]))[:, :n]
    assert float(np.min(g["cut_distance_per_segment"])) >= 4e-7
    x = O.stft(np.ascontiguousarray(mix[0]))
    assert x.shape[1] == 39 * 93 + 185
    params = O.ConformerParams(mc_state[0])
    wta = unpack2(g["wta_packed"], g["wta_shape"])
    for i in (0, 21):
        m = O.conformer_forward(params, O.features(x[:, i * 93:i * 93 + T]))                        # [4, F, T]
        spk = np.moveaxis(m[:S], 0, 2)
        assert np.abs(spk[::8, ::6] - g["masks_spk_dec"][i]).max() < 6e-6
        assert int((np.argmax(m, axis=0) != wta[i]).sum()) <= 2

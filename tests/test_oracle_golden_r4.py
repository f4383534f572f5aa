"""The oracle against the round-4 fixtures: the reference in the regime a trained separator works in (ideal ratio masks of a
turn-taking conversation: saturated masks, silent talkers, tied winners, shuffled speaker order, a gate that toggles by
itself; tests/golden/gen_golden_r4.py, tests/irm_separator.py).  CPU only."""
import numpy as np
import pytest

import css_oracle as O
import irm_separator as IRM
from conftest import pkg, rel_rms


@pytest.fixture(scope="module")
def conversation(golden):
    g = golden("realistic_r4.npz")
    mix, images = pkg("synth").synth_conversation(float(g["mix_seconds"]), 7, seed=int(g["mix_seed"]), return_sources=True)
    masks = IRM.IdealMasks(images)
    return g, mix, masks


def test_the_masks_are_the_generators_and_what_they_exercise(conversation):
    g, _, masks = conversation
    nseg = int(g["num_segments"])
    assert masks.sha256(nseg) == str(g["masks_sha256"])
    st = masks.statistics(nseg)
    assert st["mask_values_exactly_0"] > 0.3 * 4 * st["tf_points"] and st["mask_values_exactly_1"] > 0.05 * 4 * st["tf_points"]
    assert st["tf_points_with_tied_winners"] > 10000
    assert min(st["segments_with_talker_all_zero"]) >= 5              # every talker is silent through >= 5 whole segments


def test_realistic_mc_oracle_vs_reference(conversation):
    g, mix, masks = conversation
    ow, side = O.separate_and_stitch(mix, None, 16000, O.OracleCssCfg(activity_th=0.3),
                                     separate_fn=lambda i, seg: masks.segment(i), mvdr_cplx=np.complex128)
    perms = np.array(side["perms"][1:], np.int32)
    assert np.array_equal(perms, g["mc_pit_perm"])                    # incl. 9 boundaries whose optimum is exactly tied
    assert len({tuple(p) for p in perms}) == 6
    shape = tuple(g["mc_activity_shape"])
    assert np.array_equal(side["activity_b"], IRM.unpack_bits(g["mc_activity_b"], shape))
    act_f = IRM.unpack_bits(g["mc_activity_final"], shape)
    assert np.array_equal(side["activity_final"][0], act_f)
    assert all(int(np.abs(np.diff(act_f[:, k].astype(int))).sum()) >= 6 for k in range(3))   # the gate toggles by itself
    dec, n_out = int(g["dec"]), int(g["mc_wav_len"])
    assert len(ow[0]) == n_out
    # the reference's own code in complex128: every sample of every stream
    for k in range(3):
        assert rel_rms(ow[k][::dec], g["mc_wav_c128"][k]) < 5e-6, k
    # the reference as it is (complex64), wherever it reproduces itself
    ok_f, ok_s = IRM.reproduced_samples(g["mc_c64_vs_c128_per_segment"], g["mc_pit_perm"], float(g["repro_tau"]), shape[0], n_out, dec)
    assert 0.75 < ok_f.mean() < 0.9          # 100 of the 120 (segment, stream) pairs; the other 20 are >= 2.3e-2 apart
    for k in range(3):
        assert rel_rms(ow[k][::dec][ok_s[k]], g["mc_wav_c64"][k][ok_s[k]]) < 1e-4, k
    assert np.abs(side["mask_stitched"][0, ::16, ::8] - g["mc_mask_stitched"]).max() < 1e-6


def test_realistic_sc_oracle_vs_reference(conversation):
    g, mix, masks = conversation
    ow, side = O.separate_and_stitch(np.ascontiguousarray(mix[:, :, :1]), None, 16000, O.OracleCssCfg(activity_th=0.3),
                                     separate_fn=lambda i, seg: masks.segment(i))
    assert np.array_equal(np.array(side["perms"][1:], np.int32), g["sc_pit_perm"])
    shape = tuple(g["sc_activity_shape"])
    assert np.array_equal(side["activity_final"][0], IRM.unpack_bits(g["sc_activity_final"], shape))
    for k in range(3):
        assert rel_rms(ow[k][::int(g["dec"])], g["sc_wav"][k]) < 2e-6, k

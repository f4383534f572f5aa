"""e2e60_r6_self.npz (tests/golden/gen_golden_r6.py): the REFERENCE against itself on the screened configs[1] meeting at 8 / 4 / 2 /
1 torch threads.  The facts the free-running bars of tests/test_hip_headline.py and tests/test_hip_golden_r5.py rest on, checked
on the CPU: the reference's masks move by ~2e-6 between thread counts, ONE of its winner-take-all decisions (top-2 margin
3.6e-7) flips at 2 threads, and that flip alone puts its whole-meeting free-running distance to itself at ~1e-4 -- SURVEY 8(d)'s
1e-4 bar is not met by the reference against itself on this input (css/css.py:110-338, mvdr_util.py:50-55)."""
import numpy as np

from conftest import margins_at, reference_noise


def test_reference_self_distance(golden):
    g6 = golden("e2e60_r6_self.npz")
    assert bool(g6["base_run_equals_e2e60_r5"])            # the 8-thread run IS the run e2e60_r5.npz holds (waveforms bit for bit)
    mask_noise, self_dist = reference_noise(g6)
    assert 1e-6 < mask_noise < 4e-6
    assert 9e-5 < float(self_dist.max()) < 2e-4           # the reference against itself: one flip, ~1e-4 whole-meeting
    d2 = g6["wta_differ_t2"]
    assert d2.shape == (1, 3) and [int(v) for v in d2[0]] == [4, 132, 145] and margins_at(g6, d2)[0] < 1e-6
    for t in (4, 1):                                       # no flip: the distance is the float32 noise floor
        assert g6[f"wta_differ_t{t}"].shape[0] == 0 and float(g6[f"wav_rel_rms_dec64_t{t}"].max()) < 2e-5


def test_near_tie_list_and_float64(golden):
    """143 of the reference's 1.9 million decisions have a top-2 margin below 2e-5; at ONE of them its float32 decision is not
    the float64 network's (oracle, float64 parameters) -- the very point where its 2-thread run ties exactly."""
    g6 = golden("e2e60_r6_self.npz")
    near, mg = g6["near_points"], g6["near_margin"]
    assert near.shape == (len(mg), 3) and 100 < len(mg) < 200 and float(mg.max()) < 2e-5
    m8, m64 = g6["near_masks_t8"], g6["near_masks_f64"]
    srt = np.sort(m8, axis=-1)
    assert np.allclose(srt[:, -1] - srt[:, -2], mg, atol=1e-9)
    off = np.flatnonzero(np.argmax(m8, -1) != np.argmax(m64, -1))
    assert len(off) == 1 and [int(v) for v in near[off[0]]] == [4, 132, 145]
    assert float(np.abs(m8 - m64).max()) < 3e-6            # the reference's float32 masks sit within ~1e-6 of float64 here
    i = int(np.flatnonzero((near == [4, 132, 145]).all(axis=1))[0])
    top2 = np.sort(g6["near_masks_t2"][i])[-2:]
    assert top2[0] == top2[1]                               # an EXACT tie in the reference's own 2-thread run


def test_configs3_fixture_is_the_planned_meeting(golden):
    """e2e1800_r6.npz (tests/golden/gen_golden_r6b.py): the reference's own run of configs[3].  Its shape is the oracle's plan for
    28.8 M samples (css/css.py:144-171: 1209 segments, 112 499 frames, a ragged last segment), and what it says about the
    reference on this meeting -- thousands of winner-take-all decisions inside 1e-5, most segments with an IPD feature within
    1e-5 of the atan2 branch cut -- is what tests/test_hip_long.py::test_whole_meeting_vs_the_reference leans on."""
    import css_oracle as O
    g = golden("e2e1800_r6.npz")
    plan = O.make_plan(28_800_000, 16000, O.OracleCssCfg(activity_th=0.3))
    assert (plan.num_segments, plan.mix_frames) == (1209, 112_499) == (int(g["num_segments"]), int(g["activity_shape"][0]))
    assert int(g["wav_len"]) == (112_499 - 1) * 256 + 512 and g["wav_dec128"].shape == (3, -(-int(g["wav_len"]) // 128))
    assert g["pit_perm"].shape == (1208, 3) and g["masks_grid"].shape == (1209, 9, 8, 4)
    assert np.array_equal(np.sort(g["pit_perm"], axis=1), np.tile(np.arange(3), (1208, 1)))
    assert float(g["masks_grid"].min()) > 0.0 and float(g["masks_grid"].max()) < 1.0
    assert int(g["wta_margin_below_1e-5_per_segment"].sum()) > 2000                  # 2 415 of 57.8 M decisions
    cut = g["cut_distance_per_segment"]
    assert cut.shape == (1209,) and float(np.median(cut)) < 1e-5                     # hazard 7 is the rule on this meeting, not the exception


def test_nulls_fixture_and_the_oracle_on_a_muted_array(golden, mc_state):
    """nulls_r6.npz (tests/golden/gen_golden_r6c.py): the reference on digital zeros.  A muted array gives exact zeros and identity
    permutations (mvdr_util.py:58-75's 1e-15 diagonal loading keeps the solve defined) -- and so does the oracle; the gap
    recording is exactly zero where every contributing frame is, and its beamformer is undefined (complex64 noise) around the
    gap: fewer than two thirds of its seconds are comparable at 1e-4."""
    import css_oracle as O
    g = golden("nulls_r6.npz")
    assert float(np.abs(g["all_zero_wav_dec16"]).max()) == 0.0 and (g["all_zero_pit_perm"] == np.arange(3)).all()
    w, side = O.separate_and_stitch(np.zeros((1, 8 * 16000, 7), np.float32), O.ConformerParams(mc_state[0]), 16000, O.OracleCssCfg(activity_th=0.3))
    assert all(float(np.abs(x).max()) == 0.0 for x in w) and [tuple(p) for p in side["perms"][1:]] == [tuple(p) for p in g["all_zero_pit_perm"]]
    z = np.unpackbits(g["gap_wav_is_zero_dec16"])[:g["gap_wav_dec16"].size].reshape(g["gap_wav_dec16"].shape).astype(bool)
    t = np.flatnonzero(z.all(axis=0))[1:]        # (sample 0 is zero in every output: the synthesis window starts at 0)
    assert t.size > 6000 and t[0] * 16 >= 8 * 16000 and t[-1] * 16 < int(15.5 * 16000)        # inside the gap, most of it
    defined = (g["gap_oracle_c64_vs_c128_per_second"].max(axis=1) < 1e-4) & (g["gap_oracle_c64_vs_reference_per_second"].max(axis=1) < 1e-4)
    assert 8 <= int(defined.sum()) < 16

"""The HIP path (through the C ABI) on recordings with DIGITAL ZEROS, against the reference's own runs of them
(tests/golden/gen_golden_r6c.py, nulls_r6.npz): a muted array, a 7.5 s gap of exact zeros in the middle of a meeting (three
segments entirely silent, four partly), a dead microphone (every covariance rank deficient; mvdr_util.py:58-75's 1e-15 diagonal
loading is all that keeps the solve defined).  Asserted in both arithmetic modes: finite output; the decisions of the reference
(permutations, both activity maps); exact zeros exactly where the reference has exact zeros; the waveforms within 1e-4 on
every second where the answer is defined -- where the oracle's complex64 and complex128 beamformers agree with each other and with
the reference (a talker silent in a segment makes the reference's complex64 solve return noise: DESIGN.md, hazards); and the
queue gives the synchronous call's bits.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import pkg, rel_rms
from test_hip_long import _report
from test_oracle_golden_r2 import unpack_bits

pytestmark = pytest.mark.gpu
S = 3


def _cases():
    base = pkg("synth").synth_meeting(24.0, 7, seed=1)[0]
    gap = base.copy()
    gap[8 * 16000:int(15.5 * 16000)] = 0.0
    dead = base.copy()
    dead[:, 3] = 0.0
    return {"all_zero": np.zeros((8 * 16000, 7), np.float32), "gap": gap, "dead_mic": dead}


@pytest.mark.parametrize("mode", ["exact_f32", "split_f16"])
def test_digital_zeros_vs_the_reference(mc_state, golden, mode):
    CSS, L = pkg("css"), pkg("_lib")
    g = golden("nulls_r6.npz")
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    sep = pkg("separator").HipSeparator(mc_state[0], None, device=0, max_batch_segments=64, linear_mode=mode)
    try:
        h = sep.handle
        for name, x in _cases().items():
            assert x.shape[0] == int(g[f"{name}_samples"])
            pcm = L.pinned_copy(np.ascontiguousarray(x))
            w = h.run(pcm, run_cfg).copy()
            assert w.shape == (S, int(g[f"{name}_wav_len"])) and np.isfinite(w).all()
            shape = tuple(g[f"{name}_activity_shape"])
            assert [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g[f"{name}_pit_perm"]]
            assert np.array_equal(h.read(L.BUF_ACT_B).astype(bool).T, unpack_bits(g[f"{name}_activity_b"], shape))
            assert np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, unpack_bits(g[f"{name}_activity_final"], shape))
            ref = g[f"{name}_wav_dec16"]
            got = w[:, ::16]
            ref_zero = np.unpackbits(g[f"{name}_wav_is_zero_dec16"])[:ref.size].reshape(ref.shape).astype(bool)
            # exact zeros: the separated streams of frames whose samples are all zero are zero, not "small" (and nothing else is)
            assert np.array_equal(got == 0, ref_zero), (name, int((got == 0).sum()), int(ref_zero.sum()))
            c64_c128 = g[f"{name}_oracle_c64_vs_c128_per_second"].max(axis=1)
            c64_ref = g[f"{name}_oracle_c64_vs_reference_per_second"].max(axis=1)
            defined = (c64_c128 < 1e-4) & (c64_ref < 1e-4)
            scale = np.maximum(g[f"{name}_wav_rms"], 1e-30)
            errs = []
            for s_ in np.flatnonzero(defined):
                lo, hi = s_ * 1000, min((s_ + 1) * 1000, ref.shape[1])
                errs.append(float((np.sqrt(((got[:, lo:hi].astype(np.float64) - ref[:, lo:hi]) ** 2).mean(axis=1)) / scale).max()))
            _report(f"nulls_{name}_{mode}", {"seconds": int(len(defined)), "seconds_where_the_answer_is_defined": int(defined.sum()),
                                           "worst_second_there_rel_to_stream_rms": max(errs) if errs else 0.0,
                                           "whole_recording_rel_rms": [rel_rms(got[k], ref[k]) if scale[k] > 1e-20 else 0.0 for k in range(S)],
                                           "exactly_zero_samples_dec16": int(ref_zero.sum())})
            assert int(defined.sum()) >= {"all_zero": 8, "gap": 8, "dead_mic": 12}[name]
            bad = [int(s_) for s_, e_ in zip(np.flatnonzero(defined), errs) if e_ >= 1e-4]
            if bad:
                # Seconds beyond 1e-4 are accepted only as ONE segment whose IPD features cross the atan2 branch cut (DESIGN.md hazard 7,
                # section 4: 3 % of the segments of a long meeting; with a dead microphone its phase is exactly 0 and the pair's
                # difference sits on the cut more often): the segment's masks then differ from the oracle's own by percents, every
                # other segment's by < 2e-5, and the reference does the same to itself on its other conv backend.
                import css_oracle as O
                i = int(min(bad) / 1.5)
                assert max(bad) + 1 <= 1.5 * i + 3.0 + 1e-9, (name, mode, bad)
                m = h.read(L.BUF_MASKS).reshape(S + 1, 257, -1, 186)
                X = O.stft(x)
                params = O.ConformerParams(mc_state[0])
                diffs = {}
                for j in (i - 1, i, i + 1):
                    feat = O.features(X[:, j * 93:j * 93 + 186])
                    diffs[j] = (float(np.abs(m[:, :, j, :] - O.conformer_forward(params, feat)).max()), float(np.abs(np.abs(feat[257:]) - np.pi).min()))
                _report(f"nulls_{name}_{mode}_branch_cut_segment", {"seconds_beyond_1e-4": bad, "segment": i,
                                                                     "masks_vs_oracle_max_abs_and_cut_distance": {str(k): v for k, v in diffs.items()}})
                assert diffs[i][0] > 1e-3 and diffs[i][1] < 1e-6 and diffs[i - 1][0] < 5e-5 and diffs[i + 1][0] < 5e-5, diffs
            # the queue, three sessions of it sharing a batch: the synchronous call's bits
            q = [h.run_enqueue(pcm, run_cfg, L.pinned_empty(w.shape, np.float32)) for _ in range(3)]
            h.wait()
            assert all(np.array_equal(v, w) for v in q), name
    finally:
        sep.close()

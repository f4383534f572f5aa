"""The HIP path (through the C ABI) against the second set of reference fixtures (tests/golden/gen_golden_r2.py): every
non-default CssCfg branch, three other segmentations, BASELINE.json configs[1] / configs[2] at full size (60 s), and
css_inference against the captured session triple.  Needs an MI355X."""
import hashlib
import json
import os

import numpy as np
import pytest

import css_oracle as O
from conftest import GOLDEN, pkg, rel_rms, take_windows
from test_oracle_golden_r2 import VARIANTS, sha, unpack2, unpack_bits

pytestmark = pytest.mark.gpu

F, S = 257, 3


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


def staged_run(h, L, pcm, run_cfg, wta=None):
    """the pass stage by stage, with the reference's winner-take-all decisions injected when given"""
    h.begin(pcm, pcm.shape[0], pcm.shape[1], run_cfg)
    p = h.get_plan()
    nseg, TL = int(p.num_segments), int(p.mix_frames)
    if wta is not None:
        h.write(L.BUF_WTA_OVERRIDE, wta)
    h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
    h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
    return h.read(L.BUF_WAV), h.read(L.BUF_PERMS), h.read(L.BUF_ACT_B).astype(bool).T, h.read(L.BUF_ACT_FINAL).astype(bool).T


@pytest.mark.parametrize("name", list(VARIANTS))
def test_csscfg_branches_vs_reference(L, sep_mc, mix60, golden, name):
    """css.py:211-247 and :263-271 against the REFERENCE (not the oracle): decisions exact, waveforms <= 1e-4."""
    CSS = pkg("css")
    g = golden("variants_mc.npz")
    mix = np.ascontiguousarray(mix60[0, int(g["opt_offset"]):int(g["opt_offset"]) + int(g["opt_samples"])])
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False, **VARIANTS[name]), 16000, 7)
    wav, perms, act_b, act_f = staged_run(sep_mc.handle, L, mix, run_cfg, g["opt_wta_index"])
    p = f"opt_{name}"
    assert wav.shape[1] == int(g[p + "_wav_len"])
    assert [tuple(x) for x in perms[1:]] == [tuple(x) for x in g[p + "_pit_perm"]]
    shape = tuple(g[p + "_activity_shape"])
    assert np.array_equal(act_b, unpack_bits(g[p + "_activity_b"], shape))
    assert np.array_equal(act_f, unpack_bits(g[p + "_activity_final"], shape))
    ww = take_windows(wav, 4)
    for k in range(S):   # the fourth window lies in the ragged, ill-conditioned last segment (test_oracle_golden_r2.py)
        assert rel_rms(ww[k][:3], g[p + "_wav_windows"][k][:3]) < 1e-4, (name, k)


# (3 s, 0.5 s): six segments over every frame; (5 s, 2.5 s): 311-frame segments -- round-3 fixtures (gen_golden_r3.py);
# (10 s, 5 s): 624-frame segments, beyond what the tuned kernels hold -- round 4 (gen_golden_r4b.py)
@pytest.mark.parametrize("seg_hop", [(3.0, 2.0), (4.0, 2.0), (2.0, 1.0), (3.0, 0.5), (5.0, 2.5), (10.0, 5.0)])
def test_other_segmentations_vs_reference(L, sep_mc, mix60, golden, seg_hop):
    CSS = pkg("css")
    g = golden("segs_long_r4.npz" if seg_hop == (10.0, 5.0) else "segs_r3.npz" if seg_hop in ((3.0, 0.5), (5.0, 2.5)) else "variants_mc.npz")
    name = f"seg{int(seg_hop[0])}{int(seg_hop[1])}"
    mix = np.ascontiguousarray(mix60[0, int(g["seg_offset"]):int(g["seg_offset"]) + int(g["seg_samples"])])
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False, segment_size_sec=seg_hop[0], hop_size_sec=seg_hop[1])
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    wta = unpack2(g[name + "_wta_index"], g[name + "_wta_shape"])
    h = sep_mc.handle
    wav, perms, act_b, act_f = staged_run(h, L, mix, run_cfg, wta)
    Ts = int(g[name + "_segment_frames"])
    assert int(run_cfg.c.segment_frames) == Ts and h.get_plan().num_segments == wta.shape[0]
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, wta.shape[0], Ts)
    assert np.abs(np.moveaxis(m[:S, :, 0], 0, 2)[::8, ::4] - g[name + "_masks_spk_seg0"]).max() < 1.5e-5
    # winner-take-all maps: rounding-level ties only -- except in a segment with an IPD feature ON the atan2 branch cut
    # (test_oracle_golden_r2.py: the reference is discontinuous there), identified by the feature itself
    X = O.stft(mix)
    hop_f = int(run_cfg.c.hop_frames)
    def on_cut(i):
        seg = np.zeros((257, Ts, 7), np.complex64)
        part = X[:, i * hop_f:i * hop_f + Ts]
        seg[:, :part.shape[1]] = part
        f = O.features(seg)[257:].reshape(6, 257, -1)[:, 1:256]
        return bool(np.abs(np.abs(f) - np.pi).min() < 1e-6)
    per_seg = [int((np.argmax(m[:, :, i], axis=0) != wta[i]).sum()) for i in range(wta.shape[0])]
    cut = [i for i in range(wta.shape[0]) if per_seg[i] > 3 and on_cut(i)]
    assert sum(n for i, n in enumerate(per_seg) if i not in cut) <= 1e-5 * wta.size + 3, per_seg
    assert all(per_seg[i] <= 0.005 * 257 * Ts for i in cut) and len(cut) <= 1, (cut, per_seg)
    assert [tuple(x) for x in perms[1:]] == [tuple(x) for x in g[name + "_pit_perm"]]
    assert np.array_equal(act_f, unpack_bits(g[name + "_activity_final"], tuple(g[name + "_activity_shape"])))
    # windows outside the on-cut segments (their mask VALUES weight the covariances) and the ragged last segment
    from conftest import window_starts
    ww = take_windows(wav, 4)
    keep = [j for j, s0 in enumerate(window_starts(wav.shape[1], 4)[:3])
            if not any(i * hop_f * 256 - 2048 <= s0 <= (i * hop_f + Ts) * 256 for i in cut)]
    assert keep, (cut, per_seg)
    for k in range(S):
        assert rel_rms(ww[k][keep], g[name + "_wav_windows"][k][keep]) < 1e-4, (name, k, keep)


def test_config2_60s_mc_vs_reference(L, sep_mc, mix60, golden, mc_state):
    """BASELINE.json configs[1] at full size against the reference's own 60 s run: decisions, winner-take-all maps,
    stitched masks, and the waveforms -- free-running where no decision differs, and on the reference's decisions."""
    CSS = pkg("css")
    g = golden("e2e60_mc.npz")
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep_mc.handle
    pcm = np.ascontiguousarray(mix60[0])
    free = h.run(pcm, run_cfg)
    wta = unpack2(g["wta_packed"], g["wta_shape"])
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, 40, 186)
    perms = h.read(L.BUF_PERMS)
    act_f = h.read(L.BUF_ACT_FINAL).astype(bool).T
    mask_st = h.read(L.BUF_MASK_ST)                                    # [S, F, T_long]
    assert free.shape == (3, int(g["wav_len"])) and h.get_plan().num_segments == int(g["num_segments"])
    assert sha(np.ascontiguousarray(perms[1:]).astype(np.int32)) == str(g["sha_pit_perm"])
    # segments with an IPD feature ON the atan2 branch cut (see test_oracle_golden_r2.py): the reference is discontinuous
    # there; everything else must agree to rounding
    X = O.stft(pcm)
    def on_cut(i):
        f = O.features(X[:, i * 93:i * 93 + 186])[257:].reshape(6, 257, -1)[:, 1:256]
        return bool(np.abs(np.abs(f) - np.pi).min() < 1e-6)
    cut = [i for i in range(39) if on_cut(i)]
    # a decision = the SET of masks that equal the maximum (mvdr_util.py:53-54): an exact tie of two of our masks where the
    # reference has a single winner is a differing decision even when argmax agrees (about one such point per meeting is what
    # chance gives any float32 evaluation: 143 of the reference's own margins are below 2e-5, e2e60_r6_self.npz)
    ours_win = m == m.max(axis=0, keepdims=True)
    ref_win = np.arange(4)[:, None, None, None] == np.moveaxis(wta, 0, 1)[None]
    per_seg = [int(np.any(ours_win[:, :, i] != ref_win[:, :, i], axis=0).sum()) for i in range(40)]
    assert sum(n for i, n in enumerate(per_seg) if i not in cut) <= 1e-5 * wta.size + 3, per_seg
    assert all(per_seg[i] <= 0.005 * 257 * 186 for i in cut) and len(cut) <= 8, (cut, per_seg)
    # out of the comparisons: on-cut segments whose decisions actually moved (an IPD feature landed on the other side of
    # the cut: their mask VALUES differ too) and the ragged last segment (ill-conditioned, test_oracle_golden_r2.py); an
    # on-cut segment without a flipped decision stays in (its masks are held to the stitched-mask bound below)
    moved = [i for i in cut if per_seg[i] > 0]
    stable_t = np.ones(3749, bool)
    for i in moved + [39]:
        stable_t[i * 93:i * 93 + 186 + 2] = False
    ref_act = unpack_bits(g["activity_final"], tuple(g["activity_shape"]))
    assert np.array_equal(act_f[stable_t], ref_act[stable_t])
    if all(per_seg[i] == 0 for i in cut):
        assert sha(np.packbits(act_f)) == str(g["sha_activity_final"])
    ms = np.abs(mask_st.transpose(1, 2, 0)[::32, ::16] - g["mask_stitched"])
    assert ms[:, stable_t[::16]].max() < 1.5e-5
    # waveforms, one sample per frame, outside the unstable regions: on the reference's decisions, and free-running
    # wherever no decision of the covering segments differs
    forced, _, _, _ = staged_run(h, L, pcm, run_cfg, wta)
    t = np.flatnonzero(stable_t)
    for k in range(S):
        assert rel_rms(forced[k, ::256][t], g["wav_dec"][k][t]) < 1e-4, k
    clean_t = stable_t.copy()
    for i, n in enumerate(per_seg):
        if n:
            clean_t[max(i * 93 - 2, 0):i * 93 + 186 + 2] = False
    t = np.flatnonzero(clean_t)
    free_err = [rel_rms(free[k, ::256][t], g["wav_dec"][k][t]) for k in range(S)]
    forced_err = [rel_rms(forced[k, ::256][np.flatnonzero(stable_t)], g["wav_dec"][k][np.flatnonzero(stable_t)]) for k in range(S)]
    # how much of the meeting each comparison covers ("the evidence should say how much was compared")
    from test_hip_long import _report
    _report("config2_60s_mc", {
        "frames": 3749, "segments": 40,
        "segments_on_the_ipd_branch_cut": cut, "of_which_with_moved_decisions": moved,
        "segments_with_wta_flips": [i for i, n in enumerate(per_seg) if n],
        "wta_flips_per_segment": per_seg, "wta_decisions": int(wta.size),
        "frames_compared_on_the_reference_decisions": int(stable_t.sum()), "fraction_on_the_reference_decisions": round(float(stable_t.mean()), 4),
        "frames_compared_free_running": int(clean_t.sum()), "fraction_free_running": round(float(clean_t.mean()), 4),
        "waveform_rel_rms_on_the_reference_decisions": forced_err, "waveform_rel_rms_free_running": free_err})
    # measured (round 3, MI355X): three on-cut segments with moved decisions + the ragged last one leave 0.842 of the frames
    # for the comparison on the reference's decisions (6.7 ... 7.6e-6), one more segment with a single rounding-level flip
    # takes the free-running comparison to 0.790 (6.6 ... 7.5e-6); asserted with a margin of two segments (round 6, the default
    # exact float32 arithmetic: one more segment with an exact tie of two of our masks)
    assert stable_t.mean() >= 0.80 and clean_t.mean() >= 0.69, (stable_t.mean(), clean_t.mean())
    for k in range(S):
        assert free_err[k] < 1e-4, k
    # ---- the frames left out above -- on-cut segments whose decisions moved, the ragged last one -- are held to the ORACLE,
    # so that no frame of configs[1] goes unchecked: (a) their masks against the oracle's Conformer on the very feature rows
    # the HIP kernel produced for the segment (the discontinuity sits in the features, not behind them); (b) the whole
    # meeting's waveforms against the oracle's float64 chain on the HIP masks, reported for exactly the left-out samples
    params = O.ConformerParams(mc_state[0])
    left_out = sorted(set(moved + [39]))
    mask_err = {}
    for i in left_out:
        h.begin(pcm, pcm.shape[0], pcm.shape[1], run_cfg)
        h.stage_stft()
        h.stage_masknet(i, i + 1)
        feat = h.read(L.BUF_FEATURES)[:186, :1799]
        om = O.conformer_forward(params, feat.T, affine_applied=True)
        hm = h.read(L.BUF_MASKS).reshape(S + 1, F, 40, 186)[:, :, i]
        assert np.array_equal(hm, m[:, :, i])                      # the segment alone == the segment in the free run
        mask_err[i] = float(np.abs(hm - om).max())
        assert mask_err[i] < 1.5e-5, (i, mask_err)
    hip_masks = [(np.moveaxis(m[:S, :, i], 0, 2), np.moveaxis(m[S:, :, i], 0, 2)) for i in range(40)]
    ow, oside = O.separate_and_stitch(mix60, None, 16000, O.OracleCssCfg(activity_th=0.3),
                                      separate_fn=lambda i, seg: hip_masks[i], mvdr_cplx=np.complex128)
    assert np.array_equal(np.array(oside["perms"]), perms) and np.array_equal(oside["activity_final"][0], act_f)
    t_out = np.flatnonzero(~clean_t)
    out_err = [rel_rms(free[k, ::256][t_out], ow[k][::256][t_out]) for k in range(S)]
    all_err = [rel_rms(free[k, ::256], ow[k][::256]) for k in range(S)]
    _report("config2_60s_mc_left_out_frames_vs_oracle", {
        "segments": left_out, "frames": int((~clean_t).sum()), "masks_max_abs_vs_oracle_on_hip_features": mask_err,
        "waveform_rel_rms_vs_oracle_on_hip_masks_left_out_frames": out_err, "waveform_rel_rms_vs_oracle_on_hip_masks_all_frames": all_err})
    for k in range(S):
        assert out_err[k] < 1e-4 and all_err[k] < 1e-4, (k, out_err, all_err)


def test_config3_60s_sc_vs_reference(L, sc_state, mix60, golden):
    """BASELINE.json configs[2] at full size: single-channel model, no beamformer, mask multiplication."""
    CSS = pkg("css")
    g = golden("e2e60_sc.npz")
    sep = pkg("separator").HipSeparator(sc_state[0], None, device=0, max_batch_segments=64)
    try:
        h = sep.handle
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 1)
        wav = h.run(np.ascontiguousarray(mix60[0, :, :1]), run_cfg)
        assert h.get_plan().num_segments == 40 and wav.shape[1] == int(g["wav_len"])
        assert sha(np.ascontiguousarray(h.read(L.BUF_PERMS)[1:]).astype(np.int32)) == str(g["sha_pit_perm"])
        assert sha(np.packbits(h.read(L.BUF_ACT_FINAL).astype(bool).T)) == str(g["sha_activity_final"])
        for k in range(S):
            assert rel_rms(wav[k, ::256], g["wav_dec"][k]) < 2e-5, k
            assert rel_rms(take_windows(wav, 4)[k], g["wav_windows"][k]) < 2e-5, k
    finally:
        sep.close()


def test_css_inference_against_the_captured_triple(tmp_path, mc_state, mix60):
    """css_inference (css.py:51-107) on the session the reference was run on (tests/golden/gen_golden_r4.py: 19.4 s of the
    config-2 meeting at full scale, PCM16 files in, the reference's own load_audio / write_wav around its separate_and_stitch):
    same Series columns, same files, same lengths, the input mixture bit for bit, and the separated streams -- the product
    as a consumer sees it, wav in, wav out, free-running -- within ONE PCM16 step on >= 99.9 % of the samples of every
    stream; cache and pass-through rules."""
    import pandas as pd
    import torch
    import yaml
    with open(os.path.join(GOLDEN, "session_triple_r4.json")) as f:
        t = json.load(f)
    ref16 = np.load(os.path.join(GOLDEN, "session_triple_r4_pcm16.npz"))
    CSS, W = pkg("css"), pkg("wavio")
    n, off, gain = t["input"]["n_samples"], t["input"]["mix_offset"], t["input"]["pcm16_gain"]
    pcm16 = np.clip(np.rint(mix60[0, off:off + n] * np.float32(gain) * 32768.0), -32768, 32767).astype(np.int16)
    names = []
    for c in range(7):
        p = tmp_path / "in" / f"ch{c}.wav"
        W.write_pcm16_samples(p, pcm16[:, c], 16000)
        names.append(str(p))
    # a checkpoint directory as the trainer writes it (css/helpers.py:14-37): one yaml, one .pt with "module." keys
    mdir = tmp_path / "models" / "notsofar" / "conformer1.0" / "mc"
    mdir.mkdir(parents=True)
    torch.save({"model": {"module." + k: torch.from_numpy(np.asarray(v)) for k, v in mc_state[0].items()}}, mdir / "ckpt.pt")
    with open(mdir / "train.yaml", "w") as f:
        yaml.safe_dump({"conformer_css_cfg": {"nnet_conf": {"conformer_conf": {
            "attention_dim": 512, "attention_heads": 8, "num_blocks": 18, "dropout_rate": 0.0}}}}, f)
    session = pd.Series({"session_id": t["input"]["session_id"], "is_mc": True, "wav_file_names": names, "device_name": "synth"})
    out_dir = tmp_path / "out"
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    res = CSS.css_inference(str(out_dir), str(tmp_path / "models"), session, cfg, fetch_from_cache=False)
    rel = lambda p: os.path.relpath(str(p), str(out_dir))
    assert sorted(res.index.tolist()) == t["output_columns"]
    assert [rel(p) for p in res["sep_wav_file_names"]] == t["sep_wav_file_names"]
    files = sorted(rel(os.path.join(dp, f)) for dp, _, fs in os.walk(out_dir) for f in fs)
    assert files == t["files"]
    margins = {}
    for name in files:
        pcm, sr = W.read_wav_pcm16(out_dir / name)
        assert sr == 16000 and len(pcm) == t["lengths"][name]
        if name.endswith("input_mixture.wav"):
            assert sha(pcm.astype(np.int16)) == t["pcm16_sha256"][name]
        else:
            step = np.abs(pcm.astype(np.int64) - ref16[name.replace("/", "__")].astype(np.int64))
            margins[os.path.basename(name)] = {"equal": round(float((step == 0).mean()), 5), "within_1_lsb": round(float((step <= 1).mean()), 6),
                                               "max_steps": int(step.max())}
            assert (step <= 1).mean() >= 0.999, (name, margins)
    from test_hip_long import _report
    _report("session_triple_pcm16_steps_vs_reference_files", margins)
    res2 = CSS.css_inference(str(out_dir), str(tmp_path / "models"), session, cfg, fetch_from_cache=True)
    assert [rel(p) for p in res2["sep_wav_file_names"]] == t["cached_sep_wav_file_names"]
    res3 = CSS.css_inference(str(out_dir), "unused", session, CSS.CssCfg(pass_through_ch0=True), fetch_from_cache=False)
    assert [os.path.basename(p) for p in res3["sep_wav_file_names"]] == t["pass_through"]


def test_validation_loss_vs_reference(L):
    """css_validation_loss_host (train.py:411 _calc_loss on the device) against the reference's own _calc_loss +
    PitWrapper: every loss / base-loss / clipping branch; per-sample speaker losses, target permutations, the scalar."""
    from test_oracle_golden_r2 import loss_inputs
    with open(os.path.join(GOLDEN, "val_loss.json")) as f:
        g = json.load(f)
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=g["num_blocks"])
    st = w.apply_golden_recipe(w.portable_state_dict(desc, g["weights_seed"]))
    mix, gt_spk0, gt_noise0 = loss_inputs(g["seed"], g["batch"], g["n"])
    sep = pkg("separator").HipSeparator(st, None, device=0)
    try:
        for mode in ("split_f16", "exact_f32"):
            sep.handle.set_linear_mode(mode)
            for case in g["cases"]:
                loss, spk, noi, perms = sep.handle.validation_loss(mix, gt_spk0, gt_noise0, case["loss_name"], case["base_loss"],
                                                                   case["clip_gt_to_mixture"], case["noise_weight"])
                assert perms.tolist() == case["perms"], (mode, case)
                assert np.allclose(spk, case["spk_loss"], rtol=5e-5), (mode, case, spk)
                assert abs(loss / case["loss"] - 1) < 5e-5, (mode, case, loss)
                ol, ospk, onoi, _ = O.validation_loss(O.ConformerParams(st), mix, gt_spk0, gt_noise0, case["loss_name"],
                                                     case["base_loss"], case["clip_gt_to_mixture"], case["noise_weight"])
                assert np.allclose(noi, onoi, rtol=5e-5) and abs(loss / ol - 1) < 5e-5
        with pytest.raises(L.CssError):
            sep.handle.validation_loss(mix[:, :, :1], gt_spk0, gt_noise0)      # one channel into the 7-channel model
    finally:
        sep.close()

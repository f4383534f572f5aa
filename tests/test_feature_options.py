"""ExtractorCfg options the shipped v1.0 models do not use (conformer_wrapper.py:11-24; FeatureExtractor / IPDFeature,
css/css_with_conformer/executor/feature.py:198-249,478-508): log_spectrogram, mvn_spectrogram off, IPD time-mean
normalisation off / versions 2 and 3, ipd_cos, other ipd_index pairs.  Fixtures from the reference
(tests/golden/gen_golden_r3.py: features and masks of a seeded 2-block model per option set).  CPU: the oracle against the
fixtures; GPU: the HIP front end against the oracle and the fixtures."""
import numpy as np
import pytest

import css_oracle as O
from conftest import pkg

OPTION_SETS = {
    "log_v2_cos": dict(log_spectrogram=True, ipd_mean_normalize_version=2, ipd_cos=True),
    "v3_pairs": dict(ipd_mean_normalize_version=3, ipd_index="1,4;2,5;3,6;1,0;2,0;3,0"),
    "nomvn_nonorm": dict(mvn_spectrogram=False, ipd_mean_normalize=False),
    "three_pairs_cos": dict(ipd_index="1,4;2,5;3,6", ipd_cos=True, log_spectrogram=True),
}


def _setup(name, golden, mix60):
    S = pkg("separator")
    W = pkg("weights")
    g = golden("feature_opts_r3.npz")
    e = S.ExtractorCfg(**OPTION_SETS[name])
    pairs = S.ipd_pairs(e.ipd_index)
    desc = W.ModelDesc(num_blocks=2, in_features=257 * (1 + len(pairs)))
    assert desc.in_features == int(g[f"{name}_in_features"])
    st = W.apply_golden_recipe(W.portable_state_dict(desc, 5))
    cfg = S.ConformerCssCfg(extractor_conf=e, nnet_conf=S.NnetCfg(in_features=desc.in_features, conformer_conf=S.ConformerCfg(
        attention_dim=desc.attention_dim, attention_heads=desc.attention_heads, num_blocks=2, dropout_rate=0.0,
        linear_units=desc.linear_units, kernel_size=desc.kernel_size)))
    mix = mix60[:, int(g["offset"]):int(g["offset"]) + int(g["samples"])]
    okw = dict(S.feature_options(e))
    return g, st, desc, cfg, mix, okw


def _angle_or_value_diff(a, b, is_angle_rows):
    d = np.abs(a - b)
    d[is_angle_rows] = np.minimum(d[is_angle_rows], 2 * np.pi - d[is_angle_rows])   # raw angles: +-pi is one point
    return d


@pytest.mark.parametrize("name", sorted(OPTION_SETS))
def test_oracle_features_and_masks_vs_reference(name, golden, mix60):
    g, st, desc, cfg, mix, okw = _setup(name, golden, mix60)
    x = O.stft(mix[0])                                          # [F, 186, 7]
    assert x.shape[1] == 186
    f = O.features(x, **okw)
    ref = g[f"{name}_features"]
    got = f[::4, ::3]
    assert got.shape == ref.shape
    angle = np.zeros(f.shape[0], bool)
    angle[257:] = not okw["ipd_cos"]
    d = _angle_or_value_diff(got, ref, angle[::4])
    scale = max(float(np.abs(ref).max()), 1.0)
    assert np.percentile(d, 99) < 2e-5 * scale and d.max() < 2e-3 * scale, (name, float(d.max()), float(np.percentile(d, 99)))
    masks = O.conformer_forward(O.ConformerParams(st), f)      # [S + 1, F, T]
    assert np.abs(np.moveaxis(masks[:3], 0, 2)[::8, ::4] - g[f"{name}_spk_masks"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OPTION_SETS))
def test_hip_features_and_masks_vs_oracle_and_reference(name, golden, mix60):
    import torch
    L = pkg("_lib")
    g, st, desc, cfg, mix, okw = _setup(name, golden, mix60)
    sep = pkg("separator").HipSeparator(st, cfg, device=0)
    try:
        x = O.stft(mix[0])
        of = O.features(x, **okw)
        om = O.conformer_forward(O.ConformerParams(st), of)
        out = sep.separate(sep.stft(torch.from_numpy(mix)))
        spk = out["spk_masks"].numpy()[0]                        # [F, T, S]
        assert np.abs(spk - np.moveaxis(om[:3], 0, 2)).max() < 1.5e-5
        assert np.abs(spk[::8, ::4] - g[f"{name}_spk_masks"]).max() < 2e-5
        # the feature rows themselves, through a staged session (two segments; segment 0 = the fixture's 186 frames;
        # input affine of the seeded model: bias 0, scale 4 = the golden recipe's input gain)
        CSS = pkg("css")
        g_off = int(g["offset"])
        long_mix = np.ascontiguousarray(mix60[0, g_off:g_off + 64000])
        h = sep.handle
        h.begin(long_mix, long_mix.shape[0], 7, CSS.make_run_cfg(CSS.CssCfg(show_progressbar=False), 16000, 7))
        h.stage_stft()
        h.stage_masknet(0, 1)
        feat = h.read(L.BUF_FEATURES)[:186, :desc.in_features]
        W = pkg("weights")
        bias = np.asarray(st[W.PREFIX + "input_bias"], np.float32).reshape(-1)
        scale_in = np.asarray(st[W.PREFIX + "input_scale"], np.float32).reshape(-1)
        fo = (of.T + bias) * scale_in                              # conformer.py:298-299
        d = np.abs(feat - fo)
        if not okw["ipd_cos"]:                                     # raw angles: +-pi is one point
            wrap = np.abs(d[:, 257:] - 2 * np.pi * scale_in[257:])
            d[:, 257:] = np.minimum(d[:, 257:], wrap)
        scale = max(float(np.abs(fo).max()), 1.0)
        assert np.percentile(d, 99) < 1e-4 * scale, (name, float(np.percentile(d, 99)))
    finally:
        sep.close()


def test_unsupported_extractor_options_are_rejected_loudly():
    S = pkg("separator")
    for kw in (dict(ang_index="1,0;2,0"), dict(frame_len=400, frame_hop=150), dict(frame_len=402, frame_hop=160), dict(frame_len=256, frame_hop=512)):
        with pytest.raises(NotImplementedError):
            S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(**kw)))
    # other frame sizes are mapped as init_kernel maps them (feature.py:27): 400 samples on 512 FFT points -> 257 bins
    d = S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(frame_len=400, frame_hop=160)))
    assert (d.frame_len, d.frame_hop, d.num_bins) == (400, 160, 257)
    with pytest.raises(AssertionError):    # 1024 FFT points are 513 bins: not this network's 1799 = 257 x 7 inputs
        S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(frame_len=1024, frame_hop=256)))
    with pytest.raises(RuntimeError):
        S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(ipd_mean_normalize_version=4)))
    with pytest.raises(RuntimeError):      # init_kernel's own error (feature.py:24-25)
        S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(window="hamming")))
    # both windows of init_kernel and the no-op round_pow_of_two are accepted
    for kw in (dict(window="sqrt_hann"), dict(round_pow_of_two=False)):
        assert S.desc_from_cfg(S.ConformerCssCfg(extractor_conf=S.ExtractorCfg(**kw))).frame_len == 512


# ---------------------------------------------------------------------------------------------------------------------
# The second analysis window init_kernel builds (feature.py:19-45): ExtractorCfg(window='sqrt_hann'); fixture from the
# reference (tests/golden/gen_golden_r4b.py).  round_pow_of_two=False is the same kernel at frame_len 512 (recorded there).
# ---------------------------------------------------------------------------------------------------------------------
def _window_setup(golden, mix60):
    S, W = pkg("separator"), pkg("weights")
    g = golden("window_r4.npz")
    assert bool(g["round_pow_of_two_false_is_identical"])
    e = S.ExtractorCfg(window="sqrt_hann", round_pow_of_two=False)
    desc = W.ModelDesc(num_blocks=2)
    st = W.apply_golden_recipe(W.portable_state_dict(desc, 5))
    cfg = S.ConformerCssCfg(extractor_conf=e, nnet_conf=S.NnetCfg(in_features=desc.in_features, conformer_conf=S.ConformerCfg(
        attention_dim=desc.attention_dim, attention_heads=desc.attention_heads, num_blocks=2, dropout_rate=0.0,
        linear_units=desc.linear_units, kernel_size=desc.kernel_size)))
    mix = mix60[:, int(g["offset"]):int(g["offset"]) + int(g["samples"])]
    return g, st, desc, cfg, mix


def test_oracle_sqrt_hann_window_vs_reference(golden, mix60):
    g, st, desc, cfg, mix = _window_setup(golden, mix60)
    x = O.stft(mix[0], window="sqrt_hann")                      # [F, 186, 7]
    ref = g["sqrt_hann_stft"]
    scale = float(np.abs(ref).max())
    assert np.abs(x[::4, ::3] - ref).max() < 2e-6 * scale       # (the bar of the Hann transform)
    # the kernel itself: sqrt of the float32 Hann window over S = 16; not the Hann kernel
    k = O.stft_kernel(window="sqrt_hann")
    assert np.allclose(k[0], np.sqrt(O.hann_periodic(512)) / 16.0, rtol=0, atol=1e-9)
    assert np.abs(k - O.stft_kernel()).max() > 0.01
    with pytest.raises(RuntimeError):
        O.stft_kernel(window="hamming")
    f = O.features(x)
    d = np.abs(f[::8, ::3] - g["sqrt_hann_features"])
    angle = np.arange(f.shape[0])[::8] >= 257
    d[angle] = np.minimum(d[angle], 2 * np.pi - d[angle])
    assert np.percentile(d, 99) < 2e-5 * max(float(np.abs(g["sqrt_hann_features"]).max()), 1.0) and d.max() < 2e-3 * 4
    masks = O.conformer_forward(O.ConformerParams(st), f)
    assert np.abs(np.moveaxis(masks[:3], 0, 2)[::8, ::4] - g["sqrt_hann_spk_masks"]).max() < 2e-5


@pytest.mark.gpu
def test_hip_sqrt_hann_window_vs_oracle_and_reference(golden, mix60):
    import torch
    g, st, desc, cfg, mix = _window_setup(golden, mix60)
    sep = pkg("separator").HipSeparator(st, cfg, device=0)
    try:
        X = sep.stft(torch.from_numpy(mix))                      # [1, F, 186, 7] complex64
        x = X.numpy()[0]
        xo = O.stft(mix[0], window="sqrt_hann")
        rel_rms = lambda a, b: float(np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2)))
        assert rel_rms(x, xo) < 1e-6                             # (the bars of the Hann transform, tests/test_hip_parity.py)
        assert rel_rms(x[::4, ::3], g["sqrt_hann_stft"]) < 2e-6
        assert np.abs(x.imag[0]).max() == 0 and np.abs(x.imag[256]).max() == 0
        out = sep.separate(X)
        spk = out["spk_masks"].numpy()[0]
        om = O.conformer_forward(O.ConformerParams(st), O.features(xo))
        assert np.abs(spk - np.moveaxis(om[:3], 0, 2)).max() < 1.5e-5
        assert np.abs(spk[::8, ::4] - g["sqrt_hann_spk_masks"]).max() < 2e-5
        # back to the Hann window on the same handle: the shipped transform again
        sep.handle.set_analysis_window("hann")
        xh = sep.stft(torch.from_numpy(mix)).numpy()[0]
        assert rel_rms(xh, O.stft(mix[0])) < 1e-6
        assert rel_rms(xh, xo) > 0.1
        with pytest.raises(RuntimeError):
            sep.handle.set_analysis_window("hamming")
    finally:
        sep.close()

#!/usr/bin/env python3
"""Round-3 fixtures, produced by running the REAL reference in the build container (rules as in gen_golden.py: the
reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r3.py       # writes segs_r3.npz, feature_opts_r3.npz

  segs_r3.npz   two segmentations outside what rounds 1-2 covered, on the 11 s input of variants_mc.npz:
                (3 s, 0.5 s): six segments over every frame (hop = T / 6, css.py:144-171 accepts any hop) and
                (5 s, 2.5 s): 311-frame segments (> 256 frames: the long-segment attention / feature / covariance kernels).
                Same keys as the seg* entries of variants_mc.npz (gen_golden_r2.py).
  feature_opts_r3.npz   ExtractorCfg options the shipped models do not use (log_spectrogram, mvn_spectrogram off, IPD
                normalisation off / versions 2 and 3, ipd_cos, other ipd_index pairs): the reference FeatureExtractor's
                features (decimated) and the speaker masks of a seeded 2-block model built with each option set."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up sys.path / stubs, imports the reference as G.RC)
import gen_golden_r2 as G2  # noqa: E402

import torch  # noqa: E402

RC, W, SYN = G.RC, G.W, G.SYN


def main():
    torch.manual_seed(0)
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    mix60 = SYN.synth_meeting(60.0, 7, seed=1)
    base = dict(show_progressbar=False, activity_th=0.3)
    mix_s = mix60[:, 32000:32000 + 11 * 16000 + 300]
    out = {"seg_offset": 32000, "seg_samples": mix_s.shape[1]}
    for seg, hop in ((3.0, 0.5), (5.0, 2.5)):
        name = f"seg{int(seg)}{int(hop)}"
        w3, s3, t3, _ = G.run_reference(model, mix_s, RC.CssCfg(**base, segment_size_sec=seg, hop_size_sec=hop))
        out.update(G2.outputs(name, w3, s3, t3.pit))
        out[f"{name}_wta_index"] = G2.pack2(G2.wta_of(t3.masks))
        out[f"{name}_wta_shape"] = np.array(G2.wta_of(t3.masks).shape)
        out[f"{name}_segment_frames"] = int(s3["segment_frames"])
        out[f"{name}_masks_spk_seg0"] = t3.masks[0]["spk_masks"][0, ::8, ::4]
        print(name, "segments", len(t3.masks), "frames", int(s3["segment_frames"]), "perms", sorted({tuple(p) for _, p in t3.pit}))
    np.savez_compressed(os.path.join(HERE, "segs_r3.npz"), **out)

    # ------------------------------------------------------------------ ExtractorCfg options beyond the shipped ones
    # (feature.py:198-249,478-508): the reference's FeatureExtractor features and the masks of a 2-block model built with
    # each option set, on one 186-frame segment
    from css.training.conformer_wrapper import ConformerCssWrapper, ConformerCssCfg, NnetCfg, ConformerCfg, ExtractorCfg
    OPTION_SETS = {
        "log_v2_cos": dict(log_spectrogram=True, ipd_mean_normalize_version=2, ipd_cos=True),
        "v3_pairs": dict(ipd_mean_normalize_version=3, ipd_index="1,4;2,5;3,6;1,0;2,0;3,0"),
        "nomvn_nonorm": dict(mvn_spectrogram=False, ipd_mean_normalize=False),
        "three_pairs_cos": dict(ipd_index="1,4;2,5;3,6", ipd_cos=True, log_spectrogram=True),
    }
    seg_mix = mix60[:, 16000:16000 + 48000]
    fo = {"offset": 16000, "samples": 48000}
    for name, kw in OPTION_SETS.items():
        e = ExtractorCfg(**kw)
        npairs = len(e.ipd_index.split(";")) if e.ipd_index else 0
        d2 = W.ModelDesc(num_blocks=2, in_features=257 * (1 + npairs))
        st2 = W.apply_golden_recipe(W.portable_state_dict(d2, 5))
        cfg = ConformerCssCfg(extractor_conf=e, nnet_conf=NnetCfg(in_features=d2.in_features, conformer_conf=ConformerCfg(
            attention_dim=d2.attention_dim, attention_heads=d2.attention_heads, num_blocks=2, dropout_rate=0.0)))
        m2 = ConformerCssWrapper(cfg).eval()
        missing, unexpected = m2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st2.items()}, strict=False)
        assert not unexpected and all(k.endswith(".K") for k in missing), (missing, unexpected)
        with torch.no_grad():
            stft = m2.stft(torch.from_numpy(seg_mix))                       # [1, F, 186, 7]
            x = stft.moveaxis(3, 1).contiguous()
            _, _, feat = m2.executor.extractor(mix=None, mag=x.abs(), pha=x.angle())   # [1, D, 186]
            masks = m2.separate(stft)
        fo[f"{name}_features"] = feat.numpy()[0, ::4, ::3]
        fo[f"{name}_spk_masks"] = masks["spk_masks"].numpy()[0, ::8, ::4]
        fo[f"{name}_in_features"] = d2.in_features
        print(name, "features", tuple(feat.shape), "masks", tuple(masks["spk_masks"].shape))
    np.savez_compressed(os.path.join(HERE, "feature_opts_r3.npz"), **fo)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The second analysis window of the reference's init_kernel (feature.py:19-45), produced by running the REAL reference
in the build container (rules as in gen_golden.py: the reference is imported in place, nothing of it is copied; only
seeds and output tensors are written).

    python tests/golden/gen_golden_r4b.py       # writes window_r4.npz, segs_long_r4.npz

  window_r4.npz   ExtractorCfg(window='sqrt_hann') -- and round_pow_of_two=False, which changes nothing at frame_len 512
                  -- on the 3 s clip of feature_opts_r3.npz: the wrapper's complex STFT (decimated), the reference
                  FeatureExtractor's features (decimated) and the speaker masks of a seeded 2-block model.
  segs_long_r4.npz   the reference's separate_and_stitch with 10 s segments every 5 s (624 frames: beyond the 512 the tuned
                  kernels hold; css.py:144-171 takes any segment_size_sec) on a 27 s input, v1.0-MC model with the golden
                  weights.  Same keys as the seg* entries of segs_r3.npz (gen_golden_r3.py)."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up sys.path / stubs, imports the reference as G.RC)

import torch  # noqa: E402

W, SYN = G.W, G.SYN


def main():
    torch.manual_seed(0)
    from css.training.conformer_wrapper import ConformerCssWrapper, ConformerCssCfg, NnetCfg, ConformerCfg, ExtractorCfg
    mix60 = SYN.synth_meeting(60.0, 7, seed=1)
    seg_mix = mix60[:, 16000:16000 + 48000]
    out = {"offset": 16000, "samples": 48000}
    for name, kw in {"sqrt_hann": dict(window="sqrt_hann"), "sqrt_hann_npow2": dict(window="sqrt_hann", round_pow_of_two=False)}.items():
        e = ExtractorCfg(**kw)
        d2 = W.ModelDesc(num_blocks=2)
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
        out[f"{name}_stft"] = stft.numpy()[0, ::4, ::3]                      # [65, 62, 7] complex64
        out[f"{name}_features"] = feat.numpy()[0, ::8, ::3]
        out[f"{name}_spk_masks"] = masks["spk_masks"].numpy()[0, ::8, ::4]
        print(name, "stft", tuple(stft.shape), "|X| max", float(stft.abs().max()), "features", tuple(feat.shape))
    # round_pow_of_two=False: the same kernel at a power-of-two frame length -- recorded as a fact, not as a second copy
    same = all(np.array_equal(out[f"sqrt_hann_{k}"], out.pop(f"sqrt_hann_npow2_{k}")) for k in ("stft", "features", "spk_masks"))
    assert same
    out["round_pow_of_two_false_is_identical"] = same
    np.savez_compressed(os.path.join(HERE, "window_r4.npz"), **out)

    # ------------------------------------------------------------------ segments beyond 512 frames through the reference
    import gen_golden_r2 as G2
    RC = G.RC
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    # (the clip starts at frame 1375 of the recording: of the offsets searched -- every 25 frames -- the one whose four full
    # segments keep all of their 960 000 inter-channel angles furthest from the atan2 branch cut, 3.6e-6: DESIGN.md hazard 7)
    mix_s = mix60[:, 352000:352000 + 27 * 16000 + 300]
    lo = {"seg_offset": 352000, "seg_samples": mix_s.shape[1]}
    seg, hop = 10.0, 5.0
    name = f"seg{int(seg)}{int(hop)}"
    w3, s3, t3, _ = G.run_reference(model, mix_s, RC.CssCfg(show_progressbar=False, activity_th=0.3, segment_size_sec=seg, hop_size_sec=hop))
    lo.update(G2.outputs(name, w3, s3, t3.pit))
    lo[f"{name}_wta_index"] = G2.pack2(G2.wta_of(t3.masks))
    lo[f"{name}_wta_shape"] = np.array(G2.wta_of(t3.masks).shape)
    lo[f"{name}_segment_frames"] = int(s3["segment_frames"])
    lo[f"{name}_masks_spk_seg0"] = t3.masks[0]["spk_masks"][0, ::8, ::4]
    print(name, "segments", len(t3.masks), "frames", int(s3["segment_frames"]), "perms", sorted({tuple(p) for _, p in t3.pit}))
    np.savez_compressed(os.path.join(HERE, "segs_long_r4.npz"), **lo)


if __name__ == "__main__":
    main()

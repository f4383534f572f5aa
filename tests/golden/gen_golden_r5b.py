#!/usr/bin/env python3
"""Frame sizes other than 512 / 256 (ExtractorCfg.frame_len / frame_hop, init_kernel feature.py:19-45) through the REAL
reference in the build container (rules as in gen_golden.py: the reference is imported in place, nothing of it is copied;
only seeds and output tensors are written).

    python tests/golden/gen_golden_r5b.py       # writes frames_r5.npz

The reference's wrapper builds its network with 257 mask bins whatever the extractor says (NnetCfg carries no num_bins,
conformer.py:260), so through it only frame lengths whose FFT size is 512 can run: frame_len in (256, 512] with
round_pow_of_two.  Two geometries, both on the v1.0-MC golden weights (the 1799 input features are 257 bins x 7 either way):
  f400_160   frame_len 400, hop 160 (25 ms / 10 ms: three frames over a sample; 298-frame segments)
  f512_128   frame_len 512, hop 128 (75 % overlap: four frames over a sample; 372-frame segments)
Per geometry: the wrapper's STFT of the first 3 s (decimated), segment 0's speaker masks (decimated), and the whole
separate_and_stitch run on 12.3 s: winner-take-all maps, permutations, activity, waveforms (every 64th sample + windows)."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import gen_golden_r2 as G2  # noqa: E402

import torch  # noqa: E402

RC, W, SYN = G.RC, G.W, G.SYN


def main():
    torch.manual_seed(0)
    from css.training.conformer_wrapper import ConformerCssWrapper, ConformerCssCfg, NnetCfg, ConformerCfg, ExtractorCfg
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    mix = SYN.synth_meeting(60.0, 7, seed=1)[:, 8000:8000 + 197000]        # 12.3 s, a ragged tail
    out = {"mix_seed": 1, "mix_offset": 8000, "mix_samples": mix.shape[1]}
    for name, (fl, fh) in {"f400_160": (400, 160), "f512_128": (512, 128)}.items():
        cfg = ConformerCssCfg(extractor_conf=ExtractorCfg(frame_len=fl, frame_hop=fh),
                              nnet_conf=NnetCfg(conformer_conf=ConformerCfg(attention_dim=desc.attention_dim, attention_heads=desc.attention_heads,
                                                                            num_blocks=desc.num_blocks, dropout_rate=0.0)))
        model = ConformerCssWrapper(cfg).eval()
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=False)
        assert not unexpected and all(k.endswith(".K") for k in missing), (missing, unexpected)
        with torch.no_grad():
            stft3 = model.stft(torch.from_numpy(mix[:, :48000]))               # [1, F, T, 7]
        w, side, tap, dt = G.run_reference(model, mix, RC.CssCfg(show_progressbar=False, activity_th=0.3))
        wta = G2.wta_of(tap.masks)
        out.update({
            f"{name}_frame": np.array([fl, fh, 512]),
            f"{name}_segment_frames": int(side["segment_frames"]),
            f"{name}_stft": stft3.numpy()[0, ::4, ::5],
            f"{name}_masks_spk_seg0": tap.masks[0]["spk_masks"][0, ::8, ::4],
            f"{name}_wta_packed": G2.pack2(wta), f"{name}_wta_shape": np.array(wta.shape),
            f"{name}_pit_perm": np.array([p for _, p in tap.pit], np.int32),
            f"{name}_activity_final": np.packbits(side["activity_final"].numpy()[0]),
            f"{name}_activity_b": np.packbits(side["activity_b"].numpy()),
            f"{name}_activity_shape": np.array(side["activity_b"].shape),
            f"{name}_wav_dec64": np.stack(w)[:, ::64],
            f"{name}_wav_windows": G.take_windows(np.stack(w), 4),
            f"{name}_wav_len": len(w[0]),
            f"{name}_mask_stitched": side["mask_stitched"].numpy()[0, ::16, ::8],
        })
        print(name, "segment frames", int(side["segment_frames"]), "segments", len(tap.masks), "stft", tuple(stft3.shape), "wav", len(w[0]), f"{dt:.1f} s", flush=True)
    np.savez_compressed(os.path.join(HERE, "frames_r5.npz"), **out)


if __name__ == "__main__":
    main()

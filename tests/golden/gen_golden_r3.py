#!/usr/bin/env python3
"""Round-3 fixtures, produced by running the REAL reference in the build container (rules as in gen_golden.py: the
reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r3.py       # writes segs_r3.npz

  segs_r3.npz   two segmentations outside what rounds 1-2 covered, on the 11 s input of variants_mc.npz:
                (3 s, 0.5 s): six segments over every frame (hop = T / 6, css.py:144-171 accepts any hop) and
                (5 s, 2.5 s): 311-frame segments (> 256 frames: the long-segment attention / feature / covariance kernels).
                Same keys as the seg* entries of variants_mc.npz (gen_golden_r2.py)."""
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


if __name__ == "__main__":
    main()

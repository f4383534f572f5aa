#!/usr/bin/env python3
"""configs[1] with NON-TRIVIAL decisions from the estimator's own masks (on the golden weights the
permutations are all identities and the gate never closes), through the REAL reference in the build container (rules as in
gen_golden.py: the reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r5c.py       # writes e2e60_r5_decisions.npz

The screened 61 s recording of e2e60_r5.npz (seed 22) through the reference's separate_and_stitch with
  * a separator-protocol wrapper around the reference model that hands the speaker masks of segment i back in the order
    ORDERS[i] (a seeded list: all six orders occur) -- the reference's customisation point, css.py:131,199 --, so that the
    permutation solver has to undo a different shuffle at every boundary on the masks the network really produced;
  * `activity_th` set INSIDE the range the stitched masks' activity covers -- the midpoint of the widest gap between
    neighbouring activity values in the middle of their distribution, so that no value sits within rounding of it --:
    the gate opens and closes by itself, dilation and erosion do real work.
Keys as e2e60_r5.npz + `orders`, `activity_th`."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import gen_golden_r2 as G2  # noqa: E402
import gen_golden_r5 as R5  # noqa: E402

import torch  # noqa: E402

RC, W, SYN = G.RC, G.W, G.SYN
ORDERS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def segment_orders(nseg, seed=7):
    rs = np.random.RandomState(seed)
    return np.array([ORDERS[rs.randint(6)] for _ in range(nseg)], np.int32)


class ShuffledOutputs(torch.nn.Module):
    """the reference model with the speaker masks of its i-th separate() call permuted by orders[i]"""

    def __init__(self, inner, orders):
        super().__init__()
        self.inner, self.orders, self.calls = inner, orders, 0

    def stft(self, s):
        return self.inner.stft(s)

    def istft(self, s):
        return self.inner.istft(s)

    def separate(self, stft_seg):
        out = self.inner.separate(stft_seg)
        o = [int(v) for v in self.orders[self.calls]]
        self.calls += 1
        return {"spk_masks": out["spk_masks"][..., o].contiguous(), "noise_masks": out["noise_masks"]}


def main():
    torch.manual_seed(0)
    g = np.load(os.path.join(HERE, "e2e60_r5.npz"))
    seed = int(g["mix_seed"])
    desc, st = R5.mc_model()
    model = G.build_reference_model(desc, st)
    mix = R5.meeting(seed)
    # pass 1 (default threshold): the activity values, to place the threshold in a gap
    w, side, tap, _ = G.run_reference(model, mix, RC.CssCfg(show_progressbar=False, activity_th=0.3))
    act = side["mask_stitched"].numpy()[0].mean(axis=0).reshape(-1)                 # [T * S]
    srt = np.sort(act)
    lo, hi = int(0.4 * srt.size), int(0.6 * srt.size)
    k = lo + int(np.argmax(np.diff(srt[lo:hi])))
    th = float(0.5 * (srt[k] + srt[k + 1]))
    print(f"activity range {srt[0]:.4f} .. {srt[-1]:.4f}; threshold {th:.6f} in a gap of {srt[k + 1] - srt[k]:.2e}", flush=True)
    nseg = len(tap.masks)
    orders = segment_orders(nseg)
    sh = ShuffledOutputs(model, orders).eval()
    cfg = RC.CssCfg(show_progressbar=False, activity_th=th)
    with G.Tap(model) as tap2:
        w2, side2 = RC.separate_and_stitch(mix, sh, 16000, torch.device("cpu"), cfg)
    # (the Tap wraps the inner model's separate: its masks are in the network's order; apply the orders for the WTA maps)
    masks = [{"spk_masks": m["spk_masks"][..., [int(v) for v in orders[i]]], "noise_masks": m["noise_masks"]} for i, m in enumerate(tap2.masks)]
    wta = G2.wta_of(masks)
    act_f = side2["activity_final"].numpy()[0]
    e = {"mix_seed": seed, "mix_samples": mix.shape[1], "num_segments": nseg, "orders": orders, "activity_th": th,
         "activity_gap": float(srt[k + 1] - srt[k]),
         "pit_perm": np.array([p for _, p in tap2.pit], np.int32),
         "activity_final": np.packbits(act_f), "activity_b": np.packbits(side2["activity_b"].numpy()),
         "activity_shape": np.array(side2["activity_b"].shape),
         "wta_packed": G2.pack2(wta), "wta_shape": np.array(wta.shape),
         "wav_dec64": np.stack(w2)[:, ::64], "wav_windows": G.take_windows(np.stack(w2), 4), "wav_len": len(w2[0]),
         "mask_stitched": side2["mask_stitched"].numpy()[0, ::32, ::16]}
    np.savez_compressed(os.path.join(HERE, "e2e60_r5_decisions.npz"), **e)
    perms = e["pit_perm"]
    print("distinct permutations", len({tuple(p) for p in perms}), "non-identity", int((perms != np.arange(3)).any(axis=1).sum()), "of", len(perms),
          "| gate open fraction", float(act_f.mean()), "toggles per stream", [int(np.abs(np.diff(act_f[:, k].astype(int))).sum()) for k in range(3)])


if __name__ == "__main__":
    main()

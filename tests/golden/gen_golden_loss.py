#!/usr/bin/env python3
"""Golden vectors for the validation loss of the reference's training loop (css/training/train.py:411-470 _calc_loss as
train.py:529 eval_model calls it): the REAL _calc_loss, PitWrapper and ConformerCssWrapper.forward are run in the build
container on seeded inputs (see gen_golden.py for the rules; only seeds and output scalars are written).

    python tests/golden/gen_golden_loss.py      # writes tests/golden/val_loss.json
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

import torch  # noqa: E402

W = G.W


def loss_inputs(seed=7, batch=3, n=24000, mics=7, spks=3):
    """mixture [B, n, M], gt speakers at the reference microphone [B, S, n], gt noise [B, n] (portable: RandomState)"""
    rs = np.random.RandomState(seed)
    src = rs.standard_normal((batch, spks, n)).astype(np.float32)
    for b in range(batch):                      # slowly varying envelopes so that the speakers differ over time
        for s in range(spks):
            src[b, s] *= (0.2 + 0.8 * (np.sin(2 * np.pi * (np.arange(n) / n) * (1 + s + b)) > 0)).astype(np.float32) * 0.3
    noise = (rs.standard_normal((batch, n)) * 0.05).astype(np.float32)
    gains = rs.uniform(0.5, 1.0, size=(mics, spks)).astype(np.float32)
    delays = rs.randint(0, 6, size=(mics, spks))
    mix = np.zeros((batch, n, mics), np.float32)
    for m in range(mics):
        for s in range(spks):
            mix[:, :, m] += gains[m, s] * np.roll(src[:, s], int(delays[m, s]), axis=-1)
        mix[:, :, m] += noise * np.float32(1.0 + 0.1 * m)
    gt_spk0 = np.stack([gains[0, s] * np.roll(src[:, s], int(delays[0, s]), axis=-1) for s in range(spks)], axis=1)
    return mix, gt_spk0.astype(np.float32), noise


def main():
    # css.training.train pulls in the trainer's data pipeline; the two functions used here need none of it
    for name in ("mlflow", "librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import css.training.train as T
    import css.training.losses as LS
    desc = W.ModelDesc(num_blocks=2)
    st = W.apply_golden_recipe(W.portable_state_dict(desc, 13))
    model = G.build_reference_model(desc, st)
    mix, gt_spk0, gt_noise0 = loss_inputs()
    B, n, M = mix.shape
    S = gt_spk0.shape[1]
    # the batch dict of css/training/simulated_dataset.py: ground truths carry a microphone axis; only mic 0 is used
    gt_spk_full = np.zeros((B, n, M, S), np.float32)
    gt_spk_full[:, :, 0, :] = np.moveaxis(gt_spk0, 1, 2)
    gt_noise_full = np.zeros((B, n, M), np.float32)
    gt_noise_full[:, :, 0] = gt_noise0
    batch = {"mixture": torch.from_numpy(mix), "gt_spk_direct_early_echoes": torch.from_numpy(gt_spk_full),
             "gt_noise": torch.from_numpy(gt_noise_full)}
    out = {"seed": 7, "batch": B, "n": n, "weights_seed": 13, "num_blocks": 2, "cases": []}
    for loss_name, base, clip, nw in (("masked_mag", "l1", True, 1.0), ("masked_mag", "mse", False, 0.5),
                                      ("mask", "l1", True, 1.0), ("mask", "mse", True, 2.0)):
        base_fn = LS.l1_loss if base == "l1" else LS.mse_loss
        pit = LS.PitWrapper(base_fn)
        seen = {}
        orig = pit.forward

        def fwd(preds, targets, orig=orig, seen=seen):
            loss, perms = orig(preds, targets)
            seen["spk_loss"] = [float(x) for x in loss]
            seen["perms"] = [[int(i) for i in p] for p in perms]
            return loss, perms

        pit.forward = fwd
        cfg = types.SimpleNamespace(clip_gt_to_mixture=clip, calc_side_info=False, loss_name=loss_name, noise_weight=nw)
        with torch.no_grad():
            loss, _ = T._calc_loss(batch, model, base_fn, pit, cfg)
        out["cases"].append({"loss_name": loss_name, "base_loss": base, "clip_gt_to_mixture": clip, "noise_weight": nw,
                             "loss": float(loss), **seen})
        print(out["cases"][-1])
    with open(os.path.join(HERE, "val_loss.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Second set of golden fixtures, again produced by running the REAL reference in the build container (see
gen_golden.py for the rules: the reference is imported in place, nothing of it is copied; only seeds and output
tensors are written).

    python tests/golden/gen_golden_r2.py       # writes variants_mc.npz, e2e60_mc.npz, e2e60_sc.npz

  variants_mc.npz   the non-default CssCfg branches of css/css.py on a 6.05 s, 4-segment input (ragged tail):
                    normalize_segment_power (css.py:233-247), mc_mask_floor_db -6 / -12 (css.py:222-227), mc_mvdr=False
                    (css.py:211-221), stitching_loss='mse' (css.py:263), stitching_input='separation_result'
                    (css.py:267-271) -- all on the SAME reference masks (replayed), so one oracle / HIP mask set serves
                    every branch -- and three other segmentations (3 s / 2 s, 4 s / 2 s, 2 s / 1 s) on an 11 s input
  e2e60_mc.npz      BASELINE.json configs[1] at full size: the 60 s 7-ch meeting through the reference; decisions as
                    SHA-256 (+ the packed winner-take-all map), waveforms decimated and as windows
  e2e60_sc.npz      configs[2] at full size: channel 0 of the same meeting through the single-channel model
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up sys.path / stubs, imports the reference as G.RC)

import torch  # noqa: E402

RC, W, SYN, O = G.RC, G.W, G.SYN, G.O


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack2(idx: np.ndarray) -> np.ndarray:
    """values 0..3 -> 2 bits each (little end first)"""
    flat = idx.reshape(-1).astype(np.uint8)
    flat = np.concatenate([flat, np.zeros((-len(flat)) % 4, np.uint8)])
    q = flat.reshape(-1, 4)
    return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)


def wta_of(masks):
    return np.stack([np.argmax(np.concatenate([m["spk_masks"][0], m["noise_masks"][0]], -1), -1) for m in masks]).astype(np.uint8)


def outputs(prefix, wavs, side, pit, k=4):
    w = np.stack(wavs)
    return {
        f"{prefix}_pit_perm": np.array([p for _, p in pit], np.int32).reshape(-1, 3),
        f"{prefix}_activity_final": np.packbits(side["activity_final"].numpy()[0]),
        f"{prefix}_activity_b": np.packbits(side["activity_b"].numpy()),
        f"{prefix}_activity_shape": np.array(side["activity_b"].shape),
        f"{prefix}_wav_windows": G.take_windows(w, k),
        f"{prefix}_wav_rms": np.array([np.sqrt(np.mean(x.astype(np.float64) ** 2)) for x in wavs]),
        f"{prefix}_wav_len": len(wavs[0]),
    }


class Replay(torch.nn.Module):
    """Separator that replays stored masks (the reference's own, captured by a Tap) -- css.py sees a separator."""

    def __init__(self, inner, stored):
        super().__init__()
        self.inner, self.stored, self.i = inner, stored, 0

    def stft(self, s):
        return self.inner.stft(s)

    def istft(self, s):
        return self.inner.istft(s)

    def separate(self, stft_seg):
        m = self.stored[self.i]
        self.i += 1
        return {"spk_masks": torch.from_numpy(m["spk_masks"].copy()), "noise_masks": torch.from_numpy(m["noise_masks"].copy())}


def main():
    torch.manual_seed(0)
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    mix60 = SYN.synth_meeting(60.0, 7, seed=1)
    report = {}

    # ------------------------------------------------------------------ CssCfg branches on replayed masks
    base = dict(show_progressbar=False, activity_th=0.3)
    mix_o = mix60[:, 16000:16000 + 6 * 16000 + 777]                    # 4 segments, ragged tail
    wavs, side, tap, _ = G.run_reference(model, mix_o, RC.CssCfg(**base))
    stored = [dict(m) for m in tap.masks]
    out = {"opt_offset": 16000, "opt_samples": mix_o.shape[1], "opt_wta_index": wta_of(tap.masks)}
    out.update(outputs("opt_default", wavs, side, tap.pit))
    variants = {
        "mse": dict(stitching_loss="mse"),
        "sepres": dict(stitching_input="separation_result"),
        "pnorm": dict(normalize_segment_power=True),
        "nomvdr_floor6": dict(mc_mvdr=False, mc_mask_floor_db=-6.0),
        "floor12": dict(mc_mask_floor_db=-12.0),
        "floor6": dict(mc_mask_floor_db=-6.0),
    }
    for name, kw in variants.items():
        rep = Replay(model, stored).eval()
        with G.Tap(rep) as t2:
            w2, s2 = RC.separate_and_stitch(mix_o, rep, 16000, torch.device("cpu"), RC.CssCfg(**base, **kw))
        out.update(outputs(f"opt_{name}", w2, s2, t2.pit))
        report[f"opt_{name}_vs_default_relrms"] = [G.rel_rms(w2[k], wavs[k]) for k in range(3)]
    # ------------------------------------------------------------------ other segmentations (own masks)
    mix_s = mix60[:, 32000:32000 + 11 * 16000 + 300]
    out["seg_offset"], out["seg_samples"] = 32000, mix_s.shape[1]
    for seg, hop in ((3.0, 2.0), (4.0, 2.0), (2.0, 1.0)):
        name = f"seg{int(seg)}{int(hop)}"
        w3, s3, t3, _ = G.run_reference(model, mix_s, RC.CssCfg(**base, segment_size_sec=seg, hop_size_sec=hop))
        out.update(outputs(name, w3, s3, t3.pit))
        out[f"{name}_wta_index"] = pack2(wta_of(t3.masks))
        out[f"{name}_wta_shape"] = np.array(wta_of(t3.masks).shape)
        out[f"{name}_segment_frames"] = int(s3["segment_frames"])
        out[f"{name}_masks_spk_seg0"] = t3.masks[0]["spk_masks"][0, ::8, ::4]
    np.savez_compressed(os.path.join(HERE, "variants_mc.npz"), **out)

    # ------------------------------------------------------------------ 60 s, multi-channel (configs[1])
    cfg = RC.CssCfg(**base)
    w60, s60, t60, dt = G.run_reference(model, mix60, cfg)
    report["ref_60s_mc_wall_s"] = dt
    wta60 = wta_of(t60.masks)
    e = {"mix_seed": 1, "mix_seconds": 60.0, "num_segments": len(t60.masks),
         "sha_pit_perm": sha(np.array([p for _, p in t60.pit], np.int32)),
         "sha_activity_b": sha(np.packbits(s60["activity_b"].numpy())),
         "sha_activity_final": sha(np.packbits(s60["activity_final"].numpy()[0])),
         "pit_perm": np.array([p for _, p in t60.pit], np.int32),
         "activity_final": np.packbits(s60["activity_final"].numpy()[0]),
         "activity_shape": np.array(s60["activity_b"].shape),
         "wta_packed": pack2(wta60), "wta_shape": np.array(wta60.shape),
         "wav_dec": np.stack(w60)[:, ::256],
         "wav_windows": G.take_windows(np.stack(w60), 4),
         "wav_rms": np.array([np.sqrt(np.mean(x.astype(np.float64) ** 2)) for x in w60]),
         "wav_len": len(w60[0]),
         "mask_stitched": s60["mask_stitched"].numpy()[0, ::32, ::16]}
    np.savez_compressed(os.path.join(HERE, "e2e60_mc.npz"), **e)

    # ------------------------------------------------------------------ 60 s, single channel (configs[2])
    desc_sc = W.ModelDesc.sc_v1()
    st_sc = W.portable_state_dict(desc_sc, 0)
    model_sc = G.build_reference_model(desc_sc, st_sc)
    mix_sc = mix60[:, :, :1].copy()
    ws, ss, ts, dt = G.run_reference(model_sc, mix_sc, cfg)
    report["ref_60s_sc_wall_s"] = dt
    e = {"mix_seed": 1, "mix_seconds": 60.0, "num_segments": len(ts.masks),
         "sha_pit_perm": sha(np.array([p for _, p in ts.pit], np.int32)),
         "sha_activity_final": sha(np.packbits(ss["activity_final"].numpy()[0])),
         "pit_perm": np.array([p for _, p in ts.pit], np.int32),
         "activity_final": np.packbits(ss["activity_final"].numpy()[0]),
         "activity_shape": np.array(ss["activity_b"].shape),
         "wav_dec": np.stack(ws)[:, ::256],
         "wav_windows": G.take_windows(np.stack(ws), 4),
         "wav_rms": np.array([np.sqrt(np.mean(x.astype(np.float64) ** 2)) for x in ws]),
         "wav_len": len(ws[0])}
    np.savez_compressed(os.path.join(HERE, "e2e60_sc.npz"), **e)

    # (the css_inference triple moved to gen_golden_r4.py: a longer, full-scale session)
    with open(os.path.join(HERE, "golden_report_r2.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()

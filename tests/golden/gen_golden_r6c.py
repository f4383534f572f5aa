#!/usr/bin/env python3
"""Round-6 fixture for erasures and nulls, produced by running the REAL reference in the build container (rules as in
gen_golden.py: the reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r6c.py            (about five minutes)

  nulls_r6.npz   what a recording with DIGITAL ZEROS does to the reference (css/css.py:110-338 with mvdr_util.py:58-75's 1e-15
                 diagonal loading), on 24 s of the configs[1] meeting and weights:
                   all_zero  8 s of exact zeros on every channel (a muted array)
                   gap       samples [8 s, 15.5 s) exactly zero on every channel: three segments entirely silent, four partly
                   dead_mic  channel 3 exactly zero throughout (a dead microphone: every covariance is rank deficient)
                 Kept per case: permutations, both activity maps, the waveforms every 16th sample, where they are exactly zero,
                 and -- because a talker who is silent in a segment makes the reference's complex64 solve return noise
                 (hazards; gen_golden_r4.py) -- per second how far the ORACLE's complex64 and complex128 beamformers are
                 from each other and from the reference: the seconds where the answer is defined at all.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

import torch  # noqa: E402

RC, W, SYN, O = G.RC, G.W, G.SYN, G.O
SECONDS = 24.0


def cases():
    base = SYN.synth_meeting(SECONDS, 7, seed=1)
    gap = base.copy()
    gap[:, 8 * 16000:int(15.5 * 16000)] = 0.0
    dead = base.copy()
    dead[:, :, 3] = 0.0
    return {"all_zero": np.zeros((1, 8 * 16000, 7), np.float32), "gap": gap, "dead_mic": dead}


def per_second(a, b, scale):
    n = a.shape[1]
    out = []
    for s in range(int(np.ceil(n / 16000))):
        lo, hi = s * 16000, min((s + 1) * 16000, n)
        out.append(np.sqrt(((a[:, lo:hi].astype(np.float64) - b[:, lo:hi]) ** 2).mean(axis=1)) / scale)
    return np.array(out)      # [seconds, streams]


def main():
    torch.manual_seed(0)
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    params = O.ConformerParams(st)
    cfg = RC.CssCfg(show_progressbar=False, activity_th=0.3)
    e, rep = {}, {}
    for name, mix in cases().items():
        w, side, tap, dt = G.run_reference(model, mix, cfg)
        w = np.stack(w)
        scale = np.maximum(np.sqrt((w.astype(np.float64) ** 2).mean(axis=1)), 1e-30)
        o64, oside = O.separate_and_stitch(mix, params, 16000, O.OracleCssCfg(activity_th=0.3))
        o128, _ = O.separate_and_stitch(mix, params, 16000, O.OracleCssCfg(activity_th=0.3), mvdr_cplx=np.complex128)
        o64, o128 = np.stack(o64), np.stack(o128)
        assert [tuple(p) for p in oside["perms"][1:]] == [p for _, p in tap.pit]
        e[f"{name}_samples"] = mix.shape[1]
        e[f"{name}_pit_perm"] = np.array([p for _, p in tap.pit], np.int32)
        e[f"{name}_activity_final"] = np.packbits(side["activity_final"].numpy()[0])
        e[f"{name}_activity_b"] = np.packbits(side["activity_b"].numpy())
        e[f"{name}_activity_shape"] = np.array(side["activity_b"].shape)
        e[f"{name}_wav_dec16"] = w[:, ::16].copy()
        e[f"{name}_wav_is_zero_dec16"] = np.packbits(w[:, ::16] == 0)
        e[f"{name}_wav_zero_run"] = np.array([int(np.flatnonzero((w != 0).any(axis=0))[0]) if (w != 0).any() else -1,
                                              int(np.flatnonzero((w != 0).any(axis=0))[-1]) if (w != 0).any() else -1])
        e[f"{name}_wav_rms"] = np.sqrt((w.astype(np.float64) ** 2).mean(axis=1))
        e[f"{name}_wav_len"] = w.shape[1]
        e[f"{name}_oracle_c64_vs_c128_per_second"] = per_second(o64, o128, scale)
        e[f"{name}_oracle_c64_vs_reference_per_second"] = per_second(o64, w, scale)
        zero = (w == 0).all(axis=0)
        rep[name] = {"reference_wall_s": dt, "finite": bool(np.isfinite(w).all()), "wav_rms": [float(x) for x in e[f"{name}_wav_rms"]],
                     "exactly_zero_samples": int(zero.sum()), "perms_non_identity": int((e[f"{name}_pit_perm"] != np.arange(3)).any(axis=1).sum()),
                     "activity_final_open_fraction": float(side["activity_final"].numpy().mean()),
                     "seconds_where_c64_and_c128_agree_within_1e-4": int((e[f"{name}_oracle_c64_vs_c128_per_second"].max(axis=1) < 1e-4).sum()),
                     "seconds": int(e[f"{name}_oracle_c64_vs_c128_per_second"].shape[0])}
        print(name, rep[name], flush=True)
    np.savez_compressed(os.path.join(HERE, "nulls_r6.npz"), **e)
    path = os.path.join(HERE, "golden_report_r6.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old["nulls_r6"] = rep
    json.dump(old, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

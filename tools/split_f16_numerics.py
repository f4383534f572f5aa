#!/usr/bin/env python3
"""CPU experiment: how accurate is a 3-product split-f16 GEMM (hi*hi + hi*lo + lo*hi, f32 accumulate) for the
Conformer's linear layers, compared with plain f32 and measured against a float64 run of the same network?

    python tools/split_f16_numerics.py [n_segments]

Emulates the operand split in numpy (f16 values are exactly representable in f32, products of two f16 values
are exact in f32, accumulation in f32 by OpenBLAS) by replacing the oracle's `_linear`.  Build-container
experiment only (imports oracle/, like the tests do)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import importlib

import css_oracle as O

W = importlib.import_module("notsofar1_challenge_amd.weights")
SYN = importlib.import_module("notsofar1_challenge_amd.synth")


def split(x, dtype, scale_lo):
    hi = x.astype(dtype).astype(np.float32)
    lo = ((x - hi) * np.float32(scale_lo)).astype(dtype).astype(np.float32)
    return hi, lo


def make_linear(dtype, scale_lo, four=False):
    inv = np.float32(1.0 / scale_lo)

    def lin(x, w, b):
        xh, xl = split(x.astype(np.float32), dtype, scale_lo)
        wh, wl = split(w.astype(np.float32), dtype, scale_lo)
        main = xh @ wh.T
        corr = xh @ wl.T + xl @ wh.T
        y = main + corr * inv
        if four:
            y = y + (xl @ wl.T) * (inv * inv)
        return y + b
    return lin


def main():
    nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    mix = SYN.synth_meeting(60, 7, seed=1)[0][: 16000 * 20]
    st = O.stft(np.ascontiguousarray(mix, dtype=np.float32))  # [F, T, C]
    p32 = O.ConformerParams(state, np.float32)
    p64 = O.ConformerParams(state, np.float64)
    T = 186
    orig = O._linear
    rows = {}
    for s in range(nseg):
        seg = np.ascontiguousarray(st[:, s * 93:s * 93 + T, :])  # [F, T, C]
        feat = O.features(seg, np.float32)
        O._linear = orig
        truth = O.conformer_forward(p64, feat.astype(np.float64))
        variants = {
            "f32": orig,
            "f16x3 (lo unscaled)": make_linear(np.float16, 1.0),
            "f16x3 (lo * 2^11)": make_linear(np.float16, 2048.0),
            "f16x4 (lo * 2^11)": make_linear(np.float16, 2048.0, four=True),
        }
        for name, fn in variants.items():
            O._linear = fn
            m = O.conformer_forward(p32, feat)
            d = (m.astype(np.float64) - truth)
            rows.setdefault(name, []).append((np.abs(d).max(), np.sqrt((d ** 2).mean())))
        O._linear = orig
    print(f"mask error vs float64 network over {nseg} segments (max abs, rms):")
    for name, r in rows.items():
        r = np.array(r)
        print(f"  {name:24s} max {r[:, 0].max():.3e}   rms {np.sqrt((r[:, 1] ** 2).mean()):.3e}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Print the measured parity distances (not just pass/fail) of the HIP path against the reference fixtures, in
both Linear-layer modes.  GPU box only:   python tools/parity_margins.py
Uses the same fixtures and formulas as tests/test_hip_parity.py."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_rms, take_windows  # noqa: E402

pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)  # noqa: E731
F, T, S = 257, 186, 3


def main():
    W, L, CSS = pkg("weights"), pkg("_lib"), pkg("css")
    G = os.path.join(ROOT, "tests", "golden")
    cal = np.load(os.path.join(G, "calib_mc.npz"))
    desc = W.ModelDesc.mc_v1()
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"],
                               head_gain=float(cal["head_gain"]), input_gain=float(cal["input_gain"]))
    g = np.load(os.path.join(G, "e2e_mc.npz"))
    mix = pkg("synth").synth_meeting(60.0, 7, seed=1)[:, :20 * 16000]
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=64)
    h = sep.handle
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    nseg = int(g["num_segments"])
    for mode in ("exact_f32", "split_f16"):
        h.set_linear_mode(mode)
        wav = h.run(mix[0], run_cfg)
        m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
        flips = int((np.argmax(m, axis=0).transpose(1, 0, 2) != g["wta_index"]).sum())
        free = max(rel_rms(take_windows(wav)[k], g["wav_windows"][k]) for k in range(S))
        perms_ok = [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["pit_perm"]]
        h.begin(mix[0], mix.shape[1], 7, run_cfg)
        TL = h.get_plan().mix_frames
        h.write(L.BUF_WTA_OVERRIDE, g["wta_index"])
        h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
        h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
        w2 = h.read(L.BUF_WAV)
        forced = max(rel_rms(take_windows(w2)[k], g["wav_windows"][k]) for k in range(S))
        forced_dec = max(rel_rms(w2[k, ::64], g["wav_dec"][k]) for k in range(S))
        ms = np.abs(h.read(L.BUF_MASK_ST).transpose(1, 2, 0)[::16, ::8] - g["mask_stitched"]).max()
        print(f"{mode:10s} WTA flips vs reference {flips} of {g['wta_index'].size};  permutations equal {perms_ok};  "
              f"stitched-mask max|diff| {ms:.2e};  waveform rel-RMS vs reference: free-running {free:.2e}, "
              f"on the reference's WTA decisions {forced:.2e} (windows) / {forced_dec:.2e} (decimated)")
    sep.close()


if __name__ == "__main__":
    main()

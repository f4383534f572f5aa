"""How much the any-length kernels (segments beyond 512 frames) cost: one 60 s, 7-channel meeting through the v1.0-MC model
with 3 s / 8 s / 10 s / 20 s segments (hop = half a segment), synchronous css_run calls, milliseconds per meeting.
    python tools/long_segment_timing.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import pkg  # noqa: E402

CSS, W, SYN = pkg("css"), pkg("weights"), pkg("synth")
desc = W.ModelDesc.mc_v1()
st = W.apply_golden_recipe(W.portable_state_dict(desc, 1))
sep = pkg("separator").HipSeparator(st, None, device=0)
mix = SYN.synth_meeting(60.0, 7, seed=1)
print("| segment / hop | frames | segments | ms per 60 s meeting | x real time |\n|---|---:|---:|---:|---:|")
for seg in (3.0, 8.0, 10.0, 20.0):
    cfg = CSS.CssCfg(show_progressbar=False, segment_size_sec=seg, hop_size_sec=seg / 2)
    for _ in range(2):
        w, side = CSS.separate_and_stitch(mix, sep, 16000, "cuda:0", cfg)
    t0 = time.perf_counter()
    for _ in range(5):
        w, side = CSS.separate_and_stitch(mix, sep, 16000, "cuda:0", cfg)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"| {seg:g} s / {seg / 2:g} s | {side['segment_frames']} | {int(sep.handle.get_plan().num_segments)} | {ms:.1f} | {60e3 / ms:.0f} |")
sep.close()

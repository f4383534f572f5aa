#!/usr/bin/env python3
"""Writes what examples/c_host.c reads: model.bin (CssModelDesc + int64 count + the float32 weight blob of
weights.py::pack_blob) and pcm.f32 (a synthetic 7-channel recording, [n][7] float32).  A deployment exports its checkpoint the
same way: separator.load_css_model's state dict through weights.pack_blob.

    python examples/export_for_c_host.py <dir> [seconds]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
weights, synth, _lib = pkg("weights"), pkg("synth"), pkg("_lib")

out = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
os.makedirs(out, exist_ok=True)
desc = weights.ModelDesc.mc_v1()
state = weights.apply_golden_recipe(weights.portable_state_dict(desc, 0))
blob = np.ascontiguousarray(weights.pack_blob(state, desc)[0], dtype=np.float32)
with open(os.path.join(out, "model.bin"), "wb") as f:
    f.write(bytes(_lib.make_desc(desc)))
    f.write(np.int64(blob.size).tobytes())
    f.write(blob.tobytes())
np.ascontiguousarray(synth.synth_meeting(seconds, 7, seed=1)[0], dtype=np.float32).tofile(os.path.join(out, "pcm.f32"))
print(f"{out}/model.bin ({blob.size} floats), {out}/pcm.f32 ({seconds:g} s x 7 channels)")

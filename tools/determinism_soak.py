import importlib, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc.mc_v1()
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0))
mix = SYN.synth_meeting(60.0, 7, seed=1); pcm = L.pinned_copy(np.ascontiguousarray(mix[0])); n = pcm.shape[0]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128); h = sep.handle
ref = h.run(pcm, run_cfg).copy(); mref = h.read(L.BUF_MASKS).copy()
bad = 0
for lanes in (3, 2, 4, 1):
    h.set_lanes(lanes)
    for r in range(8):
        w = h.run(pcm, run_cfg)
        m = h.read(L.BUF_MASKS)
        if not (np.array_equal(w, ref) and np.array_equal(m, mref)):
            bad += 1; print("MISMATCH lanes", lanes, "rep", r, np.abs(m - mref).max())
print("repeats with mismatches:", bad, "of 32")
sep.close()

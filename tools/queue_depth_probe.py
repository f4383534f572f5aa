import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc.mc_v1()
cal = np.load('/root/repo/tests/golden/calib_mc.npz')
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(60.0, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128); h = sep.handle
plan = L.plan(desc, run_cfg, n)
pcm = L.pinned_copy(np.ascontiguousarray(mix[0])); outs = [L.pinned_empty((3, int(plan.n_out)), np.float32) for _ in range(2)]
def q(K):
    t0 = time.perf_counter()
    for k in range(K): h.run_enqueue(pcm, run_cfg, outs[k % 2])
    t1 = time.perf_counter(); h.wait(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K, 1e3 * (t1 - t0) / K
print("warm-up queue of 3:", q(3))
for r in range(5): print("queue of 20: ms per pass %.3f (host enqueue %.3f per pass)" % q(20))
for K in (2, 4, 8, 40): print(K, "passes: %.3f (enqueue %.3f)" % q(K))
sep.close()

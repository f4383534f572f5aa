import importlib, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc(num_blocks=2)
st = W.apply_golden_recipe(W.portable_state_dict(desc, 21))
mix = SYN.synth_meeting(60.0, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(st, None, device=0, max_batch_segments=64); h = sep.handle
plan = L.plan(desc, run_cfg, n)
pcm = np.ascontiguousarray(mix[0])
if len(sys.argv) > 1 and sys.argv[1] == 'pinned': pcm = L.pinned_copy(pcm)
pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda(); wd = torch.empty((3, int(plan.n_out)), device="cuda")
def dev():
    h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out)); torch.cuda.synchronize(); return wd.cpu().numpy().copy(), h.read(L.BUF_X).copy()
def host():
    w = h.run(pcm, run_cfg).copy(); return w, h.read(L.BUF_X).copy()
a, xa = dev(); b, xb = dev(); c, xc = host(); d, xd = host()
print("dev==dev", np.array_equal(a, b), "X", np.array_equal(xa, xb))
print("host==host", np.array_equal(c, d), "X", np.array_equal(xc, xd))
print("dev==host", np.array_equal(a, c), "X", np.array_equal(xa, xc))
if not np.array_equal(xc, xd):
    bad = np.argwhere(xc != xd); print("X differs at (c, row, t):", bad[:3], bad[-3:], len(bad))
if not np.array_equal(xa, xc):
    bad = np.argwhere(xa != xc); print("dev vs host X differs:", bad[:3], bad[-3:], len(bad))
sep.close()

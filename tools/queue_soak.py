import importlib, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc.mc_v1()
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0))
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=32); h = sep.handle
rs = np.random.RandomState(7)
base = SYN.synth_meeting(95.0, 7, seed=3)[0]
sess = []
for k in range(40):
    n = int(rs.uniform(3.1, 90.0) * 16000) + int(rs.randint(0, 255))
    off = int(rs.randint(0, base.shape[0] - n))
    g = float(10 ** rs.uniform(-2.5, 0.5))
    sess.append(L.pinned_copy(np.ascontiguousarray(base[off:off + n] * g)))
refs = [h.run(p, run_cfg).copy() for p in sess]
for lanes in (3, 2, 4):
    h.set_lanes(lanes)
    outs = [h.run_enqueue(p, run_cfg, L.pinned_empty(r.shape, np.float32)) for p, r in zip(sess, refs)]
    h.wait()
    bad = [k for k, (o, r) in enumerate(zip(outs, refs)) if not np.array_equal(o, r)]
    print("lanes", lanes, "sessions", len(sess), "mismatching:", bad)
sep.close()

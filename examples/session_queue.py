#!/usr/bin/env python3
"""A queue of sessions on one GPU (css_run_enqueue / css_wait): every session goes from float PCM in page-locked host
memory to separated waveforms in page-locked host memory; a session's PCIe legs run under its neighbours' kernels, and
consecutive sessions share mask-estimator batches (up to max_batch_segments segments per batch).

    python examples/session_queue.py [n_sessions]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
css, _lib, sepmod, weights, synth = (pkg(n) for n in ("css", "_lib", "separator", "weights", "synth"))

n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 8
desc = weights.ModelDesc.mc_v1()                                     # a real deployment: separator.load_css_model(dir)
state = weights.apply_golden_recipe(weights.portable_state_dict(desc, 0))
sep = sepmod.HipSeparator(state, None, device=0, max_batch_segments=128)
h = sep.handle
run_cfg = css.make_run_cfg(css.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)

# sessions of different lengths; inputs and outputs in page-locked memory (css_host_alloc)
sessions = []
for k in range(n_sessions):
    pcm = _lib.pinned_copy(np.ascontiguousarray(synth.synth_meeting(40.0 + 10.0 * (k % 4), 7, seed=k)[0]))
    plan = _lib.plan(desc, run_cfg, pcm.shape[0])
    sessions.append((pcm, _lib.pinned_empty((3, int(plan.n_out)), np.float32)))
h.run(max((p for p, _ in sessions), key=lambda p: p.shape[0]), run_cfg)   # warm-up: buffers sized for the longest session
for pcm, out in sessions:                                            # ... and one untimed queue: the handle allocates the
    h.run_enqueue(pcm, run_cfg, out)                                 # per-session buffers of a shared batch on first use
h.wait()

t0 = time.perf_counter()
views = [h.run_enqueue(pcm, run_cfg, out) for pcm, out in sessions]  # returns at once
h.wait()                                                             # every `out` is valid now
queued = time.perf_counter() - t0
t0 = time.perf_counter()
refs = [h.run(pcm, run_cfg).copy() for pcm, _ in sessions]           # the same sessions, one synchronous call each
sync = time.perf_counter() - t0
audio = sum(p.shape[0] for p, _ in sessions) / 16000.0
print(f"{n_sessions} sessions, {audio:.0f} s of 7-channel audio: queued {1e3 * queued:.1f} ms ({audio / queued:.0f} x real time), "
      f"one call per session {1e3 * sync:.1f} ms ({audio / sync:.0f} x real time); same bits: "
      f"{all(np.array_equal(v, r) for v, r in zip(views, refs))}")
if os.environ.get("DEBUG"):
    for k, (v, r) in enumerate(zip(views, refs)):
        bad = np.flatnonzero((v != r).any(axis=0))
        print(k, v.shape, "equal" if bad.size == 0 else f"{bad.size} samples differ, first {bad[0]}, last {bad[-1]}, max {np.abs(v - r).max():.3e}")
sep.close()

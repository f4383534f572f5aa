#!/usr/bin/env python3
"""Soak of the round-6 queue (tools only): sessions of mixed lengths and mixed edges (float PCM -> float waveforms, PCM16 planes ->
peak-normalised PCM16) go through css_run_enqueue / css_run_enqueue_pcm16 in a new random order every round, released with
css_wait_sessions (a rolling window) or css_wait; every output of every round is compared, bit for bit, with the session's own
synchronous css_run / css_run_pcm16.  A rare scheduling-dependent fault (the store hazard of profiles/r06_store_guard.txt wrote a
few hundred wrong elements in 11 M) shows up here as a mismatch count.

    python tools/queue_soak_r6.py [minutes per mode = 3] [modes = exact_f32,split_f16]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    modes = (sys.argv[2] if len(sys.argv) > 2 else "exact_f32,split_f16").split(",")
    W, SYN, CSS, SEP, L = (pkg(x) for x in ("weights", "synth", "css", "separator", "_lib"))
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    mk = lambda **kw: CSS.make_run_cfg(CSS.CssCfg(show_progressbar=False, **kw), 16000, 7)
    cfgs = [mk(activity_th=0.3), mk(activity_th=0.45), mk(activity_th=0.3, stitching_loss="mse")]
    rs = np.random.RandomState(66)
    base = SYN.synth_meeting(125.0, 7, seed=3)[0]
    for mode in modes:
        sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=256, linear_mode=mode)
        h = sep.handle
        sessions = []
        for k in range(36):
            seconds = 60.0 if k % 3 == 0 else float(rs.uniform(3.1, 120.0))
            n = int(seconds * 16000) + int(rs.randint(0, 255))
            off = int(rs.randint(0, base.shape[0] - n))
            gain = float(10 ** rs.uniform(-2.0, -0.3))
            cfg = cfgs[k % 3]
            q = np.clip(np.rint(base[off:off + n] * gain * 32768.0), -32768, 32767).astype(np.int16)
            if k % 2:
                blk = L.pinned_empty((7, n), np.int16)
                blk[:] = q.T
                planes = [blk[c] for c in range(7)]
                ref16, refpk = h.run_pcm16(planes, cfg)
                sessions.append(("pcm16", planes, cfg, ref16.copy(), refpk.copy(), blk))
            else:
                f32 = L.pinned_copy(np.ascontiguousarray(q.astype(np.float32) / np.float32(32768.0)))
                sessions.append(("float", f32, cfg, h.run(f32, cfg).copy(), None, None))
        audio_s = sum((s[1][0].shape[0] if s[0] == "pcm16" else s[1].shape[0]) / 16000.0 for s in sessions)
        print(f"[{mode}] {len(sessions)} sessions, {audio_s:.0f} s of audio per round, references by css_run / css_run_pcm16", flush=True)
        outs = [[(L.pinned_empty(s[3].shape, s[3].dtype), L.pinned_empty((3,), np.float32)) for s in sessions] for _ in range(2)]
        checked = bad = rounds = 0
        first_bad = []
        t_end = time.time() + 60.0 * minutes
        t0 = time.time()
        while time.time() < t_end:
            order = rs.permutation(len(sessions))
            o = outs[rounds % 2]
            rolling = mode == "exact_f32" and rounds % 2 == 0
            for k in order:
                o[k][0][...] = 0
            done = 0
            for pos, k in enumerate(order):
                kind, src, cfg = sessions[k][:3]
                if kind == "pcm16":
                    h.run_enqueue_pcm16(src, cfg, o[k][0], o[k][1])
                else:
                    h.run_enqueue(src, cfg, o[k][0])
                if rolling and pos + 1 - done >= 16:
                    h.wait_sessions(done + 8)
                    for j in order[done:done + 8]:      # released while the younger ones are still on the device
                        ok = np.array_equal(o[j][0], sessions[j][3]) and (sessions[j][4] is None or np.array_equal(o[j][1], sessions[j][4]))
                        checked += 1
                        if not ok:
                            bad += 1
                            first_bad.append((rounds, int(j), "rolling"))
                    done += 8
            h.wait()
            for j in order[done:]:
                ok = np.array_equal(o[j][0], sessions[j][3]) and (sessions[j][4] is None or np.array_equal(o[j][1], sessions[j][4]))
                checked += 1
                if not ok:
                    bad += 1
                    first_bad.append((rounds, int(j), "wait"))
            rounds += 1
        dt = time.time() - t0
        print(f"[{mode}] {rounds} rounds in {dt:.0f} s ({audio_s * rounds / dt:.0f} x real time incl. the comparisons): "
              f"{checked} session outputs compared bit for bit with their synchronous results, {bad} differ {first_bad[:8]}", flush=True)
        sep.close()


if __name__ == "__main__":
    main()

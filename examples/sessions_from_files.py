#!/usr/bin/env python3
"""The CSS leg of inference_pipeline/inference.py:59-63 on one GPU, files to files: a model directory as the reference's training
loop leaves it (one *.yaml + one *.pt, css/helpers.py:14-37), N sessions of 7 mono PCM16 wav files each, and
pipeline.css_sessions -- the sessions go through the library's queue (css_run_enqueue_pcm16 / css_wait_sessions: shared
estimator batches, both wav edges on the device) while worker threads decode the next sessions and write the finished ones.
Every session ends as css_inference leaves it: <out>/css_inference/<session_id>/{input_mixture,sep_stream0..2}.wav and a
`sep_wav_file_names` column for the ASR / diarization legs.

    python examples/sessions_from_files.py [n_sessions] [seconds]"""
import importlib, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
css, pipeline, weights, synth, wavio = (pkg(n) for n in ("css", "pipeline", "weights", "synth", "wavio"))
import pandas as pd
import torch
import yaml

n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
tmp = tempfile.mkdtemp(prefix="css_sessions_example_")
try:
    # ---- a model directory (here: seeded weights of the v1.0 multi-channel architecture; a deployment points at its checkpoint)
    desc = weights.ModelDesc.mc_v1()
    state = weights.apply_golden_recipe(weights.portable_state_dict(desc, 0))
    mdir = os.path.join(tmp, "models", "notsofar", "conformer1.0", "mc")
    os.makedirs(mdir)
    torch.save({"model": {"module." + k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}}, os.path.join(mdir, "model.pt"))
    with open(os.path.join(mdir, "train_cfg.yaml"), "w") as f:
        yaml.safe_dump({"train_dir": "x", "val_dir": "x", "out_dir": "x",
                        "conformer_css_cfg": {"nnet_conf": {"conformer_conf": {"attention_dim": 512, "attention_heads": 8, "num_blocks": 18,
                                                                               "dropout_rate": 0.0}}}}, f)
    # ---- the sessions: 7 mono 16-bit wav files each (channel 0 = the centre microphone)
    rows = []
    for i in range(n_sessions):
        mix = synth.synth_meeting(seconds, 7, seed=100 + i)[0]
        names = []
        for c in range(7):
            p = os.path.join(tmp, f"session{i:03d}_ch{c}.wav")
            wavio.write_pcm16_samples(p, np.clip(np.rint(mix[:, c] * 0.1 * 32768.0), -32768, 32767).astype(np.int16), 16000)
            names.append(p)
        rows.append({"wav_file_names": names, "session_id": f"session{i:03d}", "is_mc": True})
    sessions = pd.DataFrame(rows)
    cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)           # configs/inference/inference_v1.yaml:17
    for attempt in ("first call (loads the model, sizes the page-locked pools)", "second call"):
        stats = {}
        t0 = time.perf_counter()
        out = pipeline.css_sessions(os.path.join(tmp, "out_" + attempt.split()[0]), os.path.join(tmp, "models"), sessions, cfg, stats=stats)
        dt = time.perf_counter() - t0
        print(f"{attempt}: {n_sessions} sessions x {seconds:g} s in {dt:.2f} s = {n_sessions * seconds / dt:.0f} x real time, files to files, "
              f"model load included (in css_wait {stats['in_css_wait_s']:.2f} s; with the model resident -- css_sessions(..., separators=...) -- "
              f"and 48 sessions bench.py measures 6 300 x)")
    print(out[["session_id", "sep_wav_file_names"]].head(3).to_string())
finally:
    shutil.rmtree(tmp, ignore_errors=True)

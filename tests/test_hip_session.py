"""Module-level drop-in on the GPU: css_inference (css/css.py:51-107) and the session loop, from wav files
and a checkpoint directory (one *.yaml + one *.pt, keys prefixed `module.`, css/helpers.py:14-37) to
`css_inference/<session_id>/sep_stream{i}.wav` and the `sep_wav_file_names` column."""
import os

import numpy as np
import pytest

import css_oracle as O
from conftest import pkg, rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_models(tmp_path_factory):
    """A 2-block MC model and a 1-block SC model saved the way the reference's training loop saves them."""
    import torch
    import yaml
    w = pkg("weights")
    root = tmp_path_factory.mktemp("css_models")
    out = {}
    for kind, desc, blocks in (("mc", w.ModelDesc(num_blocks=2), 2), ("sc", w.ModelDesc(num_mics=1, in_features=257, num_blocks=1), 1)):
        d = root / "notsofar" / "conformer1.0" / kind
        d.mkdir(parents=True)
        st = w.apply_golden_recipe(w.portable_state_dict(desc, 21)) if kind == "mc" else w.portable_state_dict(desc, 22)
        ckpt = {"model": {"module." + k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}}
        torch.save(ckpt, d / "model.pt")
        cfg = {"train_dir": "x", "val_dir": "x", "out_dir": "x",
               "conformer_css_cfg": {"nnet_conf": {"conformer_conf": {"attention_dim": 512, "attention_heads": 8,
                                                                      "num_blocks": blocks, "dropout_rate": 0.0}}}}
        if kind == "sc":
            cfg["conformer_css_cfg"]["extractor_conf"] = {"ipd_index": ""}
            cfg["conformer_css_cfg"]["nnet_conf"]["in_features"] = 257
        with open(d / "train_cfg.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        out[kind] = (st, desc)
    return str(root), out


def _write_session(tmp_path, wavio, mix, sid, is_mc):
    names = []
    for c in range(mix.shape[2]):
        p = tmp_path / f"{sid}_ch{c}.wav"
        wavio.write_wav(p, mix[0, :, c], 16000, max_norm=False)
        names.append(str(p))
    return {"wav_file_names": names, "session_id": sid, "is_mc": is_mc}


def test_css_inference_from_files(tmp_path, tiny_models):
    import pandas as pd
    css, wavio, sep_mod, L = pkg("css"), pkg("wavio"), pkg("separator"), pkg("_lib")
    models_dir, models = tiny_models
    mix = (pkg("synth").synth_meeting(6.0, 7, seed=9) * 0.05).astype(np.float32)   # inside [-1, 1] for PCM16
    session = pd.Series(_write_session(tmp_path, wavio, mix, "MTG_1_plaza_0", True))
    cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)
    out = css.css_inference(str(tmp_path / "out"), models_dir, session, cfg, fetch_from_cache=False)
    names = out["sep_wav_file_names"]
    d = tmp_path / "out" / "css_inference" / "MTG_1_plaza_0"
    assert names == [str(d / f"sep_stream{i}.wav") for i in range(3)] and (d / "input_mixture.wav").exists()
    assert "sep_wav_file_names" not in session                       # the input Series is copied (css.py:70)

    # the files hold what separate_and_stitch returns for the PCM16-quantised input, peak-normalised to 0.99
    mixq, sr = wavio.load_audio(session.wav_file_names, is_mc=True)
    assert sr == 16000 and mixq.shape == mix.shape and np.abs(mixq - mix).max() <= 1.0 / 32768 + 1e-7
    st, desc = models["mc"]
    sep = sep_mod.HipSeparator(st, None, device=0)
    wavs, _ = css.separate_and_stitch(mixq, sep, 16000, "cuda:0", cfg)
    sep.close()
    for i in range(3):
        y, sr_i = wavio.read_wav(names[i])
        ref = wavs[i] * 0.99 / (np.max(np.abs(wavs[i])) + 1e-7)
        assert sr_i == 16000 and len(y) == len(ref)
        # PCM16 only: half an LSB of rounding plus the 32767 (write) vs 32768 (read) scale conventions
        assert np.abs(y - ref).max() <= 1.6 / 32768
    # ... and the oracle agrees with that model loaded from the checkpoint directory (masks injected from HIP
    # are not needed here: 2 blocks, benign MVDR)
    ow, _ = O.separate_and_stitch(mixq, O.ConformerParams(st), 16000, O.OracleCssCfg(activity_th=0.3),
                                  mvdr_cplx=np.complex128)
    assert max(rel_rms(wavs[k], ow[k]) for k in range(3)) < 5e-3    # free-running WTA decisions: flips allowed

    # cache hit returns the existing files (css.py:79-82); pass-through returns channel 0 (css.py:73-75)
    again = css.css_inference(str(tmp_path / "out"), models_dir, session, cfg, fetch_from_cache=True)
    assert [str(p) for p in again["sep_wav_file_names"]] == names
    pt = css.css_inference(str(tmp_path / "out"), models_dir, session, css.CssCfg(pass_through_ch0=True), False)
    assert pt["sep_wav_file_names"] == session.wav_file_names[:1]
    with pytest.raises(FileNotFoundError):
        css.css_inference(str(tmp_path / "out2"), str(tmp_path / "nope"), session, cfg, False)


def _session_rows(tmp_path, wavio):
    mix = (pkg("synth").synth_meeting(8.0, 7, seed=10) * 0.25).astype(np.float32)
    rows = [_write_session(tmp_path, wavio, mix[:, : 100000 + 1000 * i], f"S{i}_mc", True) for i in range(2)]
    rows.append(_write_session(tmp_path, wavio, mix[:, :72000, :1], "S2_sc", False))
    return rows


def _check_session_outputs(part, models, wavio, css, sep_mod):
    """Every row's files hold exactly what the handle returns for that session's PCM16 planes (bit for bit), and the
    oracle -- driven by its own masks, free-running -- agrees with them to the conditioning of these short clips."""
    cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)
    for _, row in part.iterrows():
        kind = "mc" if row.is_mc else "sc"
        st, desc = models[kind]
        raw = [wavio.read_wav_pcm16(p) for p in row.wav_file_names]
        sep = sep_mod.HipSeparator(st, None, device=0)
        try:
            run_cfg = css.make_run_cfg(cfg, 16000, len(raw), desc.frame_len, desc.frame_hop)
            direct, _ = sep.handle.run_pcm16([r[0] for r in raw], run_cfg)
        finally:
            sep.close()
        assert len(row.sep_wav_file_names) == 3
        mixq = np.stack([r[0].astype(np.float32) / np.float32(32768.0) for r in raw], axis=1)[None]
        ow, _ = O.separate_and_stitch(mixq, O.ConformerParams(st), 16000, O.OracleCssCfg(activity_th=0.3), mvdr_cplx=np.complex128)
        for i, f in enumerate(row.sep_wav_file_names):
            assert os.path.basename(f) == f"sep_stream{i}.wav" and os.path.basename(os.path.dirname(f)) == row.session_id
            got, sr = wavio.read_wav_pcm16(f)
            assert sr == 16000 and np.array_equal(got, direct[i]) and np.abs(got).max() == 32439
            ref = ow[i] * 0.99 / (np.max(np.abs(ow[i])) + 1e-7)
            y = got.astype(np.float64) / 32767.0
            gain = float(y @ ref / (ref @ ref))
            assert abs(gain - 1) < 0.05 and rel_rms(y, gain * ref) < 5e-2, (row.session_id, i, gain)


def test_session_loop_shards_by_session(tmp_path, tiny_models):
    """pipeline.css_sessions (inference.py:37-107's CSS leg): rank r takes every world-th session, one resident model
    per kind, each row produced by css_inference itself (PCM16 device edges); both virtual ranks, then the cache rule."""
    import pandas as pd
    css, wavio, pipe, sep_mod = pkg("css"), pkg("wavio"), pkg("pipeline"), pkg("separator")
    models_dir, models = tiny_models
    df = pd.DataFrame(_session_rows(tmp_path, wavio))
    cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)
    parts = [pipe.css_sessions(str(tmp_path / "o"), models_dir, df, cfg, rank=r, world=2, device="cuda:0") for r in range(2)]
    assert [list(p.session_id) for p in parts] == [["S0_mc", "S2_sc"], ["S1_mc"]]
    for p in parts:
        assert list(p.columns) == list(df.columns) + ["sep_wav_file_names"]
        _check_session_outputs(p, models, wavio, css, sep_mod)
    cached = pipe.css_sessions(str(tmp_path / "o"), "no_models_needed_for_a_cache_hit", df, cfg, fetch_from_cache=True, rank=0, world=1)
    assert [[os.path.basename(str(f)) for f in r] for r in cached.sep_wav_file_names] == [[f"sep_stream{i}.wav" for i in range(3)]] * 3


def _session_worker(rank, world, tmp, models_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    import pandas as pd
    css, pipe = pkg("css"), pkg("pipeline")
    df = pd.read_pickle(os.path.join(tmp, "sessions.pkl"))
    part = pipe.css_sessions(os.path.join(tmp, "o2"), models_dir, df, css.CssCfg(activity_th=0.3, show_progressbar=False))
    part.to_pickle(os.path.join(tmp, f"part{rank}.pkl"))


def test_session_loop_in_two_processes(tmp_path, tiny_models):
    """The same loop as torchrun would start it: two processes, RANK / WORLD_SIZE / LOCAL_RANK from the environment
    (utils/torch_utils.py:10-11), one GPU shared by both; together they cover every session exactly once."""
    import pandas as pd
    import torch.multiprocessing as mp
    css, wavio, sep_mod = pkg("css"), pkg("wavio"), pkg("separator")
    models_dir, models = tiny_models
    df = pd.DataFrame(_session_rows(tmp_path, wavio))
    df.to_pickle(tmp_path / "sessions.pkl")
    mp.spawn(_session_worker, args=(2, str(tmp_path), models_dir), nprocs=2, join=True)
    parts = [pd.read_pickle(tmp_path / f"part{r}.pkl") for r in range(2)]
    assert sorted(sum((list(p.session_id) for p in parts), [])) == sorted(df.session_id)
    for p in parts:
        _check_session_outputs(p, models, wavio, css, sep_mod)


def test_pcm16_wav_edges_on_device_are_bit_exact(tmp_path, tiny_models):
    """css_run_pcm16: int16 planes in, peak-normalised PCM16 out, both conversions on the device,
    must give exactly the samples the host path (load_audio -> css_run -> write_wav) puts into the files."""
    css, wavio, sep_mod = pkg("css"), pkg("wavio"), pkg("separator")
    _, models = tiny_models
    mix = (pkg("synth").synth_meeting(7.3, 7, seed=5) * 0.07).astype(np.float32)
    sess = _write_session(tmp_path, wavio, mix, "MTG_2", True)
    raw = [wavio.read_wav_pcm16(p) for p in sess["wav_file_names"]]
    assert all(r is not None and r[1] == 16000 for r in raw)
    mixq, _ = wavio.load_audio(sess["wav_file_names"], is_mc=True)
    assert np.array_equal(mixq[0, :, 3], raw[3][0].astype(np.float32) / np.float32(32768.0))
    st, desc = models["mc"]
    sep = sep_mod.HipSeparator(st, None, device=0)
    try:
        cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)
        run_cfg = css.make_run_cfg(cfg, 16000, 7)
        h = sep.handle
        pcm16, peaks = h.run_pcm16([r[0] for r in raw], run_cfg)
        wav = h.run(mixq[0], run_cfg)
        assert pcm16.dtype == np.int16 and pcm16.shape == wav.shape
        for i in range(3):
            assert peaks[i] == np.max(np.abs(wav[i]))
            y = wav[i] * 0.99 / (np.max(np.abs(wav[i])) + 1e-7)                     # utils/audio_utils.py:44-45
            assert y.dtype == np.float32
            expect = np.clip(np.rint(y.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(pcm16[i], expect)
            assert np.abs(pcm16[i]).max() == 32439                                   # rint(0.99 * 32767)
        with pytest.raises(ValueError):
            h.run_pcm16([raw[0][0], raw[1][0][:-1]], run_cfg)
    finally:
        sep.close()
    # written and read back through the wav codec
    wavio.write_pcm16_samples(tmp_path / "s.wav", pcm16[0], 16000)
    back = wavio.read_wav_pcm16(tmp_path / "s.wav")
    assert back[1] == 16000 and np.array_equal(back[0], pcm16[0])


@pytest.mark.parametrize("mode", ["exact_f32", "split_f16"])
def test_queued_pcm16_sessions_equal_css_run_pcm16(mc_state, mode):
    """css_run_enqueue_pcm16 (round 6): sessions as mono PCM16 planes through the QUEUE -- sharing estimator batches with each
    other and with float sessions of the same queue, page-locked or pageable -- give, each, the bits of its own synchronous
    css_run_pcm16 (samples and peaks); the full 18-block estimator keeps the host passes ahead of the device."""
    css, L = pkg("css"), pkg("_lib")
    st, desc = mc_state
    mk = lambda **kw: css.make_run_cfg(css.CssCfg(show_progressbar=False, **kw), 16000, 7)
    cfgs = [mk(activity_th=0.3), mk(activity_th=0.45), mk(activity_th=0.3, stitching_loss="mse"), mk(activity_th=0.3), mk(activity_th=0.3)]
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=128, linear_mode=mode)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([33.0, 20.0, 41.0, 12.3, 27.0]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=800 + k)[0, :int(seconds * 16000) - 29 * k]
            q = np.clip(np.rint(mix * (0.05 if k % 2 else 0.2) * 32768.0), -32768, 32767).astype(np.int16)      # [n, 7]
            block = L.pinned_empty((7, q.shape[0]), np.int16)
            block[:] = q.T
            planes = [block[c] for c in range(7)]
            ref16, refpk = h.run_pcm16(planes, cfgs[k])
            f32 = np.ascontiguousarray(q.astype(np.float32) / np.float32(32768.0))
            sessions.append((planes, cfgs[k], ref16.copy(), refpk.copy(), L.pinned_copy(f32), h.run(f32, cfgs[k]).copy(), block))
        for pinned_out in (True, False):
            for rounds in range(2):
                got = []
                for k, (planes, cfg, ref16, refpk, f32, reff, _) in enumerate(sessions):
                    o16 = (L.pinned_empty if pinned_out else np.empty)(ref16.shape, np.int16)
                    pk = (L.pinned_empty if pinned_out else np.empty)((3,), np.float32)
                    o16[:] = -1
                    a = h.run_enqueue_pcm16(planes, cfg, o16, pk)
                    b = None
                    if k % 2 == 0:            # a float session of the same queue in between
                        of = L.pinned_empty(reff.shape, np.float32)
                        of[:] = np.nan
                        b = h.run_enqueue(f32, cfg, of)
                    got.append((a, pk, b))
                h.wait()
                for k, ((a, pk, b), (_, _, ref16, refpk, _, reff, _)) in enumerate(zip(got, sessions)):
                    assert np.array_equal(a, ref16), (mode, pinned_out, rounds, k)
                    assert np.array_equal(pk, refpk), (mode, pinned_out, rounds, k)
                    assert b is None or np.array_equal(b, reff), (mode, pinned_out, rounds, k)
        with pytest.raises(L.CssError):       # a null plane is refused when the session is queued
            import ctypes as C
            ptrs = (C.c_void_p * 7)(*([sessions[0][0][0].ctypes.data] * 6 + [None]))
            o16 = L.pinned_empty(sessions[0][2].shape, np.int16)
            L.check(h.h, h.lib.css_run_enqueue_pcm16(h.h, ptrs, sessions[0][0][0].shape[0], 7, C.byref(cfgs[0].c), o16.ctypes.data_as(C.c_void_p),
                                                     o16.shape[1], None))
        h.wait()
    finally:
        sep.close()


def test_session_loop_through_the_queue_equals_css_inference(tmp_path, tiny_models):
    """pipeline.css_sessions (the queue: css_run_enqueue_pcm16 / css_wait, wav decode and file writes on worker threads) against
    css_inference called session by session as the reference's loop does (inference.py:59-63): the same files, byte for byte,
    the same rows -- MC and SC sessions interleaved, more sessions than one queue holds, one session that is NOT 16-bit PCM
    (float path inside the loop), the mode opt-in passed through."""
    import filecmp
    import pandas as pd
    css, wavio, pipe, sep_mod = pkg("css"), pkg("wavio"), pkg("pipeline"), pkg("separator")
    models_dir, models = tiny_models
    mix = (pkg("synth").synth_meeting(9.0, 7, seed=31) * 0.25).astype(np.float32)
    rows = []
    for i in range(7):
        is_mc = i not in (2, 5)
        part = mix[:, 4000 * i: 4000 * i + 70000 + 3000 * i, : (7 if is_mc else 1)]
        rows.append(_write_session(tmp_path, wavio, part, f"Q{i}_{'mc' if is_mc else 'sc'}", is_mc))
    # session 3's channel 0 as a 32-bit float wav: not the PCM16 edge's case (css.py::css_inference takes load_audio for it)
    import struct
    x = mix[0, 12000:12000 + 79000, 0].astype("<f4")
    payload = x.tobytes()
    with open(rows[3]["wav_file_names"][0], "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32) +
                b"data" + struct.pack("<I", len(payload)) + payload)
    df = pd.DataFrame(rows)
    cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)
    for mode in ("exact_f32", "split_f16"):
        got = pipe.css_sessions(str(tmp_path / f"queue_{mode}"), models_dir, df, cfg, queue_depth=3, io_threads=3, linear_mode=mode)
        assert list(got.session_id) == list(df.session_id) and list(got.columns) == list(df.columns) + ["sep_wav_file_names"]
        resident = {k: sep_mod.load_css_model(os.path.join(models_dir, "notsofar", "conformer1.0", "mc" if k else "sc"), linear_mode=mode)[0] for k in (True, False)}
        try:
            for (_, row), (_, grow) in zip(df.iterrows(), got.iterrows()):
                one = css.css_inference(str(tmp_path / f"single_{mode}"), models_dir, row, cfg, False, separator=resident[bool(row.is_mc)])
                assert [os.path.basename(a) for a in one["sep_wav_file_names"]] == [os.path.basename(a) for a in grow["sep_wav_file_names"]]
                for a, b in zip(one["sep_wav_file_names"], grow["sep_wav_file_names"]):
                    assert filecmp.cmp(a, b, shallow=False), (mode, row.session_id, a)
                assert filecmp.cmp(os.path.join(os.path.dirname(one["sep_wav_file_names"][0]), "input_mixture.wav"),
                                   os.path.join(os.path.dirname(grow["sep_wav_file_names"][0]), "input_mixture.wav"), shallow=False)
        finally:
            for sep in resident.values():
                sep.close()


def test_device_handoff_to_whisper_front_end(tiny_models):
    """SURVEY.md 8f N4 (css.py:313 "drop silent parts to save ASR compute"): after a device-resident pass, each
    stream's active regions -- the time map -- and Whisper's log-mel features of their concatenation, computed on the
    GPU, against the oracle's restatement of whisper/audio.py on the same samples (whisper is not under the reference
    tree; the restatement is held to transformers.WhisperFeatureExtractor in tests/test_oracle_whisper_pin.py)."""
    import torch
    css, sep_mod, L = pkg("css"), pkg("separator"), pkg("_lib")
    _, models = tiny_models
    st, desc = models["mc"]
    mix = pkg("synth").synth_meeting(20.0, 7, seed=12)
    sep = sep_mod.HipSeparator(st, None, device=0)
    try:
        h = sep.handle
        # a threshold inside the range of this model's activity values, so that the gate really toggles
        probe_cfg = css.make_run_cfg(css.CssCfg(activity_th=0.0, show_progressbar=False), 16000, 7)
        n = mix.shape[1]
        plan = L.plan(desc, probe_cfg, n)
        pcm = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda()
        wav = torch.empty((3, int(plan.n_out)), dtype=torch.float32, device="cuda")
        h.run_device(pcm.data_ptr(), n, 7, probe_cfg, wav.data_ptr(), int(plan.n_out))
        act = h.read(L.BUF_ACTIVITY)
        th = float(np.percentile(act, 70))
        run_cfg = css.make_run_cfg(css.CssCfg(activity_th=th, show_progressbar=False, activity_dilation_sec=0.05,
                                              activity_erosion_sec=0.02), 16000, 7)
        h.run_device(pcm.data_ptr(), n, 7, run_cfg, wav.data_ptr(), int(plan.n_out))
        torch.cuda.synchronize()
        act_f = h.read(L.BUF_ACT_FINAL).astype(bool)            # [S, T_long]
        host = wav.cpu().numpy()
        assert 0.05 < act_f.mean() < 0.95
        for k in range(3):
            for n_mels, pad in ((80, 8), (128, 0)):
                mel, regions = h.handoff_logmel(wav.data_ptr(), int(plan.n_out), k, n_mels=n_mels, pad_frames=pad)
                want = O.active_regions(act_f[k], pad, int(plan.n_out))
                assert np.array_equal(regions, want) and len(regions) >= 1
                assert (regions[1:, 0] > regions[:-1, 1]).all() and regions[-1, 1] <= plan.n_out
                cut = np.concatenate([host[k, a:b] for a, b in regions])
                ref = O.whisper_log_mel(cut, n_mels)
                assert mel.shape == ref.shape == (n_mels, len(cut) // 160)
                assert np.abs(mel - ref).max() < 2e-3 and np.sqrt(np.mean((mel - ref) ** 2)) < 2e-4, (k, n_mels)
            full, whole = h.handoff_logmel(wav.data_ptr(), int(plan.n_out), k, drop_silence=False)
            assert whole.tolist() == [[0, int(plan.n_out)]] and full.shape == (80, int(plan.n_out) // 160)
            assert np.abs(full - O.whisper_log_mel(host[k], 80)).max() < 2e-3
            assert full.shape[1] > mel.shape[1] * 0 + 1          # (the gated version is shorter whenever the gate is off somewhere)
        with pytest.raises(L.CssError):
            h.handoff_logmel(wav.data_ptr(), int(plan.n_out), 3)
    finally:
        sep.close()


def test_c_host_equals_the_python_shim(tmp_path):
    """examples/c_host.c -- a host in plain C on include/css_mi355.h (css_make_run_cfg, css_plan, css_host_alloc, css_create,
    css_run, css_run_enqueue, css_wait), compiled with gcc on the box -- against the Python shim on the same model and
    recording: the waveforms it writes are HipSeparator's, bit for bit, and its queued sessions equal its css_run."""
    import subprocess
    from test_cabi import _build_c_host, _write_c_host_inputs
    CSS, W, L = pkg("css"), pkg("weights"), pkg("_lib")
    desc = W.ModelDesc(num_blocks=2)
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 21))
    mix = pkg("synth").synth_meeting(21.0, 7, seed=5)[0]
    exe = _build_c_host(tmp_path)
    _write_c_host_inputs(tmp_path, desc, state, mix)
    out = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "pcm.f32"), "7", str(tmp_path / "wav.f32")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "every session equals css_run bit for bit: yes" in out.stdout, out.stdout
    sep = pkg("separator").HipSeparator(state, None, device=0, max_batch_segments=256)
    try:
        ref = sep.handle.run(np.ascontiguousarray(mix), CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7))
        got = np.fromfile(tmp_path / "wav.f32", dtype=np.float32).reshape(ref.shape)
        assert np.isfinite(got).all() and float(np.abs(got).max()) > 1e-3
        assert np.array_equal(got, ref)
    finally:
        sep.close()


def test_wait_sessions_releases_the_oldest_first(mc_state):
    """css_wait_sessions(h, n) (round 6): after it returns the FIRST n sessions queued since the last css_wait hold their final
    outputs -- float and PCM16 sessions, held-back ones flushed on demand -- while the later ones keep running; n beyond what
    was queued is an error, n = 0 a no-op, and css_wait starts the count again."""
    css, L = pkg("css"), pkg("_lib")
    st, desc = mc_state
    cfg = css.make_run_cfg(css.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=256)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([12.0, 31.0, 7.5, 44.0, 18.0, 25.0, 9.0, 36.0, 15.0]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=900 + k)[0]
            q = np.clip(np.rint(mix * 0.1 * 32768.0), -32768, 32767).astype(np.int16)
            if k % 3 == 1:
                blk = L.pinned_empty((7, q.shape[0]), np.int16)
                blk[:] = q.T
                planes = [blk[c] for c in range(7)]
                ref, _ = h.run_pcm16(planes, cfg)
                sessions.append(("pcm16", planes, ref.copy(), blk))
            else:
                f32 = L.pinned_copy(np.ascontiguousarray(q.astype(np.float32) / np.float32(32768.0)))
                sessions.append(("float", f32, h.run(f32, cfg).copy(), None))
        for rounds in range(2):
            outs = []
            for kind, src, ref, _ in sessions:
                o = L.pinned_empty(ref.shape, ref.dtype)
                o[...] = 77
                if kind == "pcm16":
                    h.run_enqueue_pcm16(src, cfg, o, L.pinned_empty((3,), np.float32))
                else:
                    h.run_enqueue(src, cfg, o)
                outs.append(o)
            h.wait_sessions(0)
            for n in (1, 2, 5, 9):
                h.wait_sessions(n)          # (the first call also flushes what was still held back for company)
                for k in range(n):
                    assert np.array_equal(outs[k], sessions[k][2]), (rounds, n, k)
            with pytest.raises(L.CssError, match="only 9 sessions"):
                h.wait_sessions(10)
            h.wait()
            with pytest.raises(L.CssError, match="only 0 sessions"):
                h.wait_sessions(1)
            for k in range(9):
                assert np.array_equal(outs[k], sessions[k][2])
    finally:
        sep.close()

"""Host-side checks of the C ABI that need no GPU: the library loads, exports every symbol the header
declares, and its pure-host entry points (planning, permutation scan, validation) agree with the oracle."""
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest

import css_oracle as O
from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def L():
    return pkg("_lib")


def header_functions():
    text = open(os.path.join(ROOT, "include", "css_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(css_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(L):
    lib = L.load()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libcss_mi355.so does not export {n}"
    assert sorted(L.SIGNATURES) == names, "the ctypes binding and the header disagree"
    assert lib.css_version().startswith(b"css_mi355")


def test_struct_layouts_match_header(L):
    assert C.sizeof(L.CssModelDesc) == 13 * 4
    assert C.sizeof(L.CssRunCfg) == 8 * 4 + 2 * 4 + 3 * 8
    assert C.sizeof(L.CssPlan) == 5 * 8 + 2 * 4
    assert C.sizeof(L.CssTimings) == 10 * 4 + 8 + 8 + 2 * 4


def test_header_constants_agree_with_the_binding(L):
    """The header's macros the Python side mirrors: the segment-length bound and the analysis windows; and the seconds ->
    frames conversion on either side of the tuned kernels' 512 frames (css.py:144-152)."""
    text = open(os.path.join(ROOT, "include", "css_mi355.h")).read()
    macro = lambda name: int(re.search(rf"#define\s+{name}\s+(\d+)", text).group(1))
    assert macro("CSS_MAX_SEGMENT_FRAMES") == L.MAX_SEGMENT_FRAMES
    assert {"hann": macro("CSS_WINDOW_HANN"), "sqrt_hann": macro("CSS_WINDOW_SQRT_HANN")} == L.ANALYSIS_WINDOWS
    CSS = pkg("css")
    frames = lambda seg, hop: int(CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=seg, hop_size_sec=hop), 16000, 7).c.segment_frames)
    assert (frames(3.0, 1.5), frames(8.0, 4.0), frames(9.0, 4.5), frames(10.0, 5.0), frames(60.0, 30.0)) == (186, 499, 561, 624, 3749)
    with pytest.raises(NotImplementedError):
        CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=300.0, hop_size_sec=150.0), 16000, 7)      # beyond the 262 s sanity bound
    with pytest.raises(NotImplementedError, match="1 <= hop < segment"):
        CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=3.0, hop_size_sec=3.0), 16000, 7)


def test_blob_size_agrees_with_packer(L):
    w = pkg("weights")
    for desc in (w.ModelDesc.mc_v1(), w.ModelDesc.sc_v1(), w.ModelDesc(num_blocks=2)):
        assert L.load().css_blob_num_floats(C.byref(L.make_desc(desc))) == w.blob_num_floats(desc)
    bad = w.ModelDesc(attention_heads=4)  # head size 128: rejected
    assert L.load().css_blob_num_floats(C.byref(L.make_desc(bad))) == -1


def test_pack_blob_layout():
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=1)
    st = w.portable_state_dict(desc, 7)
    blob, d2 = w.pack_blob(st)
    assert d2 == desc and blob.size == w.blob_num_floats(desc)
    off = 0
    secs = dict()
    for name, n in w.blob_sections(desc):
        secs[name] = (off, n)
        off += (n + 15) // 16 * 16
    o, n = secs["embed_w"]
    e = blob[o:o + n].reshape(512, desc.k_in_padded)
    assert np.array_equal(e[:, :1799], st[w.PREFIX + "conformer.embed.0.weight"]) and (e[:, 1799:] == 0).all()
    o, n = secs["b0.wqkv"]
    q = blob[o:o + n].reshape(1536, 512)
    assert np.array_equal(q[512:1024], st[w.PREFIX + "conformer.encoders.0.self_attn.linear_k.weight"])
    o, n = secs["b0.dw_wt"]
    assert np.array_equal(blob[o:o + n].reshape(33, 512).T, st[w.PREFIX + "conformer.encoders.0.conv.dw_conv_1d.weight"][:, 0])
    # a DDP checkpoint ("module." prefix, css/helpers.py:32-36) packs identically
    blob2, _ = w.pack_blob({"module." + k: v for k, v in st.items()})
    assert np.array_equal(blob, blob2)


def test_plan_matches_oracle(L):
    css = pkg("css")
    w = pkg("weights")
    desc = w.ModelDesc.mc_v1()
    for kw in ({}, dict(hop_size_sec=2.0), dict(segment_size_sec=4.0, hop_size_sec=2.0)):
        cfg = css.CssCfg(**kw)
        ocfg = O.OracleCssCfg(**kw)
        rc = css.make_run_cfg(cfg, 16000, 7)
        for n in (0, 100, 511, 512, 48000, 48320, 50000, 64356, 160000, 960000, 960001, 28800000):
            p = L.plan(desc, rc, n)
            op = O.make_plan(n, 16000, ocfg)
            assert (p.stft_frames, p.mix_frames, p.num_segments) == (op.stft_frames, op.mix_frames, op.num_segments), (kw, n)
            assert p.last_valid == op.seg_range(op.num_segments - 1)[2]
            assert p.n_out == (op.mix_frames - 1) * 256 + 512
            assert (rc.c.segment_frames, rc.c.hop_frames, rc.c.dilation_frames, rc.c.erosion_frames) == \
                (op.segment_frames, op.hop_frames, op.dilation_frames, op.erosion_frames)
            assert bool(p.zero_weight) == (op.num_segments == 1), (kw, n)   # css.py:297 fires for single-segment inputs


def test_segment_weight_bit_exact_vs_reference(golden):
    css = pkg("css")
    g = golden("segment_weight.npz")
    assert np.array_equal(css.calc_segment_weight(186, 9, 18, is_first_seg=True), g["first"])
    assert np.array_equal(css.calc_segment_weight(186, 9, 18), g["mid"])
    assert np.array_equal(css.calc_segment_weight(186, 9, 18, is_last_seg=True), g["last"])
    with pytest.raises(AssertionError, match="not enough frames"):
        css.calc_segment_weight(30, 9, 18)


def test_pit_scan_matches_bruteforce(L):
    rs = np.random.RandomState(3)
    costs = rs.rand(50, 3, 3)
    perms = L.pit_scan(costs, 3)
    assert tuple(perms[0]) == (0, 1, 2)
    for b in range(50):
        lp = perms[b]
        best = min(itertools.permutations(range(3)), key=lambda s: sum(costs[b, lp[a], s[a]] for a in range(3)))
        assert tuple(perms[b + 1]) == best
    # the reference's known-answer test (losses.py:109-123): exact permutation, zero loss
    t = rs.rand(100, 257, 4).astype(np.float32)
    p = (3, 0, 2, 1)
    _, _, c = O.pit_perm(t[..., p], t, "mse")
    assert tuple(L.pit_scan(c[None], 4)[1]) == p


def test_create_without_gpu_fails_loudly(L):
    if L.load().css_device_count() > 0:
        pytest.skip("a GPU is present")
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=1)
    blob, _ = w.pack_blob(w.portable_state_dict(desc, 0), desc)
    with pytest.raises(L.CssError) as e:
        L.Handle(desc, blob)
    assert e.value.code == L.CSS_ERR_NO_DEVICE
    # and the Python driver has no fallback either
    css, sep = pkg("css"), pkg("separator")
    s = sep.HipSeparator(w.portable_state_dict(desc, 0))
    with pytest.raises(L.CssError):
        css.separate_and_stitch(np.zeros((1, 64000, 7), np.float32), s, 16000, "cuda:0", css.CssCfg())
    # a foreign separator-protocol object gets the HIP stages around its masks (tests/test_hip_protocol.py) -- and no
    # CPU path either: without a GPU the stages' handle cannot be created
    with pytest.raises(L.CssError) as e:
        css.separate_and_stitch(np.zeros((1, 64000, 7), np.float32), object(), 16000, "cuda:0", css.CssCfg())
    assert e.value.code == L.CSS_ERR_NO_DEVICE


def test_wav_roundtrip_and_css_inference_plumbing(tmp_path):
    """load_audio / write_wav counterparts (css/helpers.py:40, utils/audio_utils.py:37)."""
    wavio = pkg("wavio")
    rs = np.random.RandomState(0)
    x = (rs.randn(7, 4000) * 0.1).astype(np.float32)
    names = []
    for c in range(7):
        p = tmp_path / f"ch{c}.wav"
        wavio.write_wav(p, x[c], 16000, max_norm=False)
        names.append(str(p))
    mix, sr = wavio.load_audio(names, is_mc=True)
    assert mix.shape == (1, 4000, 7) and sr == 16000 and mix.dtype == np.float32
    assert np.abs(mix[0].T - x).max() <= 1.0 / 32767 + 1e-7
    import scipy.io.wavfile as wf   # the pipeline's other reader (utils/audio_utils.py:23-31) parses our files
    sr2, y = wf.read(names[3])
    assert sr2 == 16000 and y.dtype == np.int16 and np.array_equal(y, np.rint(x[3].astype(np.float64) * 32767).astype(np.int16))
    wavio.write_wav(tmp_path / "n.wav", x[0] * 3.0)           # peak normalisation to 0.99
    y, _ = wavio.read_wav(tmp_path / "n.wav")
    assert abs(np.abs(y).max() - 0.99) < 1e-3
    m1, _ = wavio.load_audio(names[:1], is_mc=False)
    assert m1.shape == (1, 4000, 1)
    # pass_through_ch0 / cache short-circuits of css_inference (css.py:73-82) need no GPU
    import pandas as pd
    css = pkg("css")
    session = pd.Series({"wav_file_names": names, "session_id": "S1", "is_mc": True})
    out = css.css_inference(str(tmp_path), "unused", session, css.CssCfg(pass_through_ch0=True), False)
    assert out["sep_wav_file_names"] == names[:1] and "sep_wav_file_names" not in session
    d = tmp_path / "css_inference" / "S1"
    d.mkdir(parents=True)
    for i in range(3):
        wavio.write_wav(d / f"sep_stream{i}.wav", x[i])
    out = css.css_inference(str(tmp_path), "unused", session, css.CssCfg(), True)
    assert [os.path.basename(str(p)) for p in out["sep_wav_file_names"]] == [f"sep_stream{i}.wav" for i in range(3)]


def test_make_run_cfg_in_c_equals_the_python_shim(L, golden):
    """css_make_run_cfg (css/css.py:144-152 + calc_segment_weight :341-390 in C, for hosts without the Python shim): frames,
    knobs and the three windows are the Python shim's, bit for bit -- whose windows are the reference's (segment_weight.npz)."""
    CSS, W = pkg("css"), pkg("weights")
    lib = L.load()
    cases = [dict(), dict(activity_th=0.3), dict(segment_size_sec=4.0, hop_size_sec=2.0), dict(segment_size_sec=2.0, hop_size_sec=0.5, seg_weight_m1_sec=0.25),
             dict(segment_size_sec=8.0, hop_size_sec=4.0, seg_weight_m0_sec=0.2, seg_weight_m1_sec=0.9), dict(segment_size_sec=1.0, hop_size_sec=0.75),
             dict(stitching_loss="mse", stitching_input="separation_result", normalize_segment_power=True, mc_mvdr=False, mc_mask_floor_db=-6.0),
             dict(segment_size_sec=3.3, hop_size_sec=1.1, activity_dilation_sec=0.7, activity_erosion_sec=0.1, seg_weight_m0_sec=0.0, seg_weight_m1_sec=0.05)]
    for frame_len, frame_hop in ((512, 256), (400, 160), (512, 128)):
        for fs in (16000, 8000):
            for ch in (7, 1):
                for kw in cases:
                    cfg = CSS.CssCfg(show_progressbar=False, **kw)
                    desc = W.ModelDesc(num_mics=ch, in_features=257 * (1 + (ch - 1)), frame_len=frame_len, frame_hop=frame_hop)
                    floor_db = cfg.mc_mask_floor_db if ch > 1 else cfg.sc_mask_floor_db
                    sec = L.CssCfgSeconds(cfg.segment_size_sec, cfg.hop_size_sec, cfg.seg_weight_m0_sec, cfg.seg_weight_m1_sec, cfg.activity_dilation_sec,
                                          cfg.activity_erosion_sec, cfg.activity_th, floor_db, int(cfg.mc_mvdr), {"l1": 0, "mse": 1}[cfg.stitching_loss],
                                          {"mask": 0, "separation_result": 1}[cfg.stitching_input], int(cfg.normalize_segment_power))
                    out = L.CssRunCfg()
                    win = np.full(3 * 16384, np.nan, np.float32)
                    rc = lib.css_make_run_cfg(C.byref(L.make_desc(desc)), C.byref(sec), fs, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), win.size)
                    try:
                        ref = CSS.make_run_cfg(cfg, fs, ch, frame_len, frame_hop)
                    except AssertionError:
                        assert rc == L.CSS_ERR_WEIGHT_WINDOW, (kw, rc)
                        continue
                    assert rc == 0, (kw, fs, frame_len, rc)
                    for f, _ in L.CssRunCfg._fields_[:10]:
                        assert getattr(out, f) == getattr(ref.c, f), (f, kw, fs, frame_len)
                    T = out.segment_frames
                    for k, w in enumerate(ref._w):
                        assert np.array_equal(win[k * T:(k + 1) * T], w), (k, kw, fs, frame_len)
                    assert C.addressof(out.w_mid.contents) == win.ctypes.data + 4 * T
    # ... and the reference's own windows for the default segmentation
    g = golden("segment_weight.npz")
    sec = L.CssCfgSeconds(3.0, 1.5, 0.15, 0.3, 0.4, 0.2, 0.4, 0.0, 1, 0, 0, 0)
    out, win = L.CssRunCfg(), np.zeros(3 * 186, np.float32)
    assert lib.css_make_run_cfg(C.byref(L.make_desc(W.ModelDesc.mc_v1())), C.byref(sec), 16000, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), win.size) == 0
    keys = sorted(g.files)
    for k, name in enumerate(("first", "mid", "last")):
        cand = [x for x in keys if name in x]
        if cand:
            assert np.array_equal(win[k * 186:(k + 1) * 186], g[cand[0]].astype(np.float32).reshape(-1)[:186]), (name, cand)
    # errors: css.py:374, css.py:224, a window buffer that is too small
    bad = L.CssCfgSeconds(3.0, 1.5, 0.15, 1.6, 0.4, 0.2, 0.4, 0.0, 1, 0, 0, 0)
    assert lib.css_make_run_cfg(C.byref(L.make_desc(W.ModelDesc.mc_v1())), C.byref(bad), 16000, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), win.size) == L.CSS_ERR_WEIGHT_WINDOW
    bad = L.CssCfgSeconds(3.0, 1.5, 0.15, 0.3, 0.4, 0.2, 0.4, 3.0, 1, 0, 0, 0)
    assert lib.css_make_run_cfg(C.byref(L.make_desc(W.ModelDesc.mc_v1())), C.byref(bad), 16000, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), win.size) == L.CSS_ERR_MASK_FLOOR
    assert lib.css_make_run_cfg(C.byref(L.make_desc(W.ModelDesc.mc_v1())), C.byref(sec), 16000, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), 100) == L.CSS_ERR_INVALID_ARG
    for field, value in (("segment_size_sec", float("nan")), ("hop_size_sec", float("inf")), ("seg_weight_m0_sec", -1.0), ("activity_th", float("nan")),
                         ("mask_floor_db", float("nan")), ("stitching_loss", 7), ("segment_size_sec", 1e12)):
        bad = L.CssCfgSeconds(3.0, 1.5, 0.15, 0.3, 0.4, 0.2, 0.4, 0.0, 1, 0, 0, 0)
        setattr(bad, field, value)
        assert lib.css_make_run_cfg(C.byref(L.make_desc(W.ModelDesc.mc_v1())), C.byref(bad), 16000, C.byref(out), win.ctypes.data_as(C.POINTER(C.c_float)), win.size) == L.CSS_ERR_INVALID_ARG, field


def _build_c_host(tmp_path):
    """examples/c_host.c as strict C99 against include/css_mi355.h and the shipped library"""
    import subprocess
    exe = str(tmp_path / "c_host")
    pkg_dir = os.path.join(ROOT, "notsofar1-challenge_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-D_POSIX_C_SOURCE=199309L", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host.c"), "-o", exe, "-L" + pkg_dir, "-lcss_mi355", "-Wl,-rpath," + pkg_dir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib", "-lm"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def _write_c_host_inputs(tmp_path, desc, state, pcm):
    L, W = pkg("_lib"), pkg("weights")
    blob = np.ascontiguousarray(W.pack_blob(state, desc)[0], dtype=np.float32)
    with open(tmp_path / "model.bin", "wb") as f:
        f.write(bytes(L.make_desc(desc)))
        f.write(np.int64(blob.size).tobytes())
        f.write(blob.tobytes())
    np.ascontiguousarray(pcm, dtype=np.float32).tofile(tmp_path / "pcm.f32")


def test_c_host_compiles_as_c99_and_plans(tmp_path, L):
    """The header is valid strict C99 and a C program links against exactly what it declares; without a GPU the example stops
    at css_device_count() after css_make_run_cfg / css_plan, whose numbers are the Python shim's."""
    import subprocess
    CSS, W = pkg("css"), pkg("weights")
    exe = _build_c_host(tmp_path)
    desc = W.ModelDesc(num_blocks=1)
    n = 5 * 16000 + 77
    _write_c_host_inputs(tmp_path, desc, W.portable_state_dict(desc, 3), np.zeros((n, 7), np.float32))
    out = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "pcm.f32"), "7", str(tmp_path / "wav.f32")], capture_output=True, text=True)
    plan = L.plan(desc, CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3), 16000, 7), n)
    assert f"{n} samples x 7 channels = {plan.num_segments} segments of 186 frames, {plan.n_out} output samples per stream" in out.stdout, out.stdout + out.stderr
    if L.load().css_device_count() < 1:
        assert out.returncode == 3 and "no HIP device" in out.stderr


def test_every_handle_entry_point_rejects_a_null_handle(L):
    """Every entry point whose first parameter is the handle returns a negative status for NULL (no GPU needed, nothing
    dereferenced) -- called with zeros / NULLs for the rest, as a host with a failed css_create would."""
    lib = L.load()
    text = open(os.path.join(ROOT, "include", "css_mi355.h")).read()
    names = re.findall(r"^int\s+(css_\w+)\s*\(\s*css_handle_t\s+h", text, flags=re.M)
    assert len(names) >= 50
    for n in names:
        restype, argtypes = L.SIGNATURES[n]
        args = []
        for a in argtypes[1:]:
            args.append(0 if a in (C.c_int, C.c_int32, C.c_int64, C.c_size_t) else (0.0 if a in (C.c_float, C.c_double) else None))
        rc = getattr(lib, n)(None, *args)
        assert rc < 0 or (n == "css_destroy" and rc == 0), (n, rc)       # (destroying nothing is not an error, as free(NULL))


def test_plan_matches_oracle_on_random_inputs(L):
    """css_plan against the oracle's index arithmetic (css/css.py:155-171) for random lengths and segmentations (hypothesis):
    frames, segments, the last segment's valid frames, the output length and the zero-weight verdict."""
    from hypothesis import given, settings, strategies as st
    css, w = pkg("css"), pkg("weights")
    desc = w.ModelDesc.mc_v1()

    @settings(max_examples=300, deadline=None)
    @given(n=st.integers(min_value=0, max_value=40_000_000), seg=st.sampled_from([1.0, 2.0, 3.0, 3.3, 4.0, 8.0]), frac=st.sampled_from([0.25, 0.5, 0.75]))
    def check(n, seg, frac):
        kw = dict(segment_size_sec=seg, hop_size_sec=seg * frac, seg_weight_m0_sec=0.05 * seg, seg_weight_m1_sec=0.1 * seg)
        rc = css.make_run_cfg(css.CssCfg(**kw), 16000, 7)
        p, op = L.plan(desc, rc, n), O.make_plan(n, 16000, O.OracleCssCfg(**kw))
        assert (p.stft_frames, p.mix_frames, p.num_segments, p.n_out) == (op.stft_frames, op.mix_frames, op.num_segments, (op.mix_frames - 1) * 256 + 512), (n, kw)
        assert p.last_valid == op.seg_range(op.num_segments - 1)[2], (n, kw)
        assert (rc.c.segment_frames, rc.c.hop_frames) == (op.segment_frames, op.hop_frames)

    check()

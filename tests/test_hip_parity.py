"""Parity of the HIP path (through the C ABI) against the oracle and against the golden fixtures produced by the
real reference.  Everything here needs an MI355X: run with `-m gpu`.

Tolerances (float32 path; written where they are asserted):
  * masks vs oracle / vs reference       max-abs <= 1.5e-5 on the golden inputs (observed 3e-6 .. 9e-6: 2x margin).
                                         SURVEY.md 8(d) asks for 5e-6; that is the oracle's own distance class (1.6e-6 to
                                         the reference) but not reachable in the max norm by a float32 feature path: the
                                         IPD of a short mean-removed phasor amplifies the float32 rounding of the STFT
                                         (feature error p99 1e-5, max 7e-4, DESIGN.md hazard 3) and the masks inherit it.
                                         Arbitrary clips / model widths (tests at the end) keep 5e-5 for the same reason.
  * separated waveforms vs reference     rel-RMS <= 1e-4 on identical winner-take-all decisions
                                         (BASELINE.json north_star: "within 1e-4 RMS on the separated waveforms")
  * decisions (segment indices, permutations, activity bits): exact
"""
import importlib

import numpy as np
import pytest

import css_oracle as O
from conftest import pkg, rel_rms, take_windows

pytestmark = pytest.mark.gpu

F, T, S = 257, 186, 3


@pytest.fixture(scope="module")
def L():
    lib = pkg("_lib")
    if lib.load().css_device_count() < 1:
        pytest.fail("no HIP device visible: the parity tests must run on the GPU box")
    return lib


@pytest.fixture(scope="module")
def CSS():
    return pkg("css")


@pytest.fixture(scope="module")
def sep_mc(L, mc_state):
    st, desc = mc_state
    s = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=64)
    yield s
    s.close()


@pytest.fixture(scope="module")
def sep_sc(L, sc_state):
    st, desc = sc_state
    s = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=64)
    yield s
    s.close()


def cfgs(CSS, **kw):
    kw.setdefault("activity_th", 0.3)
    return CSS.CssCfg(show_progressbar=False, **kw), O.OracleCssCfg(**kw)


def hip_masks_per_segment(h, L, nseg):
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
    return [(np.ascontiguousarray(np.moveaxis(m[:S, :, i], 0, 2)), np.ascontiguousarray(np.moveaxis(m[S:, :, i], 0, 2)))
            for i in range(nseg)], m


def virtual_rank_run(PAR, L, h, pcm, run_cfg, world, poison=False):
    """The sharded driver's phases for `world` virtual ranks on ONE GPU: every virtual rank replays its own session from
    scratch up to the current phase (nothing a previous virtual rank computed may be needed), the exchanges are played by
    stacking the pieces.  Returns the joined waveforms [S, n_out] (numpy)."""
    import torch
    dev = torch.device("cuda", 0)
    be = PAR.HipShardBackend(h, dev)
    n, c = pcm.shape
    T, hop = int(run_cfg.c.segment_frames), int(run_cfg.c.hop_frames)
    if int(h.desc.frame_len) != 2 * int(h.desc.frame_hop):
        return _virtual_rank_run_general(PAR, L, h, be, pcm, run_cfg, world, torch)
    pieces = {0: [], 1: [], 2: []}
    ss = None
    for phase in range(3):
        for r in range(world):
            h.begin(pcm, n, c, run_cfg)
            if poison and world > 1:   # nothing a previous virtual rank left behind may be readable
                nseg = h.get_plan().num_segments
                h.write(L.BUF_MASKS, np.full(((S + 1) * F, nseg * T), np.nan, np.float32))
                h.write(L.BUF_SEP, np.full((nseg, S, F, T * 2), np.nan, np.float32))
                h.write(L.BUF_X, np.full((c, 2 * F, h.buffer_dims(L.BUF_X)[0][2]), np.nan, np.float32))
            ss = PAR.ShardedSession(be, S, T, hop, 256, r, world)
            piece = ss.segments_and_costs()
            if phase >= 1:
                piece = ss.masks_and_activity(torch.stack(pieces[0]) if world > 1 else None)
            if phase >= 2:
                piece = ss.gate_and_istft(torch.stack(pieces[1]) if world > 1 else None)
            with be.on_stream():
                pieces[phase].append(piece.clone() if piece is not None else None)
    with be.on_stream():
        out = ss.join_shards(torch.stack(pieces[2])) if world > 1 else pieces[2][0][:, :ss.n_out]
        out = out.cpu()
        if world > 1:   # gather="range": a rank finishes its own samples with its left neighbour's seam block alone
            edge = 0
            for r in range(world):
                sr = PAR.ShardedSession(be, S, T, hop, 256, r, world)
                seams = torch.stack([PAR.ShardedSession(be, S, T, hop, 256, q, world).seam_piece(pieces[2][q]).clone()
                                     for q in range(world)])
                lo, hi = sr.own_range()
                own = sr.finish_range(pieces[2][r].clone(), seams).cpu()
                assert lo == edge and tuple(own.shape) == (S, hi - lo) and torch.equal(own, out[:, lo:hi]), (r, lo, hi)
                edge = hi
            assert edge == out.shape[1]
    pieces.clear()
    be.close()
    return out.numpy()


def _virtual_rank_run_general(PAR, L, h, be, pcm, run_cfg, world, torch):
    """virtual_rank_run for frame geometries other than frame_len = 2 hop: exchange 3 carries the synthesis rows of a rank's
    last ovl - 1 frames (parallel.py), every rank finishes its own range with the single-GPU overlap-add; both gathers."""
    n, c = pcm.shape
    T, hop = int(run_cfg.c.segment_frames), int(run_cfg.c.hop_frames)
    fl, fh, S_ = int(h.desc.frame_len), int(h.desc.frame_hop), int(h.desc.num_spks)
    mk = lambda r: PAR.ShardedSession(be, S_, T, hop, fh, r, world, frame_len=fl)
    pieces = {0: [], 1: [], 2: [], 3: []}
    ss = None
    for phase in range(4):
        for r in range(world):
            h.begin(pcm, n, c, run_cfg)
            ss = mk(r)
            assert ss.general and ss.K == -(-fl // fh) - 1
            piece = ss.segments_and_costs()
            if phase >= 1:
                piece = ss.masks_and_activity(torch.stack(pieces[0]) if world > 1 else None)
            if phase >= 2:
                piece = ss.gate_and_istft(torch.stack(pieces[1]) if world > 1 else None)
            if phase >= 3:
                piece = ss.finish_range(None, torch.stack(pieces[2]))
            with be.on_stream():
                pieces[phase].append(piece.clone() if piece is not None else None)
    with be.on_stream():
        # gather="range": the ranks' finished ranges tile the output
        edge, parts = 0, []
        for r in range(world):
            lo, hi = mk(r).own_range()
            assert lo == edge and tuple(pieces[3][r].shape) == (S_, hi - lo), (r, lo, hi, pieces[3][r].shape)
            parts.append(pieces[3][r])
            edge = hi
        out = torch.cat(parts, dim=1)
        assert edge == ss.n_out == out.shape[1]
        # gather="all": the padded finished ranges, all-gathered and placed
        padded = torch.zeros((world, S_, ss.max_len), dtype=torch.float32, device=out.device)
        for r in range(world):
            padded[r, :, :parts[r].shape[1]] = parts[r]
        assert torch.equal(ss.join_shards(padded), out)
        out = out.cpu()
    pieces.clear()
    be.close()
    return out.numpy()


def _unpack(bits, shape):
    return np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape).astype(bool)


# ------------------------------------------------------------------------------------------------ stages
def test_stage_by_stage_vs_oracle(L, CSS, sep_mc, mc_state, mix_stage, golden):
    st, _ = mc_state
    params = O.ConformerParams(st)
    cfg, ocfg = cfgs(CSS)
    h = sep_mc.handle
    wav = h.run(mix_stage[0], CSS.make_run_cfg(cfg, 16000, 7))
    plan = h.get_plan()
    nseg = plan.num_segments
    assert (plan.stft_frames, plan.mix_frames, nseg, plan.n_out, plan.last_valid) == (250, 250, 2, 64256, 157)

    # STFT (feature.py:88): planes vs oracle; the DC / Nyquist imaginary parts must be EXACT zeros
    X = h.read(L.BUF_X)
    xo = O.stft(mix_stage[0])
    xc = np.moveaxis((X[:, :F] + 1j * X[:, F:])[:, :, :plan.stft_frames], 0, 2)
    assert rel_rms(xc, xo) < 1e-6
    assert (X[:, F, :plan.stft_frames] == 0).all() and (X[:, 2 * F - 1, :plan.stft_frames] == 0).all()
    g = golden("stage_mc.npz")
    assert rel_rms(xc[::16, ::4], g["stft"]) < 2e-6          # ... and vs the reference's own STFT

    # features of segment 0 (feature.py:478-569 + conformer.py:298-299)
    feat = h.read(L.BUF_FEATURES)
    seg0 = xo[:, :T]
    fo = (O.features(seg0).T + params("input_bias").reshape(-1)) * params("input_scale").reshape(-1)
    d = np.abs(feat[:T, :1799] - fo)
    assert (feat[:, 1799:] == 0).all()
    assert d.max() < 1e-2 and np.percentile(d, 99) < 1e-4 and (d > 1).sum() == 0   # no atan2 branch flips

    # encoder output and masks (conformer.py:287-310)
    taps = {}
    om = O.conformer_forward(params, O.features(seg0), taps=taps)
    hid = h.read(L.BUF_HIDDEN)
    assert np.abs(hid[:T] - taps["block17"]).max() < 1e-4
    per_seg, m = hip_masks_per_segment(h, L, nseg)
    assert np.abs(m[:, :, 0, :] - om).max() < 1.5e-5
    fd, td = int(g["fdec"]), int(g["tdec"])
    assert np.abs(per_seg[0][0][::fd, ::td] - g["masks_spk"][0]).max() < 1.5e-5    # vs the reference's masks
    assert np.abs(per_seg[1][0][::fd, ::td] - g["masks_spk"][1]).max() < 1.5e-5

    # everything downstream of the masks, against the oracle driven by the HIP masks with a float64 MVDR
    taps = {}
    ow, oside = O.separate_and_stitch(mix_stage, params, 16000, ocfg, separate_fn=lambda i, seg: per_seg[i],
                                      mvdr_cplx=np.complex128, taps=taps)
    scm = h.read(L.BUF_SCM)  # [seg, 4, F, 49] packed Hermitian
    oscm = taps["mvdr0"]["scm"]  # [4, F, 7, 7]
    # (the oracle accumulates its own STFT, which differs from the HIP STFT by float32 rounding: ~2e-7)
    assert np.abs(scm[0, :, :, :7] - np.real(np.diagonal(oscm, axis1=2, axis2=3))).max() / np.abs(oscm).max() < 5e-6
    assert rel_rms(scm[0, :, :, 7] + 1j * scm[0, :, :, 8], oscm[:, :, 0, 1]) < 5e-6
    bfw = h.read(L.BUF_BFW)  # [seg, S, F, 14]
    w = bfw[0, :, :, 0::2] + 1j * bfw[0, :, :, 1::2]
    assert rel_rms(w, taps["mvdr0"]["w"]) < 1e-4
    print(f"W vs the reference's complex64 solve: {rel_rms(w, g['w_seg0']):.2e}")
    assert rel_rms(w, g["w_seg0"]) < 1.2e-4        # the reference's complex64 solve, on its own masks (measured 5.9e-5)
    sep_ = h.read(L.BUF_SEP).reshape(nseg, S, F, T, 2)
    costs = h.read(L.BUF_PIT_COST)
    assert np.abs(costs[0].reshape(S, S) - oside["pit_costs"][0]).max() < 1e-9
    assert [tuple(p) for p in h.read(L.BUF_PERMS)] == [tuple(p) for p in oside["perms"]]
    assert np.abs(np.transpose(h.read(L.BUF_MASK_ST), (1, 2, 0)) - oside["mask_stitched"][0]).max() < 1e-6
    assert np.abs(h.read(L.BUF_ACTIVITY).T - oside["activity"]).max() < 2e-6
    assert np.array_equal(h.read(L.BUF_ACT_B).astype(bool).T, oside["activity_b"])
    assert np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, oside["activity_final"][0])
    assert np.isfinite(sep_).all()
    for k in range(S):
        assert rel_rms(wav[k], ow[k]) < 1e-5
    # ... and against the reference's waveforms (free-running decisions: no flips on this input)
    ww = take_windows(wav, 4)
    for k in range(S):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4


def test_separator_protocol(L, sep_mc, mix_stage):
    """stft / separate / istft of the plug-in interface (conformer_wrapper.py:79-146)."""
    import torch
    xo = O.stft(mix_stage[0])
    xs = sep_mc.stft(torch.from_numpy(mix_stage))
    assert tuple(xs.shape) == (1, F, 250, 7) and xs.dtype == torch.complex64
    assert rel_rms(xs.numpy()[0], xo) < 1e-6
    x1 = sep_mc.stft(torch.from_numpy(mix_stage[:, :, 0]))
    assert tuple(x1.shape) == (1, F, 250) and rel_rms(x1.numpy()[0], xo[:, :, 0]) < 1e-6
    two = torch.stack([xs[0, :, :T], xs[0, :, 64:64 + T]])
    ms = sep_mc.separate(two)
    assert tuple(ms["spk_masks"].shape) == (2, F, T, 3) and tuple(ms["noise_masks"].shape) == (2, F, T, 1)
    one = sep_mc.separate(two[1:])
    assert np.array_equal(one["spk_masks"].numpy()[0], ms["spk_masks"].numpy()[1])   # batch-invariant bits
    fwd = sep_mc.forward(torch.from_numpy(mix_stage[:, :48000]))
    assert np.abs(fwd["spk_masks"].numpy()[0] - ms["spk_masks"].numpy()[0]).max() < 1e-6
    rs = np.random.RandomState(0)
    y = (rs.randn(2, F, 50) + 1j * rs.randn(2, F, 50)).astype(np.complex64)
    assert rel_rms(sep_mc.istft(torch.from_numpy(y)).numpy(), O.istft(y)) < 2e-6


# ------------------------------------------------------------------------------------------------ end to end
@pytest.fixture(scope="module")
def e2e(L, CSS, sep_mc, mix60, golden):
    g = golden("e2e_mc.npz")
    mix = mix60[:, :20 * 16000]
    cfg, _ = cfgs(CSS)
    wavs, side = CSS.separate_and_stitch(mix, sep_mc, 16000, "cuda:0", cfg)
    h = sep_mc.handle
    per_seg, m = hip_masks_per_segment(h, L, int(g["num_segments"]))
    return g, mix, wavs, side, per_seg, m


def test_e2e_decisions_match_reference(e2e, L, sep_mc):
    g, mix, wavs, side, per_seg, m = e2e
    h = sep_mc.handle
    assert h.get_plan().num_segments == int(g["num_segments"]) and len(wavs[0]) == int(g["wav_len"])
    assert [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["pit_perm"]]
    shape = tuple(g["activity_shape"])
    assert np.array_equal(side["activity_b"].numpy(), _unpack(g["activity_b"], shape))
    assert np.array_equal(side["activity_final"].numpy()[0], _unpack(g["activity_final"], shape))
    assert tuple(side["mask_stitched"].shape) == (1, F, shape[0], 3) and side["segment_frames"] == T
    assert np.abs(side["mask_stitched"].numpy()[0, ::16, ::8] - g["mask_stitched"]).max() < 1.5e-5
    flips = int((np.argmax(m, axis=0).transpose(1, 0, 2) != g["wta_index"]).sum())
    assert flips <= 1e-5 * g["wta_index"].size + 3      # float32-rounding-level ties only


def test_e2e_waveform_vs_reference(e2e, L, CSS, sep_mc):
    """<= 1e-4 rel-RMS against the reference's separated waveforms on the reference's WTA decisions."""
    g, mix, wavs, side, per_seg, m = e2e
    h = sep_mc.handle
    cfg, _ = cfgs(CSS)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    nseg, TL = int(g["num_segments"]), h.get_plan().mix_frames
    h.begin(mix[0], mix.shape[1], 7, run_cfg)
    h.write(L.BUF_WTA_OVERRIDE, g["wta_index"])
    h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
    h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
    w2 = h.read(L.BUF_WAV)
    ww = take_windows(w2)
    for k in range(S):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4
        assert rel_rms(w2[k, ::64], g["wav_dec"][k]) < 1e-4
    # free-running decisions: 1e-4 as well when no winner-take-all decision differs from the reference's (one flipped
    # time-frequency point moves a stream by ~1e-4, a handful stay below 1e-3)
    flips = int((np.argmax(m, axis=0).transpose(1, 0, 2) != g["wta_index"]).sum())
    ww = take_windows(np.stack(wavs))
    for k in range(S):
        assert rel_rms(ww[k], g["wav_windows"][k]) < (1e-4 if flips == 0 else 1e-3), flips


def test_e2e_forced_permutations(e2e, L, CSS, sep_mc):
    """Variant (a): segment i's speaker masks rotated by i mod 3 -> non-trivial stitching permutations."""
    g, mix, wavs, side, per_seg, m = e2e
    h = sep_mc.handle
    cfg, _ = cfgs(CSS)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    nseg, TL = int(g["num_segments"]), h.get_plan().mix_frames
    rot = m.copy()
    idx = g["wta_index"].copy()
    for i in range(nseg):
        rot[:S, :, i] = np.roll(m[:S, :, i], i % 3, axis=0)
        sel = idx[i] < 3
        idx[i][sel] = (idx[i][sel] + (i % 3)) % 3
    h.begin(mix[0], mix.shape[1], 7, run_cfg)
    h.stage_stft()
    h.write(L.BUF_MASKS, rot.reshape((S + 1) * F, nseg * T))
    h.write(L.BUF_WTA_OVERRIDE, idx)
    h.stage_mvdr(0, nseg); h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
    assert [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["rot_pit_perm"]]
    ww = take_windows(h.read(L.BUF_WAV))
    for k in range(S):
        assert rel_rms(ww[k], g["rot_wav_windows"][k]) < 1e-4


def test_e2e_activity_gating(e2e, L, CSS, sep_mc):
    """Variant (b): a threshold inside the range of activity values so that the gate toggles."""
    g, mix, wavs, side, per_seg, m = e2e
    h = sep_mc.handle
    cfg, _ = cfgs(CSS, activity_th=float(g["gate_th"]))
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    nseg, TL = int(g["num_segments"]), h.get_plan().mix_frames
    h.begin(mix[0], mix.shape[1], 7, run_cfg)
    h.write(L.BUF_WTA_OVERRIDE, g["wta_index"])
    h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
    h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
    shape = tuple(g["activity_shape"])
    assert np.array_equal(h.read(L.BUF_ACT_B).astype(bool).T, _unpack(g["gate_activity_b"], shape))
    af = h.read(L.BUF_ACT_FINAL).astype(bool).T
    assert np.array_equal(af, _unpack(g["gate_activity_final"], shape)) and 0 < af.mean() < 1
    ww = take_windows(h.read(L.BUF_WAV))
    for k in range(S):
        assert rel_rms(ww[k], g["gate_wav_windows"][k]) < 1e-4


def test_e2e_single_channel(L, CSS, sep_sc, mix60, golden):
    """BASELINE.json configs[2]: 1-ch mask net, no beamformer (mask multiplication, floor -inf)."""
    g = golden("e2e_sc.npz")
    mix = mix60[:, :12 * 16000, :1].copy()
    cfg, _ = cfgs(CSS)
    wavs, side = CSS.separate_and_stitch(mix, sep_sc, 16000, "cuda:0", cfg)
    h = sep_sc.handle
    assert [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["pit_perm"]]
    assert np.array_equal(side["activity_final"].numpy()[0], _unpack(g["activity_final"], tuple(g["activity_shape"])))
    ww = take_windows(np.stack(wavs))
    for k in range(S):
        assert rel_rms(ww[k], g["wav_windows"][k]) < 1e-4


# ------------------------------------------------------------------------------------------------ options / edges
def test_non_default_options_vs_oracle(L, CSS, sep_mc, mc_state, mix60):
    st, _ = mc_state
    params = O.ConformerParams(st)
    mix = mix60[:, 16000:16000 + 6 * 16000 + 777]   # 4 segments, ragged tail
    h = sep_mc.handle
    for kw in (dict(stitching_loss='mse'), dict(stitching_input='separation_result'),
               dict(normalize_segment_power=True), dict(mc_mvdr=False, mc_mask_floor_db=-6.0),
               dict(mc_mask_floor_db=-12.0)):
        cfg, ocfg = cfgs(CSS, **kw)
        wavs, side = CSS.separate_and_stitch(mix, sep_mc, 16000, "cuda:0", cfg)
        nseg = h.get_plan().num_segments
        per_seg, m = hip_masks_per_segment(h, L, nseg)
        ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg, separate_fn=lambda i, seg: per_seg[i],
                                          mvdr_cplx=np.complex128)
        assert [tuple(p) for p in h.read(L.BUF_PERMS)] == [tuple(p) for p in oside["perms"]], kw
        for k in range(S):
            assert rel_rms(wavs[k], ow[k]) < 2e-5, (kw, k)


@pytest.mark.parametrize("seg_hop", [(3.0, 2.0), (4.0, 2.0), (2.0, 1.0)])
def test_other_segmentations_vs_oracle(L, CSS, sep_mc, mc_state, mix60, seg_hop):
    """Segment / hop sizes other than the shipped 3 s / 1.5 s: different frame counts per segment (186, 249,
    124 -> different attention tile counts, conv runs, weight windows) and different overlaps."""
    st, _ = mc_state
    params = O.ConformerParams(st)
    kw = dict(segment_size_sec=seg_hop[0], hop_size_sec=seg_hop[1])
    cfg, ocfg = cfgs(CSS, **kw)
    mix = mix60[:, 32000:32000 + 11 * 16000 + 300]
    wavs, side = CSS.separate_and_stitch(mix, sep_mc, 16000, "cuda:0", cfg)
    h = sep_mc.handle
    plan = h.get_plan()
    oplan = O.make_plan(mix.shape[1], 16000, ocfg)
    assert (plan.mix_frames, plan.num_segments, plan.last_valid) == \
        (oplan.mix_frames, oplan.num_segments, oplan.seg_range(oplan.num_segments - 1)[2])
    Ts, nseg = oplan.segment_frames, oplan.num_segments
    assert side["segment_frames"] == Ts
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, Ts)
    xo = O.stft(mix[0])
    om = O.conformer_forward(params, O.features(xo[:, :Ts]))
    assert np.abs(m[:, :, 0, :] - om).max() < 5e-5
    per_seg = [(np.ascontiguousarray(np.moveaxis(m[:S, :, i], 0, 2)), np.ascontiguousarray(np.moveaxis(m[S:, :, i], 0, 2)))
               for i in range(nseg)]
    ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg, separate_fn=lambda i, seg: per_seg[i],
                                      mvdr_cplx=np.complex128)
    assert [tuple(p) for p in h.read(L.BUF_PERMS)] == [tuple(p) for p in oside["perms"]]
    assert np.array_equal(side["activity_final"].numpy()[0], oside["activity_final"][0])
    for k in range(S):
        assert len(wavs[k]) == len(ow[k]) and rel_rms(wavs[k], ow[k]) < 2e-5


def test_short_and_error_inputs(L, CSS, sep_mc):
    cfg, _ = cfgs(CSS)
    with pytest.raises(AssertionError, match="zero weights"):           # css.py:297 on inputs <= 3.0 s
        CSS.separate_and_stitch(np.zeros((1, 48000, 7), np.float32), sep_mc, 16000, "cuda:0", cfg)
    with pytest.raises(AssertionError, match="expecting 3 dimensions"):  # css.py:139
        CSS.separate_and_stitch(np.zeros((48000, 7), np.float32), sep_mc, 16000, "cuda:0", cfg)
    with pytest.raises(L.CssError):                                      # 1-ch audio into the 7-ch model
        CSS.separate_and_stitch(np.zeros((1, 64000, 1), np.float32), sep_mc, 16000, "cuda:0", cfg)
    with pytest.raises(AssertionError):                                  # css.py:224 mask_floor_db <= 0
        CSS.separate_and_stitch(np.zeros((1, 64000, 7), np.float32), sep_mc, 16000, "cuda:0",
                                CSS.CssCfg(mc_mask_floor_db=3.0))
    # what is left outside: no overlap for the stitching cost (hop == segment) -- the reference asserts there itself
    with pytest.raises(NotImplementedError, match="1 <= hop < segment"):
        CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=3.0, hop_size_sec=3.0), 16000, 7)
    # segments beyond 512 frames (8 s) run on the any-length kernels since round 4 (tests/test_hip_long_segments.py)
    assert int(CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=9.0, hop_size_sec=4.5), 16000, 7).c.segment_frames) == 561
    with pytest.raises(NotImplementedError):
        CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=300.0, hop_size_sec=150.0), 16000, 7)   # the sanity bound (262 s)
    # ... and what round 3 brought inside (fixtures from the reference: test_hip_golden_r2.py): hop < segment / 4, 311 frames
    assert int(CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=3.0, hop_size_sec=0.5), 16000, 7).c.hop_frames) == 31
    assert int(CSS.make_run_cfg(CSS.CssCfg(segment_size_sec=5.0, hop_size_sec=2.5), 16000, 7).c.segment_frames) == 311
    # digital silence: zero-padded frames, eps clamps, 1e-15 diagonal loading -- finite output, all zeros
    w, side = CSS.separate_and_stitch(np.zeros((1, 50000, 7), np.float32), sep_mc, 16000, "cuda:0", cfg)
    assert all(np.isfinite(x).all() and np.abs(x).max() == 0 for x in w)


def test_unconditioned_weights_behaviour(L, CSS, mix60):
    """Plain seeded initial weights, NO conditioning recipe (all other parity runs use head x 4 and a calibrated head
    bias): a random-init estimator returns nearly time-constant masks, the winner-take-all step then
    leaves the interference covariance of many bins with fewer frames than microphones, and the reference's own complex64
    solve is noise there (App. C.1: it disagrees with itself by 6 ... 67 %).  What must hold regardless: no operand
    leaves the split-f16 range, every decision equals the oracle's on the same masks, the beamformer weights agree with
    the oracle's float64 solve wherever the system is well-conditioned -- and the rest is REPORTED (parity_coverage.json /
    profiles/r03_parity_margins.txt), not asserted."""
    from test_hip_long import _report
    W = pkg("weights")
    st = W.portable_state_dict(W.ModelDesc.mc_v1(), 0)                  # Linear / LayerNorm init as torch's, nothing else
    sep = pkg("separator").HipSeparator(st, None, device=0)
    try:
        mix = mix60[:, :12 * 16000]
        cfg, ocfg = cfgs(CSS)
        wavs, side = CSS.separate_and_stitch(mix, sep, 16000, "cuda:0", cfg)
        h = sep.handle
        assert h.range_status() == (0, False) and all(np.isfinite(w).all() for w in wavs)
        nseg = int(h.get_plan().num_segments)
        per_seg, m = hip_masks_per_segment(h, L, nseg)
        taps = {}
        ow, oside = O.separate_and_stitch(mix, None, 16000, ocfg, separate_fn=lambda i, seg: per_seg[i],
                                          mvdr_cplx=np.complex128, taps=taps)
        assert [tuple(p) for p in h.read(L.BUF_PERMS)] == [tuple(p) for p in oside["perms"]]
        assert np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, oside["activity_final"][0])
        bfw = h.read(L.BUF_BFW)                                        # [seg, S, F, 14]
        tv = [min(T, int(h.get_plan().mix_frames) - i * 93) for i in range(nseg)]
        well, ill, mask_span = [], [], []
        for i in range(nseg):
            win = np.argmax(m[:, :, i, :tv[i]], axis=0)                # [F, t]
            cnt = np.stack([(win == k).sum(axis=1) for k in range(S + 1)])   # frames each mask wins, per bin
            w_hip = bfw[i, :, :, 0::2] + 1j * bfw[i, :, :, 1::2]      # [S, F, 7]
            w_or = taps[f"mvdr{i}"]["w"]
            err = np.linalg.norm(w_hip - w_or, axis=2) / (np.linalg.norm(w_or, axis=2) + 1e-300)   # [S, F]
            for k in range(S):
                ok = (tv[i] - cnt[k] >= 8) & (cnt[k] >= 1)            # >= 8 interference frames (7 microphones), a target
                well.extend(err[k][ok].tolist())
                ill.extend(err[k][~ok].tolist())
            mask_span.append(float((m[:S, :, i, :tv[i]].max(axis=2) - m[:S, :, i, :tv[i]].min(axis=2)).mean()))
        well, ill = np.array(well), np.array(ill)
        wav_err = [rel_rms(wavs[k], ow[k]) for k in range(S)]
        _report("unconditioned_weights_12s", {
            "segments": nseg, "range_fallbacks": 0,
            "mean_mask_range_over_time": round(float(np.mean(mask_span)), 4),
            "bins_well_conditioned_fraction": round(float(len(well) / (len(well) + len(ill))), 4),
            "w_rel_err_well_conditioned_median_p99_max": [float(np.median(well)), float(np.percentile(well, 99)), float(well.max())] if len(well) else None,
            "w_rel_err_ill_conditioned_median_max": [float(np.median(ill)), float(ill.max())] if len(ill) else None,
            "waveform_rel_rms_vs_oracle_float64_on_the_same_masks": wav_err})
        if len(well):
            assert float(np.median(well)) < 1e-4
        for k in range(S):   # measured 3e-6: on THIS input even the raw initial weights leave the systems solvable
            assert wav_err[k] < 1e-4, (k, wav_err)
    finally:
        sep.close()


# ------------------------------------------------------------------------------------------------ full size
def test_full_size_60s_properties(L, CSS, sep_mc, mix60):
    """BASELINE.json configs[1] at full size, through size-independent properties:
    determinism, shard invariance (2 virtual ranks == fused run, bit for bit), gain linearity of the
    front end (features are scale-invariant, MVDR is linear), and frame-local support of the output."""
    import torch
    PAR = pkg("parallel")
    cfg, _ = cfgs(CSS)
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    h = sep_mc.handle
    w1 = h.run(mix60[0], run_cfg)
    plan = h.get_plan()
    assert (plan.mix_frames, plan.num_segments, plan.n_out, plan.last_valid) == (3749, 40, 960000, 122)
    w2 = h.run(mix60[0], run_cfg)
    assert np.array_equal(w1, w2)                                                # deterministic
    perms = h.read(L.BUF_PERMS)
    assert sorted(map(tuple, np.sort(perms, axis=1))) == [(0, 1, 2)] * 40        # every row is a permutation

    # virtual ranks on the same GPU (fresh sessions, buffers poisoned in between)
    results = {world: virtual_rank_run(PAR, L, h, mix60[0], run_cfg, world, poison=True) for world in (1, 2, 3)}
    assert np.array_equal(results[1], w1)
    assert np.array_equal(results[2], w1)
    assert np.array_equal(results[3], w1)

    # css_run_device (samples and waveforms in HBM): the plain stage sequence by default, the unit pipeline on request;
    # both are the host-to-host pass's result, bit for bit
    pcm_dev = torch.from_numpy(np.ascontiguousarray(mix60[0])).cuda()
    wav_dev = torch.empty((3, int(plan.n_out)), dtype=torch.float32, device="cuda")
    for pipelined in (0, 1):
        h.set_tuning("pipeline_device", pipelined)
        wav_dev.zero_()
        h.run_device(pcm_dev.data_ptr(), mix60.shape[1], 7, run_cfg, wav_dev.data_ptr(), int(plan.n_out))
        torch.cuda.synchronize()
        assert np.array_equal(wav_dev.cpu().numpy(), w1), pipelined
    h.set_tuning("pipeline_device", 0)

    # gain linearity: the features are scale-invariant up to their eps clamps and the beamformer is linear,
    # so x -> 0.5 x halves the output (up to float32-rounding-level winner-take-all flips)
    wh = h.run(0.5 * mix60[0], run_cfg)
    assert rel_rms(wh, 0.5 * w1) < 1e-3


def test_batched_forward_is_stft_then_separate(L, sep_mc, sep_sc, mix60):
    """ConformerCssWrapper.forward (conformer_wrapper.py:58-77) fused on the device (css_forward_host, the validation
    forward of the training loop) gives bit for bit what the stft -> separate protocol calls give, for a batch of clips,
    multi- and single-channel, and rejects what the reference's assert rejects."""
    import torch
    clips = np.stack([mix60[0, s:s + 48000] for s in (0, 16000, 40000, 123456, 300000)])           # [5, 48000, 7]
    out = sep_mc.forward(torch.from_numpy(clips))
    ref = sep_mc.separate(sep_mc.stft(torch.from_numpy(clips)))
    assert tuple(out["spk_masks"].shape) == (5, F, T, 3) and tuple(out["noise_masks"].shape) == (5, F, T, 1)
    assert torch.equal(out["spk_masks"], ref["spk_masks"]) and torch.equal(out["noise_masks"], ref["noise_masks"])
    short = clips[:2, :20000, :1]                                                                     # [2, 20000, 1]
    o1 = sep_sc.forward(torch.from_numpy(short))
    r1 = sep_sc.separate(sep_sc.stft(torch.from_numpy(short[:, :, 0])))
    assert tuple(o1["spk_masks"].shape) == (2, F, 77, 3) and torch.equal(o1["spk_masks"], r1["spk_masks"])
    with pytest.raises(AssertionError):
        sep_mc.forward(torch.from_numpy(short))                                                       # 1 channel into the MC model


@pytest.mark.parametrize("dims", [(256, 4, 512, 33), (768, 12, 1536, 33), (512, 8, 1024, 17), (512, 8, 1056, 31),
                                  (512, 8, 1024, 33, 100)])
def test_other_model_widths_vs_oracle(L, mix_stage, dims):
    """Narrower / wider models and another depthwise kernel size exercise the other template instantiations
    (LayerNorm NV = 1 / 3, the fused conv module for D = 256 and its two-kernel fallback, GEMM tails), and a
    relative-position table SHORTER than the segment (maxlen 100 < 186 frames: the clamp of conformer.py:24, baked into
    the attention kernel's position operands): masks of a 2-block model against the oracle in both arithmetic modes."""
    import torch
    w = pkg("weights")
    D, H, FFu, ks = dims[:4]
    extra = dict(maxlen=dims[4]) if len(dims) > 4 else {}
    desc = w.ModelDesc(attention_dim=D, attention_heads=H, linear_units=FFu, num_blocks=2, kernel_size=ks, **extra)
    st = w.apply_golden_recipe(w.portable_state_dict(desc, 31))
    params = O.ConformerParams(st)
    sep = pkg("separator").HipSeparator(st, None, device=0)
    try:
        xo = O.stft(mix_stage[0])
        om = O.conformer_forward(params, O.features(xo[:, :T]))                        # [4, F, T]
        for mode in ("split_f16", "exact_f32"):
            sep.handle.set_linear_mode(mode)
            out = sep.forward(torch.from_numpy(mix_stage[:, :48000]))
            m = np.concatenate([out["spk_masks"].numpy()[0], out["noise_masks"].numpy()[0]], axis=-1)   # [F, T, 4]
            assert np.abs(np.moveaxis(m, 2, 0) - om).max() < 5e-5, (dims, mode)
    finally:
        sep.close()


@pytest.mark.parametrize("frames", [2, 8, 24, 32, 33, 48, 64, 65, 97, 256, 371, 499])
def test_short_and_odd_segment_lengths_vs_oracle(L, mix60, frames):
    """Every tile schedule of the attention kernel (one instantiation per ceil(T / 32); T <= 32 has a two-tile position
    prologue), the conv module's short runs and the GEMM tails at clip lengths the 3 s / 4 s configurations never reach:
    masks of a 2-block model against the oracle in both arithmetic modes, multi-channel.  The two modes see the same
    features, so they must agree to the arithmetic's own noise (1e-5); against the oracle the bar is the one of the
    other mask tests (5e-5): the oracle's float64-evaluated IPD features differ from the float32 ones in rare,
    ill-conditioned elements (numerical hazard 3), more so the fewer frames the statistics average over."""
    import torch
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=2)
    st = w.apply_golden_recipe(w.portable_state_dict(desc, 17))
    params = O.ConformerParams(st)
    sep = pkg("separator").HipSeparator(st, None, device=0)
    try:
        n = (frames - 1) * 256 + 512
        clips = np.stack([mix60[0, s:s + n] for s in ((0, 1000, 9000) if frames == 499 else (0, 1000, 5000))])   # [3, n, 7]
        got = {}
        for mode in ("split_f16", "exact_f32"):
            sep.handle.set_linear_mode(mode)
            out = sep.forward(torch.from_numpy(clips))
            assert tuple(out["spk_masks"].shape) == (3, F, frames, 3)
            got[mode] = np.concatenate([out["spk_masks"].numpy(), out["noise_masks"].numpy()], axis=-1)   # [3, F, T, 4]
        assert np.abs(got["split_f16"] - got["exact_f32"]).max() < 1e-5, frames
        for b in range(3):
            feat = O.features(O.stft(clips[b]))
            # (the clips are chosen OFF the atan2 branch cut of the IPD features -- DESIGN.md hazard 7: within four float32
            # ulps of +-pi the reference itself is discontinuous, and one angle landing on the other side moves every mask
            # of the clip by ~1e-4 through the attention; with 499 frames x 1542 angle rows a clip at offset 5000 is on it)
            assert frames < 64 or np.abs(np.abs(feat[257:].reshape(6, 257, -1)[:, 1:256]) - np.pi).min() > 1e-6, (frames, b)
            om = O.conformer_forward(params, feat)                                        # [4, F, frames]
            for mode, m in got.items():
                assert np.abs(np.moveaxis(m[b], 2, 0) - om).max() < 5e-5, (frames, mode, b)
    finally:
        sep.close()


def test_eight_second_segments_dense_hop_vs_oracle(L, CSS, mix60):
    """segments of 371 .. 512 frames take kernels no fixture reached against an expected value -- the covariance
    kernel's 82 KB LDS launch (scm_kernel<8>), features_kernel<512>, the attention's 11 .. 16 key tiles.  8 s segments (499
    frames) every second (eight segments over a frame: the general overlap-add loops) on a 21 s recording, a 2-block
    model: masks of the first and the ragged last segment against the oracle's features + Conformer, decisions exact,
    covariances and beamformer weights of a segment against the oracle on the HIP masks, waveforms <= 1e-4 against the
    oracle's float64 chain on the HIP masks."""
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=2)
    st = w.apply_golden_recipe(w.portable_state_dict(desc, 23))
    params = O.ConformerParams(st)
    mix = np.ascontiguousarray(mix60[:, 8000:8000 + 21 * 16000 + 77])
    cfg, ocfg = cfgs(CSS, segment_size_sec=8.0, hop_size_sec=1.0)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=5)
    try:
        wavs, side = CSS.separate_and_stitch(mix, sep, 16000, "cuda:0", cfg)
        h = sep.handle
        plan = h.get_plan()
        nseg, Ts = int(plan.num_segments), 499
        assert side["segment_frames"] == Ts and nseg >= 12
        m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, Ts)
        X = O.stft(mix[0])
        for i in (0, nseg - 1):
            seg = np.zeros((F, Ts, 7), np.complex64)
            part = X[:, i * 62:i * 62 + Ts]
            seg[:, :part.shape[1]] = part
            om = O.conformer_forward(params, O.features(seg))
            assert np.abs(m[:, :, i] - om).max() < 5e-5, i
        hip_masks = [(np.moveaxis(m[:S, :, i], 0, 2), np.moveaxis(m[S:, :, i], 0, 2)) for i in range(nseg)]
        taps = {}
        ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg, separate_fn=lambda i, seg: hip_masks[i],
                                          mvdr_cplx=np.complex128, taps=taps)
        assert oside["plan"].hop_frames == 62 and oside["plan"].num_segments == nseg
        assert np.array_equal(h.read(L.BUF_PERMS), np.array(oside["perms"]))
        assert np.array_equal(side["activity_final"].numpy(), oside["activity_final"])
        # covariances (scm_kernel<8>) and beamformer weights of segment 1: [S + 1][F][7 real diagonal + 21 complex] / [S][F][7]
        scm = h.read(L.BUF_SCM)[1]
        o_scm = taps["mvdr1"]["scm"]                                  # [4, F, 7, 7] complex128
        iu = np.triu_indices(7, 1)
        packed = np.concatenate([np.real(np.diagonal(o_scm, axis1=2, axis2=3)),
                                 np.stack([o_scm[:, :, iu[0], iu[1]].real, o_scm[:, :, iu[0], iu[1]].imag], -1).reshape(4, F, 42)], -1)
        assert rel_rms(scm, packed) < 5e-6
        bfw = h.read(L.BUF_BFW)[1].reshape(S, F, 7, 2)
        assert rel_rms(bfw[..., 0] + 1j * bfw[..., 1], taps["mvdr1"]["w"]) < 1e-4
        for k in range(S):
            assert rel_rms(wavs[k], ow[k]) < 1e-4, k
    finally:
        sep.close()


def test_random_lengths_and_knobs_decisions_vs_oracle(L, CSS, mix60):
    """Stitching and gating on RANDOM recording lengths and CssCfg knobs (seeded): every length lands the last segment's valid
    frames somewhere else in 94 .. 186 (css.py:185-190), with random activity thresholds, dilation / erosion lengths, stitching
    losses and window margins.  The decisions -- every permutation, both activity maps -- are the oracle's stitching stage's on
    the same (HIP) masks, and the queue gives the synchronous call's bits.  A one-block model: what is tested here does not depend
    on the estimator."""
    import css_oracle as O
    W = pkg("weights")
    desc = W.ModelDesc(num_blocks=1)
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 31))
    sep = pkg("separator").HipSeparator(state, None, device=0, max_batch_segments=64)
    rs = np.random.RandomState(606)
    F, S = 257, 3
    try:
        h = sep.handle
        for case in range(20):
            n = int(rs.randint(48_320 + 256, 16 * 16000))
            off = int(rs.randint(0, mix60.shape[1] - n))
            kw = dict(activity_th=float(rs.choice([0.2, 0.3, 0.4, 0.55])), activity_dilation_sec=float(rs.choice([0.0, 0.2, 0.4, 0.8])),
                      activity_erosion_sec=float(rs.choice([0.0, 0.1, 0.2, 0.5])), stitching_loss=str(rs.choice(["l1", "mse"])),
                      seg_weight_m0_sec=float(rs.choice([0.0, 0.15, 0.3])), seg_weight_m1_sec=float(rs.choice([0.3, 0.45, 0.6])))
            run_cfg = CSS.make_run_cfg(CSS.CssCfg(show_progressbar=False, **kw), 16000, 7)
            pcm = L.pinned_copy(np.ascontiguousarray(mix60[0, off:off + n]))
            w = h.run(pcm, run_cfg).copy()
            plan = h.get_plan()
            T = run_cfg.c.segment_frames
            m = h.read(L.BUF_MASKS).reshape(S + 1, F, int(plan.num_segments), T)
            perms = h.read(L.BUF_PERMS)
            act_b = h.read(L.BUF_ACT_B).astype(bool)
            act_f = h.read(L.BUF_ACT_FINAL).astype(bool)
            hip_masks = lambda i, seg=None: (np.ascontiguousarray(np.moveaxis(m[:S, :, i], 0, 2)), np.ascontiguousarray(np.moveaxis(m[S:, :, i], 0, 2)))
            _, side = O.separate_and_stitch(pcm[None], None, 16000, O.OracleCssCfg(mc_mvdr=False, **kw), separate_fn=hip_masks)
            assert side["plan"].num_segments == plan.num_segments and side["plan"].mix_frames == plan.mix_frames, (case, n)
            assert np.array_equal(np.array(side["perms"], dtype=np.int32), perms), (case, n, kw)
            diff = np.argwhere(side["activity_b"].T != act_b)
            for s_, t_ in diff:      # a bit may differ only where the mean mask sits on the threshold to float32 rounding
                assert abs(float(side["activity"][t_, s_]) - kw["activity_th"]) < 2e-6, (case, s_, t_)
            if len(diff) == 0:
                assert np.array_equal(side["activity_final"][0].T, act_f), (case, n, kw)
            out = L.pinned_empty(w.shape, np.float32)
            h.run_enqueue(pcm, run_cfg, out)
            h.wait()
            assert np.array_equal(out, w), case
    finally:
        sep.close()


@pytest.mark.parametrize("seconds,worlds", [(4.0, (2, 3, 8)), (7.6, (3, 5, 8)), (13.0, (7, 8))])
def test_more_ranks_than_segments(L, CSS, sep_mc, mix60, seconds, worlds):
    """A short recording sharded over MORE ranks than it has segments (2, 4 and 8 segments over up to 8 ranks): the ranks
    without a segment own no frames and take part in every exchange with empty pieces; the result is the fused pass's, bit for bit."""
    PAR = pkg("parallel")
    n = int(seconds * 16000) + 17
    pcm = L.pinned_copy(np.ascontiguousarray(mix60[0, 3000:3000 + n]))
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep_mc.handle
    ref = h.run(pcm, run_cfg).copy()
    nseg = int(h.get_plan().num_segments)
    for world in worlds:
        plans = PAR.all_plans(nseg, int(h.get_plan().mix_frames), int(h.get_plan().stft_frames), 186, 93, 256, world)
        assert sum(1 for p in plans if p.own_seg_hi == p.own_seg_lo) == max(world - nseg, 0)
        out = virtual_rank_run(PAR, L, h, pcm, run_cfg, world)
        assert np.array_equal(out, ref), (seconds, world, nseg)

"""Frame sizes other than 512 / 256 (ExtractorCfg.frame_len / frame_hop; init_kernel feature.py:19-45) against fixtures from
the reference (tests/golden/gen_golden_r5b.py): frame_len 400 / hop 160 and frame_len 512 / hop 128, both on 512 FFT points
(the only FFT size the reference's wrapper can run: its network has 257 mask bins whatever the extractor says).  The oracle
on the CPU, the HIP path -- DFT-matrix analysis product, general waveform overlap-add, the plain stage sequence -- on the
GPU."""
import numpy as np
import pytest

import css_oracle as O
from conftest import pkg, rel_rms, take_windows
from test_oracle_golden_r2 import unpack2, unpack_bits

F, S = 257, 3
GEOMETRIES = ("f400_160", "f512_128")


def _mix(g):
    return pkg("synth").synth_meeting(60.0, 7, seed=int(g["mix_seed"]))[:, int(g["mix_offset"]):int(g["mix_offset"]) + int(g["mix_samples"])]


@pytest.mark.parametrize("name", GEOMETRIES)
def test_oracle_transforms_and_first_segment(golden, mc_state, name):
    g = golden("frames_r5.npz")
    fl, fh, nfft = (int(v) for v in g[name + "_frame"])
    mix = _mix(g)
    x = O.stft(np.ascontiguousarray(mix[0, :48000]), frame_len=fl, frame_hop=fh, n_fft=nfft)
    Ts = int(g[name + "_segment_frames"])
    assert x.shape == (F, Ts, 7) and Ts == (48000 - fl) // fh + 1
    ref = g[name + "_stft"]
    assert np.abs(x[::4, ::5] - ref).max() <= 2e-6 * np.abs(ref).max()
    m = O.conformer_forward(O.ConformerParams(mc_state[0]), O.features(x))
    assert np.abs(np.moveaxis(m[:S], 0, 2)[::8, ::4] - g[name + "_masks_spk_seg0"]).max() < 1.5e-5
    # the synthesis transform inverts the analysis one up to the windows' product (sqrt-Hann / S against Hann): a round trip
    # of a spectrum through istft -> stft is linear; here only its shape and the overlap count are pinned
    w = O.istft(np.moveaxis(x, 2, 0)[:1], frame_len=fl, frame_hop=fh, n_fft=nfft)
    assert w.shape == (1, (Ts - 1) * fh + fl)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GEOMETRIES)
def test_hip_path_vs_reference(golden, mc_state, name):
    L, CSS, SEP = pkg("_lib"), pkg("css"), pkg("separator")
    if L.load().css_device_count() < 1:
        pytest.fail("no HIP device visible: the parity tests must run on the GPU box")
    g = golden("frames_r5.npz")
    fl, fh, nfft = (int(v) for v in g[name + "_frame"])
    mix = _mix(g)
    cfg = SEP.ConformerCssCfg(extractor_conf=SEP.ExtractorCfg(frame_len=fl, frame_hop=fh),
                              nnet_conf=SEP.NnetCfg(conformer_conf=SEP.ConformerCfg(attention_dim=512, attention_heads=8, num_blocks=18, dropout_rate=0.0)))
    sep = SEP.HipSeparator(mc_state[0], cfg, device=0, max_batch_segments=16)
    try:
        assert (sep.desc.frame_len, sep.desc.frame_hop, sep.desc.num_bins) == (fl, fh, nfft // 2 + 1)
        h = sep.handle
        Ts = int(g[name + "_segment_frames"])
        # the analysis transform alone (separator protocol)
        planes = h.stft_host(np.ascontiguousarray(mix[0, :48000]))                      # [7, 2F, T]
        x = (planes[:, :F] + 1j * planes[:, F:]).transpose(1, 2, 0)
        ref = g[name + "_stft"]
        assert x.shape == (F, Ts, 7) and np.abs(x[::4, ::5] - ref).max() <= 2e-6 * np.abs(ref).max()
        css_cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
        run_cfg = CSS.make_run_cfg(css_cfg, 16000, 7, fl, fh)
        assert int(run_cfg.c.segment_frames) == Ts
        pcm = np.ascontiguousarray(mix[0])
        X_all = O.stft(pcm, frame_len=fl, frame_hop=fh, n_fft=nfft)
        wta = unpack2(g[name + "_wta_packed"], g[name + "_wta_shape"])
        nseg = wta.shape[0]
        shape = tuple(g[name + "_activity_shape"])
        for mode in ("split_f16", "exact_f32"):
            h.set_linear_mode(mode)
            free = h.run(pcm, run_cfg)                                                   # the fused call: plain stage sequence
            assert free.shape == (S, int(g[name + "_wav_len"])) and h.get_plan().num_segments == nseg
            m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, Ts)
            assert np.abs(np.moveaxis(m[:S, :, 0], 0, 2)[::8, ::4] - g[name + "_masks_spk_seg0"]).max() < 1.5e-5
            per_seg = [int((np.argmax(m[:, :, i], axis=0) != wta[i]).sum()) for i in range(nseg)]
            # rounding-level ties only -- except in a segment with an IPD feature ON the atan2 branch cut (DESIGN.md hazard 7:
            # the reference is discontinuous there), identified by the feature itself
            hop_f = int(run_cfg.c.hop_frames)

            def on_cut(i):
                seg = np.zeros((F, Ts, 7), np.complex64)
                part = X_all[:, i * hop_f:i * hop_f + Ts]
                seg[:, :part.shape[1]] = part
                f = O.features(seg)[F:].reshape(6, F, -1)[:, 1:F - 1]
                return bool(np.abs(np.abs(f) - np.pi).min() < 1e-6)
            cut = [i for i in range(nseg) if per_seg[i] > 3 and on_cut(i)]
            assert sum(nf for i, nf in enumerate(per_seg) if i not in cut) <= 1e-5 * wta.size + 3, per_seg
            assert all(per_seg[i] <= 0.005 * F * Ts for i in cut) and len(cut) <= 1, (cut, per_seg)
            perms = h.read(L.BUF_PERMS)
            assert [tuple(p) for p in perms[1:]] == [tuple(p) for p in g[name + "_pit_perm"]]
            assert np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, unpack_bits(g[name + "_activity_final"], shape))
            ms = np.abs(h.read(L.BUF_MASK_ST).transpose(1, 2, 0)[::16, ::8] - g[name + "_mask_stitched"])
            ok_t = np.ones(ms.shape[1] * 8, bool)
            for i in cut:
                ok_t[i * hop_f:i * hop_f + Ts] = False
            assert ms[:, ok_t[::8][:ms.shape[1]]].max() < 1.5e-5
            # waveforms on the reference's decisions (stage by stage), windows before the ragged last segment
            h.begin(pcm, pcm.shape[0], 7, run_cfg)
            TL = int(h.get_plan().mix_frames)
            h.write(L.BUF_WTA_OVERRIDE, wta)
            h.stage_stft(); h.stage_masknet(0, nseg); h.stage_mvdr(0, nseg)
            h.stage_pit_costs(0, nseg - 1); h.stage_pit_scan(); h.stage_stitch(0, TL); h.stage_istft(0, TL)
            forced = h.read(L.BUF_WAV)
            n_ok = (nseg - 1) * int(run_cfg.c.hop_frames) * fh            # samples before the last (ragged) segment's frames
            for k in range(S):
                assert rel_rms(forced[k, :n_ok:64], g[name + "_wav_dec64"][k][:n_ok // 64 + (n_ok % 64 > 0)]) < 1e-4, (mode, k)
            ww, wr = take_windows(forced, 4), g[name + "_wav_windows"]
            for k in range(S):
                assert rel_rms(ww[k][:3], wr[k][:3]) < 1e-4, (mode, k)
            if sum(per_seg) == 0 and not cut:
                for k in range(S):
                    assert rel_rms(free[k, :n_ok:64], g[name + "_wav_dec64"][k][:n_ok // 64 + (n_ok % 64 > 0)]) < 1e-4, (mode, k)
        h.set_linear_mode("exact_f32")
        # the Python drop-in on the same separator object
        wavs, side = CSS.separate_and_stitch(mix, sep, 16000, "cuda:0", css_cfg)
        assert int(side["segment_frames"]) == Ts and np.array_equal(np.stack(wavs), h.run(pcm, run_cfg))
        # ---- round 6: the sharded driver for this geometry (the seam carries the synthesis rows of the last ovl - 1 frames): 2, 3 and
        # 8 virtual ranks, own ranges and the all-gathered whole, bit for bit the fused pass -- in both arithmetic modes
        from test_hip_parity import virtual_rank_run
        PAR = pkg("parallel")
        for mode in ("exact_f32", "split_f16"):
            h.set_linear_mode(mode)
            fused = h.run(pcm, run_cfg).copy()
            for world in (2, 3, 8):
                assert np.array_equal(virtual_rank_run(PAR, L, h, pcm, run_cfg, world), fused), (mode, world)
        h.set_linear_mode("exact_f32")
        # ---- ... and the PCM16 wav edges (css_run_pcm16, css_run_enqueue_pcm16): bit for bit load_audio -> css_run -> write_wav
        q16 = np.clip(np.rint(pcm * 0.2 * 32768.0), -32768, 32767).astype(np.int16)
        planes = [np.ascontiguousarray(q16[:, c]) for c in range(7)]
        fl32 = np.ascontiguousarray(q16.astype(np.float32) / np.float32(32768.0))
        wav = h.run(fl32, run_cfg)
        got16, peaks = h.run_pcm16(planes, run_cfg)
        for i in range(S):
            assert peaks[i] == np.max(np.abs(wav[i]))
            y = wav[i] * 0.99 / (np.max(np.abs(wav[i])) + 1e-7)                         # utils/audio_utils.py:44-45
            assert np.array_equal(got16[i], np.clip(np.rint(y.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16))
        pl16 = L.pinned_empty((7, q16.shape[0]), np.int16)
        pl16[:] = q16.T
        o16, pk = L.pinned_empty(got16.shape, np.int16), L.pinned_empty((S,), np.float32)
        h.run_enqueue_pcm16([pl16[c] for c in range(7)], run_cfg, o16, pk)
        h.wait()
        assert np.array_equal(o16, got16) and np.array_equal(pk, peaks)
    finally:
        sep.close()


@pytest.mark.gpu
def test_enqueue_of_another_frame_size_applies_the_range_rule(mc_state, golden):
    """on a handle with frame_len / hop other than 512 / 256 css_run_enqueue runs the pass inside the call; a pass
    that leaves the split-f16 range must be repeated in float32 (or refused with CSS_ERR_RANGE) there, as css_run does."""
    L, CSS, SEP = pkg("_lib"), pkg("css"), pkg("separator")
    g = golden("frames_r5.npz")
    st = dict(mc_state[0])
    key = pkg("weights").PREFIX + "conformer.encoders.0.feed_forward_in.net.0.weight"
    st[key] = np.asarray(st[key], np.float32) * np.float32(3e5)          # ReLU outputs past 65504: the next split operand overflows
    cfg = SEP.ConformerCssCfg(extractor_conf=SEP.ExtractorCfg(frame_len=400, frame_hop=160),
                              nnet_conf=SEP.NnetCfg(conformer_conf=SEP.ConformerCfg(attention_dim=512, attention_heads=8, num_blocks=18, dropout_rate=0.0)))
    sep = SEP.HipSeparator(st, cfg, device=0, max_batch_segments=16, linear_mode="split_f16")
    try:
        h = sep.handle
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7, 400, 160)
        pcm = L.pinned_copy(np.ascontiguousarray(_mix(g)[0, :6 * 16000]))
        h.set_linear_mode("exact_f32")
        ref = h.run(pcm, run_cfg).copy()
        h.set_linear_mode("split_f16")
        before = h.range_status()[0]
        out = L.pinned_empty(ref.shape, np.float32)
        out[:] = np.nan
        h.run_enqueue(pcm, run_cfg, out)
        h.wait()
        assert h.range_status() == (before + 1, True) and h.linear_mode() == "split_f16"
        assert np.array_equal(out, ref)                                   # the float32 repeat's result, bit for bit
        h.set_range_fallback(False)
        with pytest.raises(L.CssError) as e:
            h.run_enqueue(pcm, run_cfg, out)
        assert e.value.code == L.CSS_ERR_RANGE
        h.wait()
    finally:
        sep.close()

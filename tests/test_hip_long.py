"""BASELINE.json configs[3] at full size: the fixed 1800 s, 7-channel meeting (1209 segments, T_long 112499).

On one MI355X: the fused pass; the sharded driver for 8 virtual ranks (== fused, bit for bit); the decisions
(permutations, activity bits) of the whole meeting against the oracle's stitching stage driven by the HIP masks; the
waveforms of a window of segments against the oracle's full chain (float64 MVDR) on the HIP masks; and the real
multi-process driver (two processes, gloo, both on GPU 0 -- RCCL refuses two ranks on one device; the 8-GPU RCCL run is
the driver's) on a shorter meeting.  Needs an MI355X; about two minutes, most of it the synthetic meeting's generator
and the oracle."""
import os
import sys

import numpy as np
import pytest

import css_oracle as O
from conftest import ROOT, pkg, rel_rms
from test_hip_parity import virtual_rank_run

pytestmark = pytest.mark.gpu

F, T, S, HOP = 257, 186, 3, 93
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def L():
    lib = pkg("_lib")
    if lib.load().css_device_count() < 1:
        pytest.fail("no HIP device visible: the parity tests must run on the GPU box")
    return lib


@pytest.fixture(scope="module")
def long_run(L, mc_state):
    """The 30-min meeting through the fused pass (css_run, page-locked buffers), kept for the tests below."""
    CSS = pkg("css")
    mix = pkg("synth").synth_meeting(1800.0, 7, seed=1)          # same weights and biases as the 60 s config, no re-calibration
    pcm = L.pinned_copy(np.ascontiguousarray(mix[0]))
    del mix
    sep = pkg("separator").HipSeparator(mc_state[0], None, device=0, max_batch_segments=128)
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    h = sep.handle
    wav = h.run(pcm, run_cfg).copy()
    yield dict(pcm=pcm, sep=sep, h=h, run_cfg=run_cfg, wav=wav)
    sep.close()


def test_plan_and_determinism(L, long_run):
    h = long_run["h"]
    p = h.get_plan()
    assert (p.n_samples, p.stft_frames, p.mix_frames, p.num_segments) == (28_800_000, 112_499, 112_499, 1209)
    assert p.n_out == (112_499 - 1) * 256 + 512
    wav = long_run["wav"]
    assert wav.shape == (3, p.n_out) and np.isfinite(wav).all()
    perms = h.read(L.BUF_PERMS)
    assert np.array_equal(np.sort(perms, axis=1), np.tile(np.arange(3, dtype=np.int32), (1209, 1)))
    again = h.run(long_run["pcm"], long_run["run_cfg"])
    assert np.array_equal(again, wav)                                                     # deterministic


def test_eight_virtual_ranks_equal_the_fused_pass(L, long_run):
    """The segment-sharded driver, 8 ranks played one after the other on this GPU (every rank replays its own session;
    exchanges by stacking the pieces): the stitched streams are the fused pass's, bit for bit."""
    PAR = pkg("parallel")
    plans = PAR.all_plans(1209, 112_499, 112_499, T, HOP, 256, 8)
    assert sum(p.seg_hi - p.seg_lo for p in plans) == 1209 + 7                            # one halo segment per seam
    assert all(151 <= p.own_seg_hi - p.own_seg_lo <= 152 for p in plans)
    out = virtual_rank_run(PAR, L, long_run["h"], long_run["pcm"], long_run["run_cfg"], 8)
    ref = long_run["wav"]
    if not np.array_equal(out, ref):
        bad = np.flatnonzero((out != ref).any(axis=0))
        fr = bad // 256
        owners = sorted({next(p.rank for p in plans if p.t_lo <= min(f, 112_498) < p.t_hi) for f in (fr[0], fr[-1])})
        pytest.fail(f"{bad.size} samples differ: first at {bad[0]} (frame {fr[0]}), last at {bad[-1]} (frame {fr[-1]}), "
                    f"ranks {owners}; seams at frames {[p.t_lo for p in plans]}")


def test_decisions_and_window_vs_oracle(L, long_run, mc_state):
    """Whole-meeting decisions against the oracle's stitching stage fed with the HIP masks (permutations and
    thresholded / dilated / eroded activity depend on the masks only: stitching_input = 'mask'), then a window of 12
    segments through the oracle's full chain (float64 MVDR) on the same masks: waveforms <= 1e-4 relative RMS."""
    h, run_cfg = long_run["h"], long_run["run_cfg"]
    h.run(long_run["pcm"], run_cfg)
    nseg = 1209
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
    perms = h.read(L.BUF_PERMS)
    act_b = h.read(L.BUF_ACT_B).astype(bool)          # [S, T_long]
    act_f = h.read(L.BUF_ACT_FINAL).astype(bool)
    mix = long_run["pcm"][None]

    def hip_masks(i, seg=None):
        return (np.ascontiguousarray(np.moveaxis(m[:S, :, i], 0, 2)), np.ascontiguousarray(np.moveaxis(m[S:, :, i], 0, 2)))

    # (a) decisions of all 1209 segments: the oracle without the beamformer (it does not enter these decisions)
    ocfg = O.OracleCssCfg(activity_th=0.3, mc_mvdr=False)
    _, side = O.separate_and_stitch(mix, None, 16000, ocfg, separate_fn=hip_masks)
    assert side["plan"].num_segments == nseg and side["plan"].mix_frames == 112_499
    assert np.array_equal(np.array(side["perms"], dtype=np.int32), perms)
    th = np.float32(0.3)
    diff = np.argwhere(side["activity_b"].T != act_b)
    for s_, t_ in diff:   # a bit may differ only where the mean mask sits on the threshold to float32 rounding
        assert abs(float(side["activity"][t_, s_]) - float(th)) < 2e-6, (s_, t_, side["activity"][t_, s_])
    assert len(diff) <= 4
    if len(diff) == 0:
        assert np.array_equal(side["activity_final"][0].T, act_f)

    # (b) a window of 12 segments (global 600..611) through the oracle's full chain on the HIP masks
    s0, ns = 600, 12
    a = s0 * HOP * 256
    nsub = ((ns - 1) * HOP + T - 1) * 256 + 512
    sub = mix[:, a:a + nsub]
    ow, oside = O.separate_and_stitch(sub, None, 16000, O.OracleCssCfg(activity_th=0.3),
                                      separate_fn=lambda i, seg: hip_masks(s0 + i), mvdr_cplx=np.complex128)
    assert oside["plan"].num_segments == ns
    # interior frames: only window segments contribute, and the dilate / erode halo (36 frames) stays inside the window
    f_lo, f_hi = HOP + 40, (ns - 1) * HOP - 40
    lo, hi = (f_lo + 1) * 256, (f_hi - 1) * 256
    wav = long_run["wav"]
    for j in range(S):
        k = int(perms[s0][j])        # the window's run starts from the identity, the meeting's from perms[s0]
        ref = ow[k][lo:hi]
        got = wav[j][a + lo:a + hi]
        assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
        assert rel_rms(got, ref) < 1e-4, (j, k, rel_rms(got, ref))


def test_masks_deep_in_the_meeting_vs_oracle(L, long_run, mc_state):
    """The estimator's masks of segments 0, 600 and 1208 (the ragged last one) of the 30-min meeting against the oracle's
    own features + Conformer on the same frames: what tests/test_hip_parity.py holds for a 20 s input must hold at any
    position of any batch (segment 600 is the 88th of the fifth 121-segment batch, 1208 closes the last)."""
    h, run_cfg = long_run["h"], long_run["run_cfg"]
    h.run(long_run["pcm"], run_cfg)
    nseg = 1209
    m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
    params = O.ConformerParams(mc_state[0])
    pcm = long_run["pcm"]
    worst = {}
    for i in (0, 600, 1208):
        a = i * HOP * 256
        x = O.stft(pcm[a:a + (T - 1) * 256 + 512])          # [F, t, C], t < T for the last segment
        seg = np.zeros((F, T, 7), np.complex64)
        seg[:, :x.shape[1]] = x
        om = O.conformer_forward(params, O.features(seg))   # [S + 1, F, T]
        worst[i] = float(np.abs(m[:, :, i, :] - om).max())
        assert worst[i] < 1.5e-5, (i, worst[i])
    assert int(h.get_plan().last_valid) == 155               # segment 1208 is ragged: 155 of 186 frames
    _report("masks_1800s_vs_oracle_max_abs", worst)


@pytest.mark.parametrize("mode", ["exact_f32", "split_f16"])
def test_whole_meeting_vs_the_reference(L, long_run, golden, mode):
    """configs[3] against the REFERENCE's own run of the same 1800 s meeting (tests/golden/gen_golden_r6b.py, e2e1800_r6.npz;
    until round 6 this configuration was held to the oracle only): all 1208 permutations, both activity maps, the stitched
    masks, every segment's masks on a grid, the three waveforms every 128th sample."""
    from test_oracle_golden_r2 import unpack_bits
    g = golden("e2e1800_r6.npz")
    h, run_cfg = long_run["h"], long_run["run_cfg"]
    h.set_linear_mode(mode)
    try:
        free = h.run(long_run["pcm"], run_cfg)
        nseg, TL = 1209, 112_499
        assert int(g["num_segments"]) == nseg and free.shape == (S, int(g["wav_len"]))
        m = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
        perms = h.read(L.BUF_PERMS)
        act_b = h.read(L.BUF_ACT_B).astype(bool).T
        act_f = h.read(L.BUF_ACT_FINAL).astype(bool).T
        mask_st = h.read(L.BUF_MASK_ST)
        shape = tuple(g["activity_shape"])
        perm_diff = [i + 1 for i in range(nseg - 1) if tuple(perms[i + 1]) != tuple(g["pit_perm"][i])]
        ab = int((act_b != unpack_bits(g["activity_b"], shape)).sum())
        af = int((act_f != unpack_bits(g["activity_final"], shape)).sum())
        grid = np.moveaxis(m[:, ::32, :, ::24], (0, 1, 2, 3), (3, 1, 0, 2))            # [nseg, 9, 8, 4]
        per_seg = np.abs(grid - g["masks_grid"]).reshape(nseg, -1).max(axis=1)
        ms = np.abs(mask_st.transpose(1, 2, 0)[::32, ::16] - g["mask_stitched"])
        ref = g["wav_dec128"]
        got = free[:, ::128]
        whole = [rel_rms(got[k], ref[k]) for k in range(S)]
        # per hop block (93 frames = 186 decimated samples), relative to the stream's own RMS (a silent block has no RMS of its own)
        nb = ref.shape[1] // 186
        d = (got[:, :nb * 186].astype(np.float64) - ref[:, :nb * 186]).reshape(S, nb, 186)
        blk = np.sqrt((d ** 2).mean(axis=2)) / g["wav_rms"][:, None]
        # Which segments can be compared at all.  (a) 39 of the 1209 segments have an IPD feature on the other side of the atan2
        # branch cut than the reference's (DESIGN.md hazard 7: a value at +-pi moves by 2 pi with the last bit of the transform;
        # the oracle puts most segments of this meeting within 1e-5 of the cut, fixture key cut_distance_per_segment) -- their
        # masks differ by up to 0.07, every other segment's by at most 5e-6.  The reference is deterministic across thread
        # counts there (its features do not depend on them: self_t4_*), and moves the same way when its own conv1d runs on
        # another backend (self_t8_nomkldnn_*).  (b) Where the masks agree, the reference's own winner-take-all near-ties
        # remain: against ITSELF at 4 threads it has 1163 - 1169 of 1209 hop blocks within 1e-4 and 0.9e-4 - 1.6e-4 whole-meeting.
        differ = per_seg > 5e-6
        near = np.zeros(nb, bool)
        for i in np.flatnonzero(differ):
            near[max(i - 1, 0):min(i + 3, nb)] = True       # a segment covers two hop blocks; the gate's dilation reaches one more
        away = ~near
        dd = d[:, away].reshape(S, -1)
        rr = ref[:, :nb * 186].reshape(S, nb, 186)[:, away].reshape(S, -1).astype(np.float64)
        away_err = [float(np.sqrt((dd[k] ** 2).mean() / (rr[k] ** 2).mean())) for k in range(S)]
        self_blk = g["self_t4_hop_block_rel"]
        rep = {
            "segments": nseg, "permutations_that_differ_at_boundaries": perm_diff, "activity_b_bits_that_differ": ab, "activity_final_bits_that_differ": af,
            "segments_with_grid_masks_within_5e-6": int((~differ).sum()), "segments_beyond": np.flatnonzero(differ).tolist(),
            "largest_branch_cut_distance_of_a_segment_beyond": float(g["cut_distance_per_segment"][differ].max()) if differ.any() else 0.0,
            "grid_masks_max_abs": float(per_seg.max()), "grid_masks_max_abs_elsewhere": float(per_seg[~differ].max()),
            "grid_masks_median_of_segment_max": float(np.median(per_seg)),
            "stitched_masks_max_abs": float(ms.max()), "stitched_masks_within_5e-6": round(float((ms <= 5e-6).mean()), 6),
            "waveform_rel_rms_whole_meeting": whole,
            "hop_blocks": nb, "hop_blocks_within_1e-4": [int((blk[k] < 1e-4).sum()) for k in range(S)],
            "hop_blocks_within_1e-3": [int((blk[k] < 1e-3).sum()) for k in range(S)], "hop_block_worst": [float(blk[k].max()) for k in range(S)],
            "hop_blocks_away_from_those_segments": int(away.sum()),
            "of_them_within_1e-4": [int((blk[k][away] < 1e-4).sum()) for k in range(S)],
            "waveform_rel_rms_there": away_err,
            "reference_against_itself_4_threads": {"segments_beyond_5e-6": int((g["self_t4_grid_max_abs"] > 5e-6).sum()),
                                                   "hop_blocks_within_1e-4": [int((self_blk[k] < 1e-4).sum()) for k in range(S)],
                                                   "of_the_same_blocks_within_1e-4": [int((self_blk[k][away] < 1e-4).sum()) for k in range(S)],
                                                   "waveform_rel_rms_whole_meeting": [float(x) for x in g["self_t4_wav_rel_rms"]]}}
        if "self_t8_nomkldnn_grid_max_abs" in g:
            rep["reference_against_itself_other_conv_backend"] = {
                "segments_beyond_5e-6": int((g["self_t8_nomkldnn_grid_max_abs"] > 5e-6).sum()),
                "of_them_also_ours": int(((g["self_t8_nomkldnn_grid_max_abs"] > 5e-6) & differ).sum()),
                "grid_masks_max_abs": float(g["self_t8_nomkldnn_grid_max_abs"].max()),
                "hop_blocks_within_1e-4": [int((g["self_t8_nomkldnn_hop_block_rel"][k] < 1e-4).sum()) for k in range(S)],
                "waveform_rel_rms_whole_meeting": [float(x) for x in g["self_t8_nomkldnn_wav_rel_rms"]]}
        _report(f"config4_1800s_{mode}_vs_reference", rep)
        # decisions: exact, all of them
        assert perm_diff == [] and ab == 0 and af == 0, (perm_diff, ab, af)
        # masks: within the survey's 5e-6 on every segment but those with a feature across the branch cut
        assert int(differ.sum()) <= 48 and float(g["cut_distance_per_segment"][differ].max()) < 2.5e-5, rep
        assert float(np.median(per_seg)) < 3e-6
        # waveforms away from those segments: what winner-take-all near-ties leave (2 415 of the reference's decisions sit inside
        # 1e-5; against itself at 4 threads it has 96.5 % of these blocks within 1e-4 and 0.9e-4 - 1.6e-4 over the whole meeting;
        # measured here: 93.7 - 96.0 % and 1.5e-4 - 3.5e-4).  The whole meeting, the 39 segments included, is where the reference
        # is against itself on its other conv backend: 2.8e-3 / 5.1e-3 / 3.4e-3 (measured here 3.3e-3 / 7.1e-3 / 3.2e-3).
        for k in range(S):
            assert int((blk[k][away] < 1e-4).sum()) >= 0.92 * int(away.sum()), rep
            assert away_err[k] < 5e-4 and whole[k] < 1e-2, rep
    finally:
        h.set_linear_mode("exact_f32")


def _report(key, value):
    """parity bookkeeping: the measured margins of this run, next to the test output (profiles/r04_parity_coverage.json)"""
    import json
    print(f"[parity] {key}: {value}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "parity_coverage.json")
        data = {}
        if os.path.exists(path):
            try:
                with open(path) as f:
                    data = json.load(f)
            except ValueError:
                data = {}
        data[key] = value
        plain = lambda o: o.item() if hasattr(o, "item") else (o.tolist() if hasattr(o, "tolist") else str(o))   # numpy scalars / arrays
        text = json.dumps(data, indent=1, sort_keys=True, default=plain)
        with open(path + ".tmp", "w") as f:
            f.write(text)
        os.replace(path + ".tmp", path)


def _two_rank_worker(rank, world, port, seconds, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, SYN, CSS, SEP, PAR, Lm = (pkg(x) for x in ("weights", "synth", "css", "separator", "parallel", "_lib"))
        desc = W.ModelDesc.mc_v1()
        cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
        state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
        mix = SYN.synth_meeting(seconds, 7, seed=1)
        n = mix.shape[1]
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
        sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=64)
        h = sep.handle
        dev = torch.device("cuda", 0)
        be = PAR.HipShardBackend(h, dev, torch.device("cpu"))
        plan = Lm.plan(desc, run_cfg, n)
        me = PAR.make_shard_plan(int(plan.num_segments), int(plan.mix_frames), int(plan.stft_frames), 186, 93, 256, rank, world)
        s_lo, s_hi = me.pcm_range(512, n)
        piece = Lm.pinned_copy(np.ascontiguousarray(mix[0, s_lo:s_hi]))
        for _ in range(2):   # twice: the second pass reuses every persistent buffer
            be.begin(piece, n, 7, run_cfg, sample_range=(s_lo, s_hi), slice_only=True)
            out = PAR.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist)
        with be.on_stream():
            got = out.cpu().numpy()
        ref = h.run(np.ascontiguousarray(mix[0]), run_cfg)
        same = np.array_equal(ref, got)
        # as bench.py --gpus N runs it: the samples in pieces (the later ones on the copy stream, under the stages of the
        # earlier ones), every rank finishing only its own range of the result
        groups, cuts = PAR.upload_schedule(me, 186, 93, 512, n, first=3, growth=2)
        assert len(groups) >= 2 and len(cuts) == len(groups) - 1
        h.run(np.ascontiguousarray(mix[0, ::-1]), run_cfg)     # other samples in the device copy than this session's
        be.begin(piece, n, 7, run_cfg, sample_range=(s_lo, s_hi), slice_only=True, cuts=cuts)
        own, (lo, hi) = PAR.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist, gather="range", segment_groups=groups)
        with be.on_stream():
            same = same and np.array_equal(ref[:, lo:hi], own.cpu().numpy())
        np.save(os.path.join(out_dir, f"same_{rank}.npy"), np.array([same]))
        np.save(os.path.join(out_dir, f"range_{rank}.npy"), np.array([lo, hi]))
        del out, own
        be.close()
        sep.close()
    finally:
        dist.destroy_process_group()


def test_two_process_sharded_run_on_one_gpu(tmp_path):
    """parallel.sharded_separate_and_stitch over a real process group: two processes, each with its own handle on GPU 0
    and only its own slice of the recording in host memory, exchanging through gloo; both end with the fused result."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_two_rank_worker, args=(2, port, 45.0, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert bool(np.load(tmp_path / f"same_{r}.npy")[0]), f"rank {r}: sharded result differs from the fused pass"
    (lo0, hi0), (lo1, hi1) = (tuple(int(v) for v in np.load(tmp_path / f"range_{r}.npy")) for r in range(2))
    assert lo0 == 0 and hi0 == lo1 and hi1 > lo1        # the two own ranges tile the output


def test_rccl_takes_the_exchange_tensors():
    """The three all-gathers through RCCL itself (backend "nccl") in a one-rank group: the dtypes (float64 costs, uint8
    activity bits, float32 shards), the regular [world, ...] receive layout and the ordering on the handle's external
    stream are what the 8-GPU run uses."""
    import torch
    import torch.distributed as dist
    PAR = pkg("parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29900 + (os.getpid() % 90))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(torch.cuda.ExternalStream(st.cuda_stream, device=dev)):
            for t in (torch.arange(18, dtype=torch.float64, device=dev).reshape(2, 9),
                      (torch.arange(300, device=dev) % 2).to(torch.uint8).reshape(3, 100),
                      torch.linspace(-1, 1, 3 * 1024, device=dev).reshape(3, 1024)):
                got = PAR._all_gather(dist, t, 1, dev)
                assert tuple(got.shape) == (1,) + tuple(t.shape) and torch.equal(got[0], t)
    finally:
        dist.destroy_process_group()


def test_c_abi_communicator_takes_the_exchange_tensors(mc_state):
    """The same three pieces through the C ABI's own RCCL entry points (css_comm_unique_id / css_comm_init /
    css_comm_all_gather: no torch.distributed on the data path -- what a host in another language calls), in a one-rank
    communicator on the handle's stream, and through parallel._all_gather's `cabi` route."""
    import torch
    L, PAR = pkg("_lib"), pkg("parallel")
    dev = torch.device("cuda", 0)
    sep = pkg("separator").HipSeparator(mc_state[0], None, device=0)
    try:
        h = sep.handle
        uid = L.comm_unique_id()
        assert len(uid) == L.COMM_ID_BYTES and any(uid)
        with pytest.raises(L.CssError):
            h.comm_info()                                  # no communicator yet: CSS_ERR_STATE, loudly
        h.comm_init(uid, 1, 0)
        info = h.comm_info()
        assert info["nranks"] == 1 and info["rank"] == 0 and info["device"] == 0 and info["rccl_version_code"] > 20000, info
        with pytest.raises(L.CssError):
            h.comm_init(uid, 1, 0)                         # one communicator per handle
        be = PAR.HipShardBackend(h, dev, cabi_comm=True)
        with be.on_stream():
            for t in (torch.arange(18, dtype=torch.float64, device=dev).reshape(2, 9),
                      (torch.arange(300, device=dev) % 2).to(torch.uint8).reshape(3, 100),
                      torch.linspace(-1, 1, 3 * 1024, device=dev).reshape(3, 1024)):
                got = PAR._all_gather(None, t, 1, dev, cabi=be.cabi_comm)
                h.sync()
                assert tuple(got.shape) == (1,) + tuple(t.shape) and torch.equal(got[0], t)
        be.close()
        h.comm_destroy()
        h.comm_destroy()                                   # idempotent
    finally:
        sep.close()

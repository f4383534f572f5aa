"""Every way the library schedules one meeting must give the same bits: the host -> host unit pipeline (css_run), the
plain stage sequence and the unit pipeline on resident samples (css_run_device), the sharded driver for several virtual
ranks (full and own-range results, samples uploaded whole or piece by piece) -- over seeded random meeting lengths,
batch sizes and lane counts, ragged tails included.  A 2-block model keeps it to seconds.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import MODES, pkg
from test_hip_parity import virtual_rank_run

pytestmark = pytest.mark.gpu

# (seconds, max_batch_segments, lanes, arithmetic mode: the default exact_f32 and the opt-in split_f16 take turns)
CASES = [(3.02, 64, 3, 0), (3.6, 64, 3, 1), (7.7, 4, 2, 0), (11.3, 64, 3, 1), (19.01, 5, 3, 0), (26.5, 7, 4, 1), (33.3, 64, 1, 0),
         (47.9, 16, 3, 1), (61.7, 13, 2, 0), (95.2, 32, 3, 1), (26.5, 7, 4, 0), (61.7, 13, 2, 1)]


@pytest.fixture(scope="module")
def model():
    W = pkg("weights")
    desc = W.ModelDesc(num_blocks=2)
    return W.apply_golden_recipe(W.portable_state_dict(desc, 21)), desc


@pytest.mark.parametrize("seconds,max_batch,lanes,mode_ix", CASES)
def test_all_schedules_give_the_same_bits(model, seconds, max_batch, lanes, mode_ix):
    import torch
    L, CSS, PAR = pkg("_lib"), pkg("css"), pkg("parallel")
    st, desc = model
    mix = pkg("synth").synth_meeting(seconds, 7, seed=int(seconds * 10))
    n = mix.shape[1] - (int(seconds * 1000) % 200)          # ragged: not a whole number of frames
    pcm = np.ascontiguousarray(mix[0, :n])
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=max_batch, linear_mode=MODES[mode_ix])
    try:
        h = sep.handle
        h.set_lanes(lanes)
        if MODES[mode_ix] == "exact_f32":
            h.set_tuning("f32_lane_rows", 1)      # (as many lanes as set_lanes gives: the float32 mode would keep short batches on one)
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
        plan = L.plan(desc, run_cfg, n)
        ref = h.run(pcm, run_cfg).copy()                                         # pageable host memory
        assert np.isfinite(ref).all() and ref.shape == (3, plan.n_out)
        pin, out = L.pinned_copy(pcm), L.pinned_empty((3, int(plan.n_out)), np.float32)
        assert np.array_equal(h.run(pin, run_cfg, out=out), ref)                 # page-locked: asynchronous PCIe legs
        pd = torch.from_numpy(pcm).cuda()
        wd = torch.empty((3, int(plan.n_out)), dtype=torch.float32, device="cuda")
        for pipelined in (0, 1):
            h.set_tuning("pipeline_device", pipelined)
            wd.zero_()
            h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))
            torch.cuda.synchronize()
            assert np.array_equal(wd.cpu().numpy(), ref), ("css_run_device", pipelined)
        h.set_tuning("pipeline_device", 0)
        for tail_per_unit, pieces in ((1, 1), (0, 2)):
            h.set_tuning("tail_per_unit", tail_per_unit)
            h.set_tuning("tail_pieces", pieces)
            assert np.array_equal(h.run(pin, run_cfg, out=out), ref), ("tuning", tail_per_unit, pieces)
        h.set_tuning("tail_per_unit", 0)
        h.set_tuning("tail_pieces", 1)
        nseg = int(plan.num_segments)
        for world in sorted({2, min(3, nseg), min(5, nseg)}):
            if world < 2:
                continue
            assert np.array_equal(virtual_rank_run(PAR, L, h, pcm, run_cfg, world), ref), world    # full and own-range
        # one rank of two with its samples arriving in pieces (copy stream) under the stages of the earlier pieces
        if nseg >= 6:
            dev = torch.device("cuda", 0)
            be = PAR.HipShardBackend(h, dev)
            me = PAR.make_shard_plan(nseg, int(plan.mix_frames), int(plan.stft_frames), 186, 93, 256, 1, 2)
            lo, hi = me.pcm_range(512, n)
            groups, cuts = PAR.upload_schedule(me, 186, 93, 512, n, first=2, growth=2)
            h.run(np.ascontiguousarray(pcm[::-1]), run_cfg)                      # other samples in the device copy
            piece = L.pinned_copy(np.ascontiguousarray(pcm[lo:hi]))
            be.begin(piece, n, 7, run_cfg, sample_range=(lo, hi), slice_only=True, cuts=cuts)
            ss = PAR.ShardedSession(be, 3, 186, 93, 256, 1, 2, groups)
            ss.segments_and_costs()
            with be.on_stream():
                masks_piecewise = h.read(L.BUF_MASKS).copy()
            be.begin(piece, n, 7, run_cfg, sample_range=(lo, hi), slice_only=True)
            PAR.ShardedSession(be, 3, 186, 93, 256, 1, 2).segments_and_costs()
            masks_whole = h.read(L.BUF_MASKS)
            cols = slice(me.seg_lo * 186, me.seg_hi * 186)
            assert np.array_equal(masks_piecewise[:, cols], masks_whole[:, cols])
            be.close()
    finally:
        sep.close()


@pytest.mark.parametrize("seg,hop,seconds,max_batch", [(3.0, 0.5, 14.3, 64), (3.0, 0.5, 9.1, 5), (5.0, 2.5, 23.7, 64),
                                                       (8.0, 1.0, 21.2, 3), (1.0, 0.1, 6.4, 16), (9.5, 2.0, 33.1, 4)])
def test_dense_and_long_segmentations_every_schedule(model, seg, hop, seconds, max_batch):
    """Segmentations beyond the shipped 3 s / 1.5 s -- more than four segments over a frame (the general overlap-add
    loops), segments of more than 256 frames (the long-segment attention / feature / covariance kernels), of more than 512
    (592: their any-length forms), and both at once: the fused host -> host pipeline, the device-resident stage sequence and the sharded driver for 2, 3 and 5
    virtual ranks (halo = ceil(T / hop) - 1 segments) give the same bits."""
    import torch
    L, CSS, PAR = pkg("_lib"), pkg("css"), pkg("parallel")
    st, desc = model
    mix = pkg("synth").synth_meeting(seconds, 7, seed=int(seconds * 7))
    pcm = np.ascontiguousarray(mix[0, :mix.shape[1] - 123])
    n = pcm.shape[0]
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=max_batch)
    try:
        h = sep.handle
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False, segment_size_sec=seg, hop_size_sec=hop), 16000, 7)
        plan = L.plan(desc, run_cfg, n)
        ref = h.run(pcm, run_cfg).copy()
        assert np.isfinite(ref).all() and ref.shape == (3, plan.n_out) and float(np.abs(ref).max()) > 0
        pin, out = L.pinned_copy(pcm), L.pinned_empty((3, int(plan.n_out)), np.float32)
        assert np.array_equal(h.run(pin, run_cfg, out=out), ref)
        pd = torch.from_numpy(pcm).cuda()
        wd = torch.empty((3, int(plan.n_out)), dtype=torch.float32, device="cuda")
        h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))
        torch.cuda.synchronize()
        assert np.array_equal(wd.cpu().numpy(), ref)
        nseg = int(plan.num_segments)
        for world in sorted({2, min(3, nseg), min(5, nseg)}):
            if world >= 2:
                assert np.array_equal(virtual_rank_run(PAR, L, h, pcm, run_cfg, world), ref), world
    finally:
        sep.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("lanes,max_batch,pinned", [(3, 64, True), (2, 7, True), (1, 64, True), (3, 16, False)])
def test_queued_sessions_equal_synchronous_passes(model, lanes, max_batch, pinned, mode):
    """css_run_enqueue / css_wait: sessions of different lengths and contents queued back to back (page-locked buffers:
    neighbouring passes overlap -- uploads under the previous estimator, stitching / synthesis / download beside the next
    one; pageable output: the passes just queue up) give, each, the bits of its own synchronous css_run."""
    L, CSS = pkg("_lib"), pkg("css")
    st, desc = model
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=max_batch, linear_mode=mode)
    try:
        h = sep.handle
        h.set_lanes(lanes)
        sessions = []
        for k, seconds in enumerate([19.0, 33.3, 7.7, 33.3, 61.7, 3.6, 47.9, 19.0]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=100 + k)
            pcm = np.ascontiguousarray(mix[0, :mix.shape[1] - 37 * k])
            ref = h.run(pcm, run_cfg).copy()
            sessions.append((pcm, ref))
        for rounds in range(2):   # twice: the second queue starts on buffers the first one left behind
            queued = []
            for pcm, ref in sessions:
                src = L.pinned_copy(pcm) if pinned else pcm
                out = L.pinned_empty(ref.shape, np.float32) if pinned else np.empty(ref.shape, np.float32)
                out[:] = np.nan
                queued.append((h.run_enqueue(src, run_cfg, out), ref, src))
            h.wait()
            for k, (got, ref, _) in enumerate(queued):
                assert np.array_equal(got, ref), (rounds, k)
        # a synchronous call with passes still queued waits for them first
        got0 = h.run_enqueue(L.pinned_copy(sessions[0][0]), run_cfg, L.pinned_empty(sessions[0][1].shape, np.float32))
        assert np.array_equal(h.run(sessions[1][0], run_cfg), sessions[1][1])
        assert np.array_equal(got0, sessions[0][1])
        # ... and so does anything else that touches the handle's state (here: reading a buffer back)
        got2 = h.run_enqueue(L.pinned_copy(sessions[2][0]), run_cfg, L.pinned_empty(sessions[2][1].shape, np.float32))
        perms = h.read(L.BUF_PERMS)
        assert np.array_equal(got2, sessions[2][1]) and perms.shape[0] == L.plan(desc, run_cfg, sessions[2][0].shape[0]).num_segments
    finally:
        sep.close()


def test_queue_that_mixes_page_locked_and_pageable_outputs(mc_state):
    """One queue, outputs alternately page-locked (passes overlap: level word / sample buffer by parity, tail beside the
    next estimator) and pageable (passes queue up on the main stream): the two modes order the shared buffers differently,
    so the library drains the device whenever the mode changes inside a queue.  Full estimator (the host runs ahead),
    sessions of different lengths and LEVELS (a wrong level word shows as a power-of-two gain error), both orders; also
    with the beamformer moved to the tail stream (CSS_TUNE_MVDR_ON_LANES = 0: such passes must not overlap at all)."""
    L, CSS = pkg("_lib"), pkg("css")
    st, desc = mc_state
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=128)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([20.0, 27.0, 34.0, 20.0, 41.0, 23.0]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=500 + k)
            pcm = L.pinned_copy(np.ascontiguousarray(mix[0] * (0.003 if k % 2 else 1.0)))
            sessions.append((pcm, h.run(pcm, run_cfg).copy()))
        for first_pinned in (True, False):
            outs = []
            for k, (pcm, ref) in enumerate(sessions):
                pinned = (k % 2 == 0) == first_pinned
                out = L.pinned_empty(ref.shape, np.float32) if pinned else np.empty(ref.shape, np.float32)
                out[:] = np.nan
                outs.append(h.run_enqueue(pcm, run_cfg, out))
            h.wait()
            for k, (got, (_, ref)) in enumerate(zip(outs, sessions)):
                assert np.array_equal(got, ref), (first_pinned, k, float(np.abs(got - ref).max()))
        h.set_tuning("mvdr_on_lanes", 0)
        outs = [h.run_enqueue(pcm, run_cfg, L.pinned_empty(ref.shape, np.float32)) for pcm, ref in sessions]
        h.wait()
        h.set_tuning("mvdr_on_lanes", 1)
        for k, (got, (_, ref)) in enumerate(zip(outs, sessions)):
            assert np.array_equal(got, ref), ("mvdr on the tail stream", k)
    finally:
        sep.close()


def test_queued_sessions_with_the_host_passes_ahead(mc_state):
    """The full 18-block estimator keeps the device busy for several milliseconds per session while the host enqueues a
    session in two: the host runs passes ahead, so everything a queued pass re-uses (sample buffer half, level word,
    events) must be protected against the passes still in flight.  Different sessions of different lengths, buffers sized
    beforehand (no re-allocation stalls) and not (growing sessions)."""
    L, CSS = pkg("_lib"), pkg("css")
    st, desc = mc_state
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=128)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([20.0, 27.0, 34.0, 41.0, 20.0, 27.0, 34.0, 41.0, 23.0]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=300 + k)
            pcm = L.pinned_copy(np.ascontiguousarray(mix[0] * (0.02 if k % 3 == 1 else 1.0)))   # levels differ too
            sessions.append((pcm, h.run(pcm, run_cfg).copy()))
        for presized in (False, True):
            if presized:
                h.run(sessions[3][0], run_cfg)
            outs = [h.run_enqueue(pcm, run_cfg, L.pinned_empty(ref.shape, np.float32)) for pcm, ref in sessions]
            h.wait()
            for k, (got, (_, ref)) in enumerate(zip(outs, sessions)):
                assert np.array_equal(got, ref), (presized, k, float(np.abs(got - ref).max()))
    finally:
        sep.close()



def test_queued_long_segment_sessions_share_estimator_batches(mc_state):
    """Sessions cut into 10 s segments (624 frames: the any-length kernels) queued back to back share estimator batches
    like any others: each, bit for bit, its own synchronous css_run."""
    L, CSS = pkg("_lib"), pkg("css")
    st, desc = mc_state
    cfg = CSS.make_run_cfg(CSS.CssCfg(show_progressbar=False, activity_th=0.3, segment_size_sec=10.0, hop_size_sec=5.0), 16000, 7)
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=24)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([41.0, 27.0, 60.0, 22.3]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=900 + k)
            pcm = L.pinned_copy(np.ascontiguousarray(mix[0, :mix.shape[1] - 13 * k]))
            sessions.append((pcm, h.run(pcm, cfg).copy()))
        assert int(h.get_plan().segment_frames if hasattr(h.get_plan(), "segment_frames") else 624) == 624
        for rounds in range(2):
            outs = []
            for pcm, ref in sessions:
                out = L.pinned_empty(ref.shape, np.float32)
                out[:] = np.nan
                outs.append(h.run_enqueue(pcm, cfg, out))
            h.wait()
            for k, (got, (_, ref)) in enumerate(zip(outs, sessions)):
                assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 0
                assert np.array_equal(got, ref), (rounds, k, float(np.abs(got - ref).max()))
    finally:
        sep.close()


@pytest.mark.parametrize("group,mode", [(1, "exact_f32"), (2, "exact_f32"), (3, "exact_f32"), (8, "exact_f32"), (3, "split_f16"), (8, "split_f16")])
def test_queued_sessions_share_estimator_batches(mc_state, group, mode):
    """css_run_enqueue merges the segments of consecutive queued sessions into ONE mask-estimator batch (run_group:
    M = 22 k rows per Linear-layer launch for three 60 s meetings): sessions of different lengths, levels and CssCfg
    (threshold, mask floor, stitching loss -- everything but the segmentation may differ inside a group), a session of
    ANOTHER segmentation in the middle (it starts a group of its own), a single left-over session at the end -- each, bit
    for bit, its own synchronous css_run; for every group limit, twice over the same buffers, and with the per-launch
    profile on (one lane)."""
    L, CSS = pkg("_lib"), pkg("css")
    st, desc = mc_state
    mk = lambda **kw: CSS.make_run_cfg(CSS.CssCfg(show_progressbar=False, **kw), 16000, 7)
    cfgs = [mk(activity_th=0.3), mk(activity_th=0.45), mk(activity_th=0.3, mc_mask_floor_db=-6.0), mk(activity_th=0.3, stitching_loss="mse"),
            mk(activity_th=0.3, segment_size_sec=4.0, hop_size_sec=2.0), mk(activity_th=0.3), mk(activity_th=0.5), mk(activity_th=0.3)]
    sep = pkg("separator").HipSeparator(st, None, device=0, max_batch_segments=128, linear_mode=mode)
    try:
        h = sep.handle
        sessions = []
        for k, seconds in enumerate([60.0, 41.0, 20.0, 60.0, 33.0, 27.0, 60.0, 12.3]):
            mix = pkg("synth").synth_meeting(seconds, 7, seed=700 + k)
            pcm = L.pinned_copy(np.ascontiguousarray(mix[0, :mix.shape[1] - 11 * k] * (0.01 if k % 3 == 2 else 1.0)))
            sessions.append((pcm, cfgs[k], h.run(pcm, cfgs[k]).copy()))
        h.set_queue_group(group)
        for profile in (False, True):
            h.set_profile(profile)
            for rounds in range(2):
                outs = []
                for pcm, cfg, ref in sessions:
                    out = L.pinned_empty(ref.shape, np.float32)
                    out[:] = np.nan
                    outs.append(h.run_enqueue(pcm, cfg, out))
                h.wait()
                for k, (got, (_, _, ref)) in enumerate(zip(outs, sessions)):
                    assert np.array_equal(got, ref), (group, profile, rounds, k, float(np.abs(got - ref).max()))
        h.set_profile(False)
        # the handle's session after a queue is the LAST session queued (its plan, its decisions)
        assert h.get_plan().n_samples == sessions[-1][0].shape[0]
        # a bad session is refused when it is queued, not when its group runs; the sessions around it are untouched
        a = h.run_enqueue(sessions[0][0], cfgs[0], L.pinned_empty(sessions[0][2].shape, np.float32))
        with pytest.raises(L.CssError):                                                    # three channels into the 7-channel model
            h.run_enqueue(L.pinned_copy(np.ascontiguousarray(sessions[1][0][:, :3])), cfgs[1], L.pinned_empty(sessions[1][2].shape, np.float32))
        b = h.run_enqueue(sessions[2][0], cfgs[2], L.pinned_empty(sessions[2][2].shape, np.float32))
        h.wait()
        assert np.array_equal(a, sessions[0][2]) and np.array_equal(b, sessions[2][2])
    finally:
        sep.close()


def test_copy_stream_runs_beside_the_main_stream_on_any_handle(model):
    """css_create deals its streams onto hardware queues by measurement (api.hip deal_streams): whatever the process
    created before -- other handles, a torch-owned main stream -- the copy stream must not share the main stream's queue.
    Round 3 measured what happens otherwise: the first 22 MB piece of a sharded upload "took" 14.9 ms because the event
    behind it waited for the 784 MB that followed on the copy stream.  Here: 300 s of audio (134 MB), first piece 32
    segments; the event behind the first piece must fire long before the rest has crossed PCIe (2.5 ms at 53 GB/s)."""
    import torch
    L, CSS, PAR = pkg("_lib"), pkg("css"), pkg("parallel")
    st, desc = model
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    dev = torch.device("cuda", 0)
    n = 300 * 16000
    pcm = L.pinned_copy(np.zeros((n, 7), np.float32))
    plan = L.plan(desc, run_cfg, n)
    me = PAR.make_shard_plan(int(plan.num_segments), int(plan.mix_frames), int(plan.stft_frames), 186, 93, 256, 0, 1)
    groups, cuts = PAR.upload_schedule(me, 186, 93, 512, n)
    seps, worst = [], 0.0
    try:
        for k in range(4):
            ts = torch.cuda.Stream(device=dev) if k % 2 else None
            sep = pkg("separator").HipSeparator(st, None, device=0, **({"stream": int(ts.cuda_stream)} if ts is not None else {}))
            seps.append((sep, ts))
            h = sep.handle
            stream = ts if ts is not None else torch.cuda.ExternalStream(h.stream_ptr(), device=dev)
            be = PAR.HipShardBackend(h, dev, dev, torch_stream=ts) if ts is not None else PAR.HipShardBackend(h, dev, dev)
            for rep in range(3):
                h.sync(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                be.begin(pcm, n, 7, run_cfg, sample_range=(0, n), slice_only=False, cuts=cuts)
                b.record(stream)
                h.sync(); torch.cuda.synchronize()
                if rep:
                    worst = max(worst, a.elapsed_time(b))
            be.close()
        assert worst < 1.6, f"the first piece waited {worst:.2f} ms: the copy stream shares the main stream's hardware queue"
    finally:
        for sep, _ in seps:
            sep.close()

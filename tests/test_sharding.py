"""The segment-sharded multi-GPU path on CPU: partition arithmetic, and the three-exchange driver of
notsofar1-challenge_amd/parallel.py run with world_size 2 and 3 over gloo against an oracle-backed stage
backend (tests/fake_backend.py).  The sharded result must equal the single-rank result BIT FOR BIT and
match the oracle's own driver."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import css_oracle as O
from conftest import ROOT, pkg, rel_rms

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_plans_partition_everything():
    par = pkg("parallel")
    T, hop = 186, 93
    for mix_frames in (187, 250, 624, 3749, 112499):
        nseg = int(np.ceil((mix_frames - (T - hop)) / hop))
        for world in (1, 2, 3, 4, 7, 8):
            plans = par.all_plans(nseg, mix_frames, mix_frames, T, hop, 256, world)
            # owned segments, frames and boundaries tile their ranges exactly once, in rank order
            assert plans[0].own_seg_lo == 0 and plans[-1].own_seg_hi == nseg
            assert plans[0].t_lo == 0 and plans[-1].t_hi == mix_frames
            assert plans[0].b_lo == 0 and plans[-1].b_hi == nseg - 1
            for a, b in zip(plans, plans[1:]):
                assert a.own_seg_hi == b.own_seg_lo and a.t_hi == b.t_lo and a.b_hi == b.b_lo
            for p in plans:
                if p.own_seg_hi == p.own_seg_lo:
                    assert p.num_frames == 0 or p.rank == world - 1
                    continue
                assert p.seg_lo == max(p.own_seg_lo - 1, 0) and p.seg_hi == p.own_seg_hi     # one halo segment
                # every owned frame is covered only by computed segments (bit-exact indexing st = i*hop)
                for t in (p.t_lo, p.t_hi - 1):
                    for seg in range(t // hop - 3, t // hop + 1):
                        if 0 <= seg < nseg and 0 <= t - seg * hop < T:
                            assert p.seg_lo <= seg < p.seg_hi, (mix_frames, world, p, t, seg)
                # both segments of every owned boundary are computed
                for b in range(p.b_lo, p.b_hi):
                    assert p.seg_lo <= b and b + 1 < p.seg_hi
                # the transform covers exactly the frames the computed segments read
                assert p.f_lo == p.seg_lo * hop and p.f_hi == min((p.seg_hi - 1) * hop + T, mix_frames)
            assert sum(p.shard_len for p in plans if p.num_frames) >= (mix_frames + 1) * 256


def test_shard_plan_halo_grows_with_overlap():
    """segment 4 s / hop 2 s: 249-frame segments, hop 124 -> a frame can sit in three segments, two halo segments"""
    par = pkg("parallel")
    T, hop, mix_frames = 249, 124, 1000
    nseg = int(np.ceil((mix_frames - (T - hop)) / hop))
    for world in (2, 3):
        for p in par.all_plans(nseg, mix_frames, mix_frames, T, hop, 256, world):
            if p.own_seg_hi > p.own_seg_lo:
                assert p.seg_lo == max(p.own_seg_lo - 2, 0)
                for t in (p.t_lo, p.t_hi - 1):
                    for seg in range(t // hop - 3, t // hop + 1):
                        if 0 <= seg < nseg and 0 <= t - seg * hop < T:
                            assert p.seg_lo <= seg < p.seg_hi


@pytest.mark.parametrize("T,hop", [(186, 31), (186, 7), (311, 155), (499, 62), (61, 6)])
def test_shard_plan_dense_and_long_segmentations(T, hop):
    """Round 3: any hop with an overlap and segments up to 512 frames (3 s / 0.5 s -> hop = T / 6, 8 s / 1 s, ...): every
    frame a rank owns is covered only by segments the rank computes, every owned boundary has both its segments, and the
    ranks' frame ranges and boundaries tile the meeting."""
    par = pkg("parallel")
    for mix_frames in (T + 1, 3 * T + 5, 2000):
        nseg = int(np.ceil((mix_frames - (T - hop)) / hop))
        for world in (2, 3, 8):
            plans = par.all_plans(nseg, mix_frames, mix_frames, T, hop, 256, world)
            assert plans[0].t_lo == 0 and plans[-1].t_hi == mix_frames and plans[0].b_lo == 0 and plans[-1].b_hi == nseg - 1
            for a, b in zip(plans, plans[1:]):
                assert a.own_seg_hi == b.own_seg_lo and a.t_hi == b.t_lo and a.b_hi == b.b_lo
            halo = -(-T // hop) - 1
            for p in plans:
                if p.own_seg_hi == p.own_seg_lo:
                    continue
                assert p.seg_lo == max(p.own_seg_lo - halo, 0) and p.seg_hi == p.own_seg_hi
                for t in range(p.t_lo, p.t_hi, max((p.t_hi - p.t_lo) // 50, 1)):
                    for seg in range(max(t // hop - halo - 1, 0), t // hop + 1):
                        if seg < nseg and 0 <= t - seg * hop < T:
                            assert p.seg_lo <= seg < p.seg_hi, (mix_frames, world, p, t, seg)
                for b in range(p.b_lo, p.b_hi):
                    assert p.seg_lo <= b and b + 1 < p.seg_hi


def _small_model():
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=1)
    return w.apply_golden_recipe(w.portable_state_dict(desc, 11)), desc


def _worker(rank, world, port, out_dir, n_samples):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, HERE)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_backend import OracleStageBackend
        par = pkg("parallel")
        st, desc = _small_model()
        mix = pkg("synth").synth_meeting(10.0, 7, seed=4)[:, :n_samples]
        be = OracleStageBackend(O.ConformerParams(st), O.OracleCssCfg(activity_th=0.3))
        be.begin(mix[0], mix.shape[1], 7)
        out = par.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist)
        np.save(os.path.join(out_dir, f"w{world}_r{rank}.npy"), out.numpy())
        # the same session again, every rank finishing only its own sample range (one 256-sample block per seam exchanged)
        be.begin(mix[0], mix.shape[1], 7)
        own, (lo, hi) = par.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist, gather="range")
        np.save(os.path.join(out_dir, f"w{world}_r{rank}_own.npy"), own.numpy())
        np.save(os.path.join(out_dir, f"w{world}_r{rank}_range.npy"), np.array([lo, hi]))
        np.save(os.path.join(out_dir, f"w{world}_r{rank}_nseg.npy"), np.array([be.calls["masknet_segments"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 9 * 16000 + 123), (3, 9 * 16000 + 123), (3, 4 * 16000 + 50)])
def test_sharded_driver_over_gloo_matches_single_rank(tmp_path, world, n):
    """(7 segments with a ragged tail over 2 and 3 ranks; 2 segments over 3 ranks: one rank owns nothing and still takes part in
    every all-gather)"""
    from fake_backend import OracleStageBackend
    par = pkg("parallel")
    st, desc = _small_model()
    params = O.ConformerParams(st)
    ocfg = O.OracleCssCfg(activity_th=0.3)
    mix = pkg("synth").synth_meeting(10.0, 7, seed=4)[:, :n]
    # single rank through the same driver
    be = OracleStageBackend(params, ocfg)
    be.begin(mix[0], n, 7)
    single = par.sharded_separate_and_stitch(be, 3, 186, 93, 256, 0, 1, None).numpy()
    # ... equals the oracle's own (reference-shaped) driver up to float32 rounding of the summation order
    ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg)
    for k in range(3):
        assert rel_rms(single[k], ow[k]) < 1e-6
    assert [tuple(p) for p in be.perms] == [tuple(p) for p in oside["perms"]]

    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), n), nprocs=world, join=True)
    total_segments = 0
    for r in range(world):
        got = np.load(tmp_path / f"w{world}_r{r}.npy")
        assert got.shape == single.shape
        assert np.array_equal(got, single), f"rank {r}: sharded result differs from the single-rank result"
        total_segments += int(np.load(tmp_path / f"w{world}_r{r}_nseg.npy")[0])
    nseg = oside["plan"].num_segments
    busy = sum(1 for r in range(world) if (r + 1) * nseg // world > r * nseg // world)      # ranks that own a segment
    assert total_segments == 2 * (nseg + (busy - 1))   # exactly one halo segment per seam between ranks that own segments (two runs)
    # gather="range": the ranks' own ranges tile the output and are the single-rank samples, bit for bit
    edge = 0
    for r in range(world):
        lo, hi = (int(v) for v in np.load(tmp_path / f"w{world}_r{r}_range.npy"))
        assert lo == edge and hi >= lo
        own = np.load(tmp_path / f"w{world}_r{r}_own.npy")
        assert own.shape == (3, hi - lo) and np.array_equal(own, single[:, lo:hi]), f"rank {r}: own range differs"
        edge = hi
    assert edge == single.shape[1]


def _range_worker(rank, world, port, out_dir, n_samples):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, HERE)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_backend import OracleStageBackend
        par, lib = pkg("parallel"), pkg("_lib")
        st, desc = _small_model()
        mix = pkg("synth").synth_meeting(10.0, 7, seed=4)[:, :n_samples]

        class Overflowing(OracleStageBackend):
            """rank 1 alone reports a split-f16 range overflow, as css_check_range would"""
            def check_range(self):
                if rank == 1:
                    raise lib.CssError(lib.CSS_ERR_RANGE, "range")

        be = Overflowing(O.ConformerParams(st), O.OracleCssCfg(activity_th=0.3))
        verdicts = []
        for _ in range(2):          # twice: a rank that raised alone would have left the collective sequence
            be.begin(mix[0], mix.shape[1], 7)
            try:
                par.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist, gather="range")
                verdicts.append(0)
            except lib.CssError as e:
                verdicts.append(e.code)
        np.save(os.path.join(out_dir, f"verdict_r{rank}.npy"), np.array(verdicts))
    finally:
        dist.destroy_process_group()


def test_range_verdict_is_collective(tmp_path):
    """with world > 1 a rank whose segments left the split-f16 range has already sent NaN costs to the others;
    every rank must raise CSS_ERR_RANGE, and none may drop out of the collective sequence (the next session still runs)."""
    lib = pkg("_lib")
    world, n = 2, 5 * 16000
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_range_worker, args=(world, port, str(tmp_path), n), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"verdict_r{r}.npy").tolist() == [lib.CSS_ERR_RANGE, lib.CSS_ERR_RANGE], r


def test_upload_schedule_covers_the_ranks_segments_and_samples():
    """parallel.upload_schedule: the groups tile [seg_lo, seg_hi) in order, every group's last sample is inside the piece
    that ends at its cut, the cuts ascend inside the rank's sample range, no crumbs at the end."""
    par = pkg("parallel")
    T, hop, fhop, flen = 186, 93, 256, 512
    for nseg, world in ((1209, 8), (1209, 2), (40, 3), (7, 2), (300, 5)):
        mix_frames = (nseg - 1) * hop + T - 11
        n = (mix_frames - 1) * fhop + flen
        for r in range(world):
            me = par.make_shard_plan(nseg, mix_frames, mix_frames, T, hop, fhop, r, world)
            for first, growth in ((32, 8), (3, 2), (1, 1)):
                groups, cuts = par.upload_schedule(me, T, hop, flen, n, first=first, growth=growth)
                if me.seg_hi == me.seg_lo:
                    assert groups == [] and cuts == []
                    continue
                assert groups[0][0] == me.seg_lo and groups[-1][1] == me.seg_hi
                assert all(a[1] == b[0] and a[1] > a[0] for a, b in zip(groups, groups[1:])) and groups[-1][1] > groups[-1][0]
                assert len(cuts) == len(groups) - 1 and cuts == sorted(cuts)
                lo, hi = me.pcm_range(flen, n)
                for (a, b), c in zip(groups, cuts):
                    assert lo < c <= hi
                    assert min(((b - 1) * hop + T - 1) * fhop + flen, n) <= c      # the group's frames read samples below the cut
                assert groups[-1][1] - groups[-1][0] >= min(first // 2, me.seg_hi - me.seg_lo)

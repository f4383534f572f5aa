"""Segment-sharded CSS of one long meeting over the GPUs of a node (one process per GPU, RCCL over xGMI).

The reference processes a meeting's sliding-window segments one after another on one device
(css/css.py:182-250) and has no multi-GPU inference path; its segments are independent until the
stitching stage, which only looks at adjacent segments.  We partition by OUTPUT FRAME RANGE with
global segment indices (segment i always covers frames [i*hop, i*hop + T), css.py:183), so results do
not depend on the number of ranks:

  rank r owns segments [S_r, S_r+1) with S_r = floor(r * num_segments / world) and the frames
  [S_r*hop, S_r+1*hop) (first rank from 0, last rank to T_long).  It computes its own segments plus the
  ceil(T/hop) - 1 halo segments before S_r that still cover the rank's first frames (ONE segment with the
  shipped 3 s / 1.5 s configuration).

Exchanges (all tiny except the last; `torch.distributed`, backend "nccl" = RCCL on ROCm, "gloo" in CPU tests):
  1. raw 3x3 PIT cost matrices of the boundaries each rank owns (all-gather, 72 B per boundary); every
     rank then replays the sequential permutation scan of css.py:266-285 identically;
  2. thresholded activity bits of the frames each rank owns (all-gather, 3 B per frame), because the
     dilate/erode gate (css.py:305-308) looks 36 frames to either side;
  3. the separated-waveform shards (all-gather, 3 x 4 B per sample): each rank inverse-transforms only
     its own frames, so adjacent shards overlap by one hop (256 samples) where the two-frame
     overlap-add crosses the rank boundary; the stitch adds the two partial blocks (a two-term
     float sum commutes, so the result is bit-identical to the single-GPU run).
"""
from __future__ import annotations

import dataclasses
from typing import List

import numpy as np


@dataclasses.dataclass
class ShardPlan:
    rank: int
    world: int
    own_seg_lo: int
    own_seg_hi: int
    seg_lo: int      # first segment computed (own_seg_lo - 1 halo, clamped)
    seg_hi: int
    t_lo: int        # frames owned
    t_hi: int
    f_lo: int        # STFT frames the computed segments read
    f_hi: int
    b_lo: int        # PIT boundaries owned (boundary b joins segments b, b+1; owner = owner of b+1)
    b_hi: int
    hop_samples: int

    @property
    def num_frames(self) -> int:
        return self.t_hi - self.t_lo

    @property
    def shard_len(self) -> int:
        """samples of the partial inverse transform: output blocks t_lo .. t_hi inclusive"""
        return (self.t_hi - self.t_lo + 1) * self.hop_samples

    @property
    def sample_lo(self) -> int:
        return self.t_lo * self.hop_samples


def make_shard_plan(num_segments: int, mix_frames: int, stft_frames: int, seg_frames: int, hop_frames: int,
                    hop_samples: int, rank: int, world: int) -> ShardPlan:
    s_lo = rank * num_segments // world
    s_hi = (rank + 1) * num_segments // world
    t_lo = 0 if rank == 0 else s_lo * hop_frames
    t_hi = mix_frames if rank == world - 1 else s_hi * hop_frames
    if s_hi == s_lo and rank != world - 1:   # a rank without segments owns no frames
        t_lo = t_hi = s_lo * hop_frames if rank else 0
    halo = -(-seg_frames // hop_frames) - 1   # earlier segments that still cover the first owned frame (1 at 3 s / 1.5 s)
    seg_lo = max(s_lo - max(halo, 1), 0) if s_hi > s_lo else s_lo
    f_lo = seg_lo * hop_frames
    f_hi = min((s_hi - 1) * hop_frames + seg_frames, stft_frames) if s_hi > s_lo else f_lo
    f_hi = max(f_hi, f_lo)
    b_lo = max(s_lo - 1, 0)
    b_hi = max(min(s_hi - 1, num_segments - 1), b_lo) if s_hi > s_lo else b_lo
    return ShardPlan(rank, world, s_lo, s_hi, seg_lo, s_hi, t_lo, t_hi, f_lo, f_hi, b_lo, b_hi, hop_samples)


def all_plans(num_segments, mix_frames, stft_frames, seg_frames, hop_frames, hop_samples, world) -> List[ShardPlan]:
    return [make_shard_plan(num_segments, mix_frames, stft_frames, seg_frames, hop_frames, hop_samples, r, world)
            for r in range(world)]


class HipShardBackend:
    """Stage calls of one rank on its GPU (the C ABI's css_stage_* entry points)."""

    def __init__(self, handle, torch_device, comm_device=None):
        """`comm_device`: where the process group exchanges tensors -- the GPU itself for "nccl" (RCCL over
        xGMI, the default), torch.device("cpu") for "gloo" (functional testing of the multi-process path)."""
        import torch
        self.h = handle
        self.dev = torch_device
        self.comm_dev = comm_device if comm_device is not None else torch_device
        self.torch = torch

    def begin(self, pcm, n, c, run_cfg):
        if hasattr(pcm, "data_ptr"):
            self._pcm_keep = pcm
            self.h.begin(pcm.data_ptr(), n, c, run_cfg, device=True)
        else:
            self.h.begin(pcm, n, c, run_cfg, device=False)

    def plan(self):
        return self.h.get_plan()

    def stft_range(self, lo, hi): self.h.stage_stft_range(lo, hi)
    def masknet(self, lo, hi): self.h.stage_masknet(lo, hi)
    def mvdr(self, lo, hi): self.h.stage_mvdr(lo, hi)
    def pit_costs(self, lo, hi): self.h.stage_pit_costs(lo, hi)
    def stitch_masks(self, lo, hi): self.h.stage_stitch_masks(lo, hi)
    def stitch_gate(self, lo, hi): self.h.stage_stitch_gate(lo, hi)

    def read_costs(self):
        from . import _lib
        return self.h.read(_lib.BUF_PIT_COST)

    def write_perms(self, perms):
        from . import _lib
        self.h.write(_lib.BUF_PERMS, perms)

    def read_act_b(self):
        from . import _lib
        return self.h.read(_lib.BUF_ACT_B)

    def write_act_b(self, act):
        from . import _lib
        self.h.write(_lib.BUF_ACT_B, act)

    def istft_partial(self, lo, hi, num_spks, shard_len):
        out = self.torch.zeros((num_spks, shard_len), dtype=self.torch.float32, device=self.dev)
        self.torch.cuda.synchronize(self.dev)  # zeros are written on torch's stream, the kernels on the handle's
        self.h.stage_istft_partial(lo, hi, out.data_ptr(), shard_len)
        self.h.sync()
        return out.to(self.comm_dev)

    def pit_scan(self, costs, num_spks):
        from . import _lib
        return _lib.pit_scan(costs, num_spks)

    def to_comm(self, arr):
        """numpy -> tensor on the device the process group communicates from"""
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.comm_dev)


def _all_gather(dist, tensor, world):
    import torch
    outs = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(outs, tensor)
    return outs


class ShardedSession:
    """One rank's share of a session that `backend.begin(...)` has opened, as three phases separated by
    the three exchanges.  `sharded_separate_and_stitch` drives them over torch.distributed; tests drive the
    same phases for several virtual ranks in one process."""

    def __init__(self, backend, num_spks: int, seg_frames: int, hop_frames: int, hop_samples: int, rank: int,
                 world: int):
        self.be, self.S, self.rank, self.world = backend, num_spks, rank, world
        plan = backend.plan()
        self.nseg, self.TL, self.n_out = int(plan.num_segments), int(plan.mix_frames), int(plan.n_out)
        self.plans = all_plans(self.nseg, self.TL, int(plan.stft_frames), seg_frames, hop_frames, hop_samples, world)
        self.me = self.plans[rank]

    # phase 1: everything per segment, then the raw PIT costs of the owned boundaries
    def segments_and_costs(self) -> np.ndarray:
        me, be = self.me, self.be
        be.stft_range(me.f_lo, me.f_hi)
        be.masknet(me.seg_lo, me.seg_hi)
        be.mvdr(me.seg_lo, me.seg_hi)
        be.pit_costs(me.b_lo, me.b_hi)
        costs = np.asarray(be.read_costs(), dtype=np.float64).reshape(-1, self.S * self.S)
        return costs[me.b_lo:me.b_hi].copy()

    @staticmethod
    def join_costs(plans, pieces) -> np.ndarray:
        return np.concatenate([np.asarray(pieces[r])[:plans[r].b_hi - plans[r].b_lo] for r in range(len(plans))], axis=0)

    # phase 2: identical permutation scan on every rank, overlap-add of the masks, activity bits
    def masks_and_activity(self, all_costs: np.ndarray) -> np.ndarray:
        assert all_costs.shape[0] == max(self.nseg - 1, 0), (all_costs.shape, self.nseg)
        self.be.write_perms(self.be.pit_scan(all_costs, self.S))
        self.be.stitch_masks(self.me.t_lo, self.me.t_hi)
        act = np.asarray(self.be.read_act_b(), dtype=np.uint8)  # [S, T_long]; only the owned columns are valid
        return act[:, self.me.t_lo:self.me.t_hi].copy()

    @staticmethod
    def join_activity(plans, pieces) -> np.ndarray:
        return np.concatenate([np.asarray(pieces[r])[:, :plans[r].num_frames] for r in range(len(plans))], axis=1)

    # phase 3: gate + partial inverse transform of the owned frames
    def gate_and_istft(self, all_act: np.ndarray):
        assert all_act.shape == (self.S, self.TL), all_act.shape
        self.be.write_act_b(all_act)
        self.be.stitch_gate(self.me.t_lo, self.me.t_hi)
        return self.be.istft_partial(self.me.t_lo, self.me.t_hi, self.S, self.me.shard_len)

    @staticmethod
    def join_shards(plans, shards, num_spks: int, n_out: int):
        """Place every rank's shard at its sample offset; the one-hop overlaps at the seams add up."""
        import torch
        out = torch.zeros((num_spks, n_out), dtype=shards[0].dtype, device=shards[0].device)
        for r, p in enumerate(plans):
            if p.num_frames == 0:
                continue
            cut = min(p.shard_len, n_out - p.sample_lo)
            out[:, p.sample_lo:p.sample_lo + cut] += shards[r][:, :cut]
        return out


def sharded_separate_and_stitch(backend, num_spks: int, seg_frames: int, hop_frames: int, hop_samples: int,
                                rank: int, world: int, dist=None):
    """Runs one rank's share of a session that `backend.begin(...)` has opened and returns the full
    separated waveforms [S, n_out] (a tensor on the backend's communication device), identical on every
    rank and identical to the single-rank result."""
    import torch
    ss = ShardedSession(backend, num_spks, seg_frames, hop_frames, hop_samples, rank, world)
    plans, me = ss.plans, ss.me
    S2 = num_spks * num_spks

    mine = ss.segments_and_costs()
    if world > 1:
        send = np.zeros((max(max(p.b_hi - p.b_lo for p in plans), 1), S2), dtype=np.float64)
        send[:mine.shape[0]] = mine
        got = _all_gather(dist, backend.to_comm(send), world)
        all_costs = ss.join_costs(plans, [g.cpu().numpy() for g in got])
    else:
        all_costs = mine

    act = ss.masks_and_activity(all_costs)
    if world > 1:
        send = np.zeros((num_spks, max(max(p.num_frames for p in plans), 1)), dtype=np.uint8)
        send[:, :act.shape[1]] = act
        got = _all_gather(dist, backend.to_comm(send), world)
        all_act = ss.join_activity(plans, [g.cpu().numpy() for g in got])
    else:
        all_act = act

    shard = ss.gate_and_istft(all_act)
    if world == 1:
        return shard[:, :ss.n_out]
    send = torch.zeros((num_spks, max(p.shard_len for p in plans)), dtype=shard.dtype, device=shard.device)
    send[:, :me.shard_len] = shard
    got = _all_gather(dist, send, world)
    return ss.join_shards(plans, got, num_spks, ss.n_out)

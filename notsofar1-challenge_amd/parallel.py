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

Exchanges (all tiny; `torch.distributed`, backend "nccl" = RCCL on ROCm, "gloo" in CPU tests):
  1. raw 3x3 PIT cost matrices of the boundaries each rank owns (all-gather, 72 B per boundary); every
     rank then replays the sequential permutation scan of css.py:266-285 identically, on its GPU;
  2. thresholded activity bits of the frames each rank owns (all-gather, 3 B per frame), because the
     dilate/erode gate (css.py:305-308) looks 36 frames to either side;
  3. the waveform seam: each rank inverse-transforms only its own frames, so its shard ends with one output block
     (one hop, 256 samples) that belongs to the NEXT rank's range -- the second half of its last frame.  Those blocks
     are exchanged (all-gather, 3 x 256 x 4 B per rank) and each rank adds its left neighbour's onto the head of its
     shard (a two-term float sum commutes, so the result is bit-identical to the single-GPU run).  A rank then holds the
     finished samples of its own range (`gather="range"`: what a serving process writes out); `gather="all"`
     all-gathers the whole shards instead (3 x 4 B per sample) and every rank assembles the full waveforms, as
     css.py:110 returns them.

Frame geometries other than frame_len = 2 hop (ExtractorCfg.frame_len / frame_hop, round 6): ovl = ceil(frame_len / hop)
frames overlap on an output sample and the overlap-add is an ORDERED float sum (oldest frame first), so partial blocks of
two ranks no longer compose bit for bit.  Exchange 3 then carries the synthesis ROWS of a rank's last ovl - 1 frames
(frame_len floats per frame and stream) and the right neighbour runs the single-GPU overlap-add over them; `gather="all"`
all-gathers the FINISHED ranges and places them (no sums).  Same bits as the single-GPU pass for any world.

Nothing of this touches the host between the upload of a rank's samples and the download of the result: the
pieces are zero-copy torch views of the handle's own device buffers (costs, activity bits), the collectives and
the few packing / unpacking copies are enqueued on the handle's HIP stream (wrapped as a torch ExternalStream),
and the permutation scan runs on the device.  The host only enqueues.
"""
from __future__ import annotations

import contextlib
import dataclasses
from typing import List

import numpy as np

from . import _lib


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

    def pcm_range(self, frame_len: int, n_samples: int):
        """samples of the recording the frames [f_lo, f_hi) read: what this rank uploads"""
        if self.f_hi <= self.f_lo:
            return 0, 0
        return self.f_lo * self.hop_samples, min((self.f_hi - 1) * self.hop_samples + frame_len, n_samples)


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


def upload_schedule(me: ShardPlan, seg_frames: int, hop_frames: int, frame_len: int, n_samples: int, first: int = 32,
                    growth: int = 8):
    """Groups of a rank's segments whose samples cross PCIe piece by piece, so that the upload of all but the first
    group hides under the stages of the groups before it: `first` segments, then `growth` times as many, ... (a segment
    costs ~10 times as long to compute as to upload).  Returns (segment_groups [(lo, hi), ...], cuts): `cuts` are the
    sample positions where the pieces end (HipShardBackend.begin) and `segment_groups` goes to ShardedSession."""
    groups, cuts, lo, size = [], [], me.seg_lo, max(int(first), 1)
    while lo < me.seg_hi:
        hi = min(lo + size, me.seg_hi)
        if me.seg_hi - hi < first // 2:   # no crumbs at the end
            hi = me.seg_hi
        groups.append((lo, hi))
        if hi < me.seg_hi:
            cuts.append(min(((hi - 1) * hop_frames + seg_frames - 1) * me.hop_samples + frame_len, n_samples))
        lo, size = hi, size * growth
    return groups, cuts


class _DeviceArray:
    """A device buffer of the handle as something torch.as_tensor can wrap without copying."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class HipShardBackend:
    """Stage calls of one rank on its GPU (the C ABI's css_stage_* entry points), plus torch views of the two device
    buffers the exchanges read and write.  Everything is enqueued on the handle's own HIP stream."""

    def __init__(self, handle, torch_device, comm_device=None, torch_stream=None, cabi_comm=False):
        """`cabi_comm`: the all-gathers go through the C ABI (css_comm_all_gather on the handle's communicator, which the
        caller has initialised with handle.comm_init) instead of torch.distributed -- the route of a non-Python host.
        `comm_device`: where the process group exchanges tensors -- the GPU itself for "nccl" (RCCL over
        xGMI, the default), torch.device("cpu") for "gloo" (functional testing of the multi-process path).
        `torch_stream`: the torch.cuda.Stream the handle was created on (css_create's `stream` argument) -- the
        preferred arrangement for long-lived processes: torch owns the stream, so its caching allocators (device and
        page-locked host memory remember the streams they were used on) never outlive it.  Default: the handle's own
        stream, wrapped; then call close() before the handle is destroyed."""
        import torch
        self.h = handle
        self.dev = torch_device
        self.comm_dev = comm_device if comm_device is not None else torch_device
        self.torch = torch
        if torch_stream is not None:
            assert int(torch_stream.cuda_stream) == handle.stream_ptr(), "the handle was not created on this stream"
            self.stream = torch_stream
        else:
            self.stream = torch.cuda.ExternalStream(handle.stream_ptr(), device=torch_device)
        self._scratch = {}
        self._keep = None
        self.cabi_comm = handle if cabi_comm else None

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def close(self):
        """Release the work tensors BEFORE the handle is destroyed: they were allocated while the handle's stream was
        current, and torch's allocator would otherwise touch that stream after css_destroy has destroyed it."""
        self._scratch.clear()
        self._keep = None
        self.torch.cuda.synchronize(self.dev)
        self.torch.cuda.empty_cache()

    def begin(self, pcm, n, c, run_cfg, sample_range=None, slice_only=False, cuts=None):
        """pcm: a device tensor [n, c] (resident input), or a float32 numpy array in host memory, of which only
        `sample_range` (default: everything) is uploaded; slice_only: the array holds just that range.  `cuts`: sample
        positions inside the range at which the upload is cut into pieces (upload_schedule): the first piece is uploaded
        on the handle's stream, the others follow on its copy stream while the stages of the earlier pieces run."""
        self._keep = pcm
        if hasattr(pcm, "data_ptr"):
            self.h.begin(pcm.data_ptr(), n, c, run_cfg, device=True)
        elif sample_range is not None:
            lo, hi = sample_range
            edges = [lo] + [int(x) for x in (cuts or []) if lo < x < hi] + [hi]
            self.h.begin_range(pcm, n, c, run_cfg, edges[0], edges[1], slice_only, base_sample=lo)
            for a, b in zip(edges[1:-1], edges[2:]):
                self.h.upload_range(pcm, c, a, b, base_sample=lo if slice_only else 0)
        else:
            self.h.begin(pcm, n, c, run_cfg, device=False)

    def plan(self):
        return self.h.get_plan()

    def stft_range(self, lo, hi): self.h.stage_stft_range(lo, hi)
    def masknet(self, lo, hi): self.h.stage_masknet(lo, hi)
    def mvdr(self, lo, hi): self.h.stage_mvdr(lo, hi)
    def pit_costs(self, lo, hi): self.h.stage_pit_costs(lo, hi)
    def pit_scan(self): self.h.stage_pit_scan()
    def stitch_masks(self, lo, hi): self.h.stage_stitch_masks(lo, hi)
    def stitch_gate(self, lo, hi): self.h.stage_stitch_gate(lo, hi)

    def _view(self, which, typestr):
        dims, _ = self.h.buffer_dims(which)
        return self.torch.as_tensor(_DeviceArray(self.h.devptr(which), dims, typestr), device=self.dev)

    def costs_view(self):
        """raw PIT costs [num_segments - 1, S*S] float64, in place on the device"""
        from . import _lib
        return self._view(_lib.BUF_PIT_COST, "<f8")

    def act_view(self):
        """thresholded activity [S, T_long] uint8, in place on the device"""
        from . import _lib
        return self._view(_lib.BUF_ACT_B, "|u1")

    def level_view(self):
        """max |sample| of the PCM this rank has laid out [1] float32, in place on the device"""
        from . import _lib
        return self._view(_lib.BUF_LEVEL, "<f4")

    def istft_partial(self, lo, hi, out):
        self.h.stage_istft_partial(lo, hi, out.data_ptr(), out.stride(0))

    # the seam of the general frame geometries (css_stage_synthesis / css_stage_seam_rows / css_stage_overlap_add)
    def synthesis(self, lo, hi):
        self.h.stage_synthesis(lo, hi)

    def seam_rows(self, lo, hi, rows, write):
        assert rows.is_contiguous() and tuple(rows.shape[1:2]) == (hi - lo,)
        self.h.stage_seam_rows(lo, hi, rows.data_ptr(), write)

    def overlap_add(self, f_lo, f_hi, q_lo, q_hi, out, out_q0):
        assert out.stride(1) == 1
        self.h.stage_overlap_add(f_lo, f_hi, q_lo, q_hi, out.data_ptr(), out.stride(0), out_q0)

    def join_shards(self, all_shards, plans, out):
        """gathered [world, S, ld] -> out [S, n_out] in one launch (css_stage_join_shards)"""
        assert all_shards.is_contiguous() and out.stride(1) == 1
        self.h.stage_join_shards(all_shards.data_ptr(), len(plans), all_shards.shape[2], [p.t_lo for p in plans],
                                 [p.t_hi for p in plans], out.data_ptr(), out.stride(0))

    def check_range(self):
        """Host-synchronising test of the split-f16 range flag after a staged session (css_check_range): raises
        CssError(CSS_ERR_RANGE) when a Linear-layer operand overflowed, i.e. the session's waveforms are not valid."""
        self.h.check_range()

    def scratch(self, name, shape, dtype):
        """persistent work tensors (send / receive pieces, index maps): allocated once per shape"""
        key = (name, tuple(shape), dtype)
        t = self._scratch.get(key)
        if t is None:
            t = self._scratch[key] = self.torch.empty(tuple(shape), dtype=dtype, device=self.dev)
        return t

    def index_tensor(self, key, make):
        """index maps of the joins: built (on the host, by `make()`) and uploaded once per session shape"""
        t = self._scratch.get(key)
        if t is None:
            t = self._scratch[key] = self.torch.from_numpy(make()).to(self.dev)
        return t


class ShardedSession:
    """One rank's share of a session that `backend.begin(...)` has opened, as three phases separated by the three
    exchanges.  A phase returns the piece this rank contributes (padded to the size of the largest rank's, so that the
    all-gather is regular); the next phase takes the gathered pieces `[world, ...]`.  `sharded_separate_and_stitch`
    chains them over torch.distributed; tests chain them for several virtual ranks in one process."""

    def __init__(self, backend, num_spks: int, seg_frames: int, hop_frames: int, hop_samples: int, rank: int,
                 world: int, segment_groups=None, frame_len=None):
        import torch
        self.torch = torch
        self.be, self.S, self.rank, self.world = backend, num_spks, rank, world
        self.seg_frames, self.hop_frames, self.segment_groups = seg_frames, hop_frames, segment_groups
        plan = backend.plan()
        self.nseg, self.TL, self.n_out = int(plan.num_segments), int(plan.mix_frames), int(plan.n_out)
        self.plans = all_plans(self.nseg, self.TL, int(plan.stft_frames), seg_frames, hop_frames, hop_samples, world)
        self.me = self.plans[rank]
        self.hop_samples = hop_samples
        self.max_b = max(max(p.b_hi - p.b_lo for p in self.plans), 1)
        self.max_t = max(max(p.num_frames for p in self.plans), 1)
        self.max_len = max(p.shard_len for p in self.plans)
        # frame geometry: frame_len = 2 hop (every shipped model) exchanges partial blocks; anything else synthesis rows
        self.frame_len = int(frame_len) if frame_len else 2 * hop_samples
        self.ovl = -(-self.frame_len // hop_samples)          # frames over an output sample
        self.general = self.frame_len != 2 * hop_samples
        self.K = self.ovl - 1                                  # frames of the left neighbour(s) a rank's first blocks read
        if self.general:
            self.max_len = max((p.num_frames + self.K) * hop_samples for p in self.plans)

    def _ctx(self):
        return self.be.on_stream() if hasattr(self.be, "on_stream") else contextlib.nullcontext()

    def _gather_index(self, name, lo_hi, width, total):
        """position of element e (owned by the rank whose [lo, hi) holds it) inside the gathered [world, width] pieces"""
        def make():
            idx = np.zeros(total, dtype=np.int64)
            for r, (lo, hi) in enumerate(lo_hi):
                idx[lo:hi] = r * width + np.arange(hi - lo)
            return idx
        return self.be.index_tensor((name, self.world, self.nseg, self.TL, width, total), make)

    # phase 1: everything per segment, then the raw PIT costs of the owned boundaries
    def segments_and_costs(self):
        me, be, torch = self.me, self.be, self.torch
        with self._ctx():
            # group by group (upload_schedule): a group's frames are transformed as soon as ITS samples have landed
            groups = self.segment_groups or [(me.seg_lo, me.seg_hi)]
            assert groups[0][0] == me.seg_lo and groups[-1][1] == me.seg_hi and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
            f_prev = me.f_lo
            for a, b in groups:
                f_hi = me.f_hi if b == me.seg_hi else min((b - 1) * self.hop_frames + self.seg_frames, me.f_hi)
                if f_hi > f_prev:
                    be.stft_range(f_prev, f_hi)
                    f_prev = f_hi
                if b > a:
                    be.masknet(a, b)
                    be.mvdr(a, b)
            be.pit_costs(me.b_lo, me.b_hi)
            if self.world == 1:
                return None
            # one more row rides along: the peak sample of this rank's slice (every rank must scale the split-f16 operand
            # of the synthesis transform by the same power of two, or the last bits would depend on the sharding)
            send = be.scratch("send_costs", (self.max_b + 1, self.S * self.S), torch.float64)
            if me.b_hi > me.b_lo:
                send[:me.b_hi - me.b_lo].copy_(be.costs_view()[me.b_lo:me.b_hi])
            send[self.max_b, 0] = be.level_view()[0]
            return send

    # phase 2: identical permutation scan on every rank, overlap-add of the masks, activity bits
    def masks_and_activity(self, all_costs):
        me, be, torch = self.me, self.be, self.torch
        with self._ctx():
            if self.world > 1:
                assert tuple(all_costs.shape) == (self.world, self.max_b + 1, self.S * self.S), all_costs.shape
                if self.nseg > 1:
                    idx = self._gather_index("idx_costs", [(p.b_lo, p.b_hi) for p in self.plans], self.max_b + 1, self.nseg - 1)
                    be.costs_view()[:self.nseg - 1].copy_(all_costs.reshape(-1, self.S * self.S).index_select(0, idx))
                be.level_view().copy_(all_costs[:, self.max_b, 0].max().to(torch.float32).reshape(1))
            be.pit_scan()
            be.stitch_masks(me.t_lo, me.t_hi)
            if self.world == 1:
                return None
            send = be.scratch("send_act", (self.S, self.max_t), torch.uint8)
            if me.num_frames:
                send[:, :me.num_frames].copy_(be.act_view()[:, me.t_lo:me.t_hi])
            return send

    # phase 3: gate + partial inverse transform of the owned frames
    def gate_and_istft(self, all_act):
        me, be, torch = self.me, self.be, self.torch
        with self._ctx():
            if self.world > 1:
                assert tuple(all_act.shape) == (self.world, self.S, self.max_t), all_act.shape
                idx = self._gather_index("idx_act", [(p.t_lo, p.t_hi) for p in self.plans], self.max_t, self.TL)
                be.act_view().copy_(all_act.permute(1, 0, 2).reshape(self.S, -1).index_select(1, idx))
            be.stitch_gate(me.t_lo, me.t_hi)
            if self.general:
                # the synthesis rows of the owned frames stay in the handle; the last K of them are what the right neighbour
                # needs (right-aligned in the piece: a rank with fewer than K frames passes on what it has)
                be.synthesis(me.t_lo, me.t_hi)
                send = be.scratch("send_rows", (self.S, max(self.K, 1), self.frame_len), torch.float32)
                send.zero_()
                cnt = min(self.K, me.num_frames)
                if cnt:
                    tmp = be.scratch(("rows", cnt), (self.S, cnt, self.frame_len), torch.float32)
                    be.seam_rows(me.t_hi - cnt, me.t_hi, tmp, False)
                    send[:, self.K - cnt:, :].copy_(tmp)
                return send
            shard = be.scratch("send_wav", (self.S, self.max_len), torch.float32)
            be.istft_partial(me.t_lo, me.t_hi, shard)
            return shard

    def own_range(self):
        """samples [lo, hi) of the output this rank finishes (the last rank also owns the closing half frame)"""
        me = self.me
        if me.num_frames == 0:
            return me.sample_lo, me.sample_lo
        return me.sample_lo, (self.n_out if me.t_hi == self.TL else me.t_hi * self.hop_samples)

    def seam_piece(self, shard):
        """The block past this rank's own range -- its last frame's second half, which the next rank adds [S, hop]."""
        torch, hop, me = self.torch, self.hop_samples, self.me
        if self.general:
            return shard      # (gate_and_istft returned the rows piece itself)
        with self._ctx():
            send = self.be.scratch("send_seam", (self.S, hop), torch.float32)
            if me.num_frames:
                send.copy_(shard[:, me.num_frames * hop:(me.num_frames + 1) * hop])
            else:
                send.zero_()
            return send

    def finish_range(self, shard, all_seams):
        """Adds the left neighbour's seam block onto the head of this rank's shard; returns the finished samples of
        own_range() as a view of the shard [S, hi - lo]."""
        hop, me = self.hop_samples, self.me
        lo, hi = self.own_range()
        if self.general:
            return self._finish_range_general(all_seams)
        with self._ctx():
            left = [p for p in self.plans[:self.rank] if p.num_frames > 0]
            if me.num_frames and left:
                assert tuple(all_seams.shape) == (self.world, self.S, hop), all_seams.shape
                shard[:, :hop].add_(all_seams[left[-1].rank])
            return shard[:, :hi - lo]

    def _finish_range_general(self, all_rows):
        """General frame geometry: the left neighbours' last K synthesis rows [world, S, K, frame_len] take their place in the
        handle's row buffer, then the single-GPU overlap-add runs over this rank's output blocks -- frames oldest first,
        exactly as the unsharded pass adds them.  Returns the finished samples of own_range() [S, hi - lo]."""
        torch, me, K, be = self.torch, self.me, self.K, self.be
        lo, hi = self.own_range()
        with self._ctx():
            out = be.scratch("own_wav", (self.S, self.max_len), torch.float32)
            if me.num_frames == 0:
                return out[:, :0]
            need_lo = max(me.t_lo - K, 0)
            edge = me.t_lo                      # frames [edge, t_lo) are in place
            for p in reversed([p for p in self.plans[:self.rank] if p.num_frames > 0]):
                if edge <= need_lo:
                    break
                cnt = min(K, p.num_frames)
                a, b = max(p.t_hi - cnt, need_lo), min(p.t_hi, edge)
                if b > a:
                    assert tuple(all_rows.shape) == (self.world, self.S, max(K, 1), self.frame_len), all_rows.shape
                    piece = all_rows[p.rank][:, K - (p.t_hi - a):K - (p.t_hi - b), :].contiguous()
                    be.seam_rows(a, b, piece, True)
                    edge = a
            q_hi = self.TL - 1 + self.ovl if me.t_hi == self.TL else me.t_hi
            be.overlap_add(edge, me.t_hi, me.t_lo, q_hi, out, me.t_lo)
            return out[:, :hi - lo]

    def join_shards(self, all_shards, out=None):
        """Place every rank's shard at its sample offset: a rank's inner output blocks are its own, the block at a seam
        is the sum of the left rank's last and the right rank's first block (one frame's contribution each).  General frame
        geometry: `all_shards` holds the ranks' FINISHED ranges, placed as they are."""
        torch, hop = self.torch, self.hop_samples
        with self._ctx():
            if out is None:
                out = torch.empty((self.S, self.n_out), dtype=all_shards.dtype, device=all_shards.device)
            if self.general:
                for p in self.plans:
                    if p.num_frames > 0:
                        a = p.sample_lo
                        b = self.n_out if p.t_hi == self.TL else p.t_hi * hop
                        out[:, a:b].copy_(all_shards[p.rank][:, :b - a])
                return out
            if hasattr(self.be, "join_shards"):      # the HIP backend: one kernel over the output
                self.be.join_shards(all_shards, self.plans, out)
                return out
            live = [p for p in self.plans if p.num_frames > 0]
            for k, p in enumerate(live):
                sh = all_shards[p.rank]
                lo, ln = p.sample_lo, min(p.shard_len, self.n_out - p.sample_lo)
                if k == 0:
                    out[:, lo:lo + hop].copy_(sh[:, :hop])
                else:
                    q = live[k - 1]
                    torch.add(all_shards[q.rank][:, q.shard_len - hop:q.shard_len], sh[:, :hop], out=out[:, lo:lo + hop])
                out[:, lo + hop:lo + ln].copy_(sh[:, hop:ln])
            return out


def _all_gather(dist, send, world, comm_dev, cabi=None):
    """[...] on every rank -> [world, ...]; through host memory when the process group lives there (gloo).
    cabi: a _lib.Handle with a communicator (css_comm_init): the transfer goes through the C ABI's css_comm_all_gather --
    RCCL on the handle's stream, no torch.distributed on the data path (what a host in another language would call)"""
    import torch
    if cabi is not None:
        src = send.contiguous()
        recv = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        cabi.comm_all_gather(src.data_ptr(), recv.data_ptr(), src.numel() * src.element_size())
        return recv
    src = send if send.device == comm_dev else send.to(comm_dev)
    recv = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(recv, src.contiguous())
    else:
        dist.all_gather(list(recv.unbind(0)), src.contiguous())
    return recv if recv.device == send.device else recv.to(send.device)


def sharded_separate_and_stitch(backend, num_spks: int, seg_frames: int, hop_frames: int, hop_samples: int,
                                rank: int, world: int, dist=None, out=None, gather: str = "all", segment_groups=None,
                                trace=None, check_range: bool = True, frame_len=None):
    """Runs one rank's share of a session that `backend.begin(...)` has opened.

    gather="all" (default): returns the full separated waveforms [S, n_out] (a tensor on the backend's device),
    identical on every rank and identical to the single-rank result -- what css.py:110 returns.
    gather="range": returns (wav, (lo, hi)): the finished samples [lo, hi) of this rank's own range, [S, hi - lo]; the
    ranges of the ranks tile [0, n_out) and their concatenation is the single-rank result, bit for bit.  Only one
    256-sample block per stream crosses between neighbours.
    segment_groups: upload_schedule's groups when the session's samples arrive piece by piece (HipShardBackend.begin(cuts=)).

    LIFETIME of the result: with `out=None`, gather="range" (and world == 1) return a VIEW of a work tensor the backend
    keeps between sessions -- valid until the next begin() on the same backend; pass `out` (gather="all": [S, n_out];
    gather="range": [S, >= hi - lo]) or copy it on the backend's stream to keep it.  gather="all" with world > 1
    allocates a fresh tensor when `out` is None.

    check_range=True (default) ends with backend.check_range(): one host synchronisation, after which a Linear-layer
    operand that left the split-f16 range (split_f16.hpp) raises CssError(CSS_ERR_RANGE) instead of returning NaN
    waveforms (repeat the session after handle.set_linear_mode("exact_f32")).  With world > 1 the verdict is agreed on
    by all ranks (one all-reduce of a flag): every rank raises when any rank overflowed, none leaves the collective
    sequence.  check_range=False keeps the call
    asynchronous on the backend's stream: the CALLER then checks once it has synchronised (bench.py does, at its barrier).
    trace: optional callable(label), called on the host between the phases (bench.py records an event on the stream)."""
    assert gather in ("all", "range"), gather
    ss = ShardedSession(backend, num_spks, seg_frames, hop_frames, hop_samples, rank, world, segment_groups, frame_len)
    comm_dev = getattr(backend, "comm_dev", None)
    cabi = getattr(backend, "cabi_comm", None)   # HipShardBackend(cabi_comm=True): the handle, its RCCL communicator initialised
    mark = trace if trace is not None else (lambda label: None)

    def finish(result):
        if not (check_range and hasattr(backend, "check_range")):
            return result
        # The verdict is COLLECTIVE: a rank whose own segments overflowed has already sent NaN costs (and, with
        # gather="all", its NaN shard) to every other rank, so all ranks must raise -- and all must stay in the same
        # collective sequence for the next session.  Every rank reads its own flag, the ranks agree on the maximum.
        mine, err = 0, None
        try:
            backend.check_range()
        except _lib.CssError as e:
            if e.code != _lib.CSS_ERR_RANGE:
                raise
            mine, err = 1, e
        if world > 1:
            import torch
            dev = comm_dev if comm_dev is not None else getattr(backend, "dev", "cpu")
            flag = torch.tensor([mine], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and err is None:
                err = _lib.CssError(_lib.CSS_ERR_RANGE, "an operand of a Linear layer left the split-f16 range on another rank "
                                                        "of this sharded session (its costs / shard reached this rank): "
                                                        "repeat the session after set_linear_mode('exact_f32')")
        if err is not None:
            raise err
        return result

    with ss._ctx():
        costs = ss.segments_and_costs()
        mark("segments")
        if world > 1:
            costs = _all_gather(dist, costs, world, comm_dev if comm_dev is not None else costs.device, cabi)
        mark("exchange_costs")
        act = ss.masks_and_activity(costs)
        mark("scan_stitch_masks")
        if world > 1:
            act = _all_gather(dist, act, world, comm_dev if comm_dev is not None else act.device, cabi)
        mark("exchange_activity")
        shard = ss.gate_and_istft(act)
        mark("gate_istft")
        if ss.general:
            # frame_len != 2 hop: the synthesis rows of the last ovl - 1 frames cross the seam, every rank finishes its own range
            # with the single-GPU overlap-add; gather="all" then all-gathers the finished ranges and places them
            rows = shard
            if world > 1:
                rows = _all_gather(dist, rows, world, comm_dev if comm_dev is not None else rows.device, cabi)
            else:
                rows = rows[None]
            own, rng = ss.finish_range(None, rows), ss.own_range()
            if gather == "range":
                if out is not None:
                    out[:, :rng[1] - rng[0]].copy_(own)
                    own = out[:, :rng[1] - rng[0]]
                mark("exchange_waveforms")
                return finish((own, rng))
            if world == 1:
                res = own[:, :ss.n_out]
                if out is not None:
                    out.copy_(res)
                    res = out
                mark("exchange_waveforms")
                return finish(res)
            full = backend.scratch("own_wav", (ss.S, ss.max_len), own.dtype)      # (the tensor `own` is a view of)
            shards = _all_gather(dist, full, world, comm_dev if comm_dev is not None else full.device, cabi)
            res = ss.join_shards(shards, out)
            mark("exchange_waveforms")
            return finish(res)
        if gather == "range":
            seams = None
            if world > 1:
                seam = ss.seam_piece(shard)
                seams = _all_gather(dist, seam, world, comm_dev if comm_dev is not None else seam.device, cabi)
            own, rng = ss.finish_range(shard, seams), ss.own_range()
            if out is not None:
                out[:, :rng[1] - rng[0]].copy_(own)
                own = out[:, :rng[1] - rng[0]]
            mark("exchange_waveforms")
            return finish((own, rng))
        if world == 1:
            res = shard[:, :ss.n_out]
            if out is not None:
                out.copy_(res)
                res = out
            mark("exchange_waveforms")
            return finish(res)
        shards = _all_gather(dist, shard, world, comm_dev if comm_dev is not None else shard.device, cabi)
        res = ss.join_shards(shards, out)
        mark("exchange_waveforms")
        return finish(res)

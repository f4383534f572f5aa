"""TEST INFRASTRUCTURE: an oracle-backed stand-in for the HIP stage backend, so that the segment-sharding
driver (notsofar1-challenge_amd/parallel.py) can be exercised with world_size > 1 on a box without a GPU
(gloo).  It implements the stage calls of HipShardBackend with oracle/css_oracle.py functions, in the
same gather form as the kernels (a frame collects its <= 2 covering segments).  Never imported by the
product package."""
import numpy as np
import torch

import css_oracle as O


class OracleStageBackend:
    def __init__(self, params, ocfg, num_spks=3):
        self.params, self.cfg, self.S = params, ocfg, num_spks
        self.calls = {"masknet_segments": 0}
        self.comm_dev = torch.device("cpu")
        self._scratch = {}

    # ---- session
    def begin(self, pcm, n, c, run_cfg=None, sample_range=None):
        x = np.asarray(pcm, dtype=np.float32).reshape(n, c)
        self.x = x
        self.c = c
        self.op = O.make_plan(n, 16000, self.cfg)
        p = self.op
        self.T, self.hop, self.TL, self.nseg = p.segment_frames, p.hop_frames, p.mix_frames, p.num_segments
        self.F = 257
        self.X = np.full((self.F, self.TL, c), np.nan + 0j, dtype=np.complex64)   # poisoned until computed
        self.masks = [None] * self.nseg
        self.sep = [None] * self.nseg
        self.costs = np.full((max(self.nseg - 1, 1), self.S * self.S), np.nan)
        self.perms = None
        self.level = np.zeros(1, np.float32)
        self.mask_st = np.full((self.S, self.F, self.TL), np.nan, np.float32)
        self.act_b = np.zeros((self.S, self.TL), np.uint8)
        self.Y = np.full((self.S, self.F, self.TL), np.nan + 0j, dtype=np.complex64)
        self.w = [O.calc_segment_weight(self.T, p.m0_frames, p.m1_frames, is_first_seg=True),
                  O.calc_segment_weight(self.T, p.m0_frames, p.m1_frames),
                  O.calc_segment_weight(self.T, p.m0_frames, p.m1_frames, is_last_seg=True)]

    def plan(self):
        class P:  # the fields ShardedSession reads from CssPlan
            pass
        p = P()
        p.num_segments, p.mix_frames = self.nseg, self.TL
        p.stft_frames = O.num_frames(self.x.shape[0])
        p.n_out = (self.TL - 1) * 256 + 512
        return p

    # ---- stages
    def stft_range(self, lo, hi):
        full = O.stft(self.x)  # [F, T, C]
        hi = min(hi, full.shape[1])
        if hi > lo:
            self.X[:, lo:hi] = full[:, lo:hi]
        if full.shape[1] < self.TL:
            self.X[:, full.shape[1]:] = 0

    def _segment(self, i):
        st, en, t = self.op.seg_range(i)
        seg = np.zeros((self.F, self.T, self.c), np.complex64)
        seg[:, :t] = self.X[:, st:en]
        assert np.isfinite(seg).all(), f"segment {i} reads frames this rank never transformed"
        return seg, t

    def masknet(self, lo, hi):
        for i in range(lo, hi):
            seg, _ = self._segment(i)
            self.masks[i] = O.separate(self.params, seg if self.c > 1 else seg[:, :, 0])
            self.calls["masknet_segments"] += 1

    def mvdr(self, lo, hi):
        for i in range(lo, hi):
            seg, _ = self._segment(i)
            spk, noi = self.masks[i]
            if self.c > 1 and self.cfg.mc_mvdr:
                mv = O.make_mvdr(np.moveaxis(spk, 2, 0), np.moveaxis(noi, 2, 0), np.moveaxis(seg, 2, 0))
                y = np.stack(mv, axis=-1).astype(np.complex64)
                floor = 10.0 ** (self.cfg.mc_mask_floor_db / 20.0)
            else:
                y = seg[:, :, :1]
                floor = 10.0 ** ((self.cfg.mc_mask_floor_db if self.c > 1 else self.cfg.sc_mask_floor_db) / 20.0)
            self.sep[i] = (y * np.maximum(spk, np.float32(floor))).astype(np.complex64)

    def pit_costs(self, lo, hi):
        ov = self.T - self.hop
        for b in range(lo, hi):
            assert self.masks[b] is not None and self.masks[b + 1] is not None, f"boundary {b}: halo segment missing"
            _, _, cost = O.pit_perm(self.masks[b][0][:, -ov:], self.masks[b + 1][0][:, :ov], self.cfg.stitching_loss)
            self.costs[b] = cost.reshape(-1)

    # views the exchanges read and write in place (the HIP backend hands out zero-copy views of device buffers)
    def costs_view(self):
        return torch.from_numpy(self.costs)

    def act_view(self):
        return torch.from_numpy(self.act_b)

    def level_view(self):
        return torch.from_numpy(self.level)

    def scratch(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        if key not in self._scratch:
            self._scratch[key] = torch.zeros(tuple(shape), dtype=dtype)
        return self._scratch[key]

    def index_tensor(self, key, make):
        return torch.from_numpy(make())

    def pit_scan(self):
        import itertools
        num_spks = self.S
        costs = self.costs[:max(self.nseg - 1, 0)]
        assert np.isfinite(costs).all(), "a boundary cost was never computed or exchanged"
        perms = [tuple(range(num_spks))]
        for c in costs.reshape(-1, num_spks, num_spks):
            lp = perms[-1]
            best, arg = None, None
            for sig in itertools.permutations(range(num_spks)):
                tot = sum(c[lp[a], sig[a]] for a in range(num_spks))
                if best is None or tot < best:
                    best, arg = tot, sig
            perms.append(arg)
        self.perms = np.array(perms, dtype=np.int32)

    def _contrib(self, t):
        out = []
        for seg in range(t // self.hop - 3, t // self.hop + 1):
            tl = t - seg * self.hop
            if 0 <= seg < self.nseg and 0 <= tl < self.T:
                w = self.w[0] if seg == 0 else (self.w[2] if seg == self.nseg - 1 else self.w[1])
                out.append((seg, tl, w[tl]))
        return out

    def stitch_masks(self, lo, hi):
        for t in range(lo, hi):
            cs = self._contrib(t)
            ws = np.float32(0)
            for _, _, w in cs:
                ws = np.float32(ws + w)
            for s in range(self.S):
                acc = None
                for seg, tl, w in cs:
                    assert self.masks[seg] is not None, f"frame {t}: segment {seg} missing on this rank"
                    v = np.float32(w) * self.masks[seg][0][:, tl, self.perms[seg][s]]
                    acc = v if acc is None else acc + v
                self.mask_st[s, :, t] = acc / ws
                a = np.float32(np.mean(self.mask_st[s, :, t], dtype=np.float64))
                self.act_b[s, t] = 1 if a >= np.float32(self.cfg.activity_th) else 0

    def stitch_gate(self, lo, hi):
        p = self.op
        final = np.stack([O.erode(O.dilate(self.act_b[s].astype(bool), p.dilation_frames), p.erosion_frames)
                          for s in range(self.S)])
        self.act_final = final
        for t in range(lo, hi):
            cs = self._contrib(t)
            ws = np.float32(0)
            for _, _, w in cs:
                ws = np.float32(ws + w)
            for s in range(self.S):
                acc = None
                for seg, tl, w in cs:
                    v = np.float32(w) * self.sep[seg][:, tl, self.perms[seg][s]]
                    acc = v if acc is None else acc + v
                self.Y[s, :, t] = (acc / ws) * np.float32(final[s, t])

    def istft_partial(self, lo, hi, out_tensor):
        num_spks = self.S
        out = out_tensor.numpy()
        out[:, :(hi - lo + 1) * 256] = 0
        if hi > lo:
            y = self.Y[:, :, lo:hi]
            assert np.isfinite(y).all()
            k = O.istft_kernel()
            c = np.concatenate([y.real, y.imag], axis=1).astype(np.float32)  # [S, 2F, nt]
            for s in range(num_spks):
                g = c[s].T @ k  # [nt, 512]
                for q in range(lo, hi + 1):
                    v = np.zeros(256, np.float32)
                    if lo <= q - 1 < hi:
                        v = g[q - 1 - lo, 256:].copy()
                    if lo <= q < hi:
                        v = v + g[q - lo, :256]
                    out[s, (q - lo) * 256:(q - lo + 1) * 256] = v

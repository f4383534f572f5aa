"""pipeline.css_sessions' rolling window on the CPU: a FAKE handle that writes a session's output only when the loop waits for
it (css_wait_sessions / css_wait), so a session handed to the file writers too early would leave its poison in the wav file.
Checks the order of calls (nothing is released before it was waited for; the window never holds more than 2 x queue_depth
sessions; a full css_wait closes a window when the model kind changes, on a float-path session and at the end), the stepwise
drain of the last window, rank striding, and that every file holds its own session's samples (css/css.py:51-107's layout)."""
import os
import threading

import numpy as np
import pytest

from conftest import pkg


class _FakeHandle:
    """css_run_enqueue_pcm16 / css_wait_sessions / css_wait with the queue's bookkeeping and no GPU: the 'separated streams' of a
    session are channel 0, 1, 2 of its input, negated -- written into the caller's buffer only when the session is waited for."""

    def __init__(self, log, name):
        self.log, self.name = log, name
        self.queue = []       # (planes, out16, peaks) since the last wait
        self.done = 0

    def run_enqueue_pcm16(self, planes, cfg, out16, peaks):
        out16[...] = 12345                       # poison: what a premature hand-over would write to the file
        self.queue.append(([np.array(p) for p in planes[:3]], out16, peaks))
        self.log.append((self.name, "enqueue", len(self.queue)))
        return out16

    def _finish(self, upto):
        for planes, out16, peaks in self.queue[self.done:upto]:
            n = out16.shape[1]
            for i in range(out16.shape[0]):
                out16[i] = -planes[min(i, len(planes) - 1)][:n]
            if peaks is not None:
                peaks[:3] = 1.0
        self.done = max(self.done, upto)

    def wait_sessions(self, n):
        assert 0 < n <= len(self.queue), (n, len(self.queue))
        self.log.append((self.name, "wait_sessions", n))
        self._finish(n)

    def wait(self):
        self.log.append((self.name, "wait", len(self.queue)))
        self._finish(len(self.queue))
        self.queue, self.done = [], 0


class _FakeSeparator:
    def __init__(self, log, name, desc):
        self.handle, self.desc, self.closed = _FakeHandle(log, name), desc, False

    def to(self, device):
        return self

    def eval(self):
        return self

    def close(self):
        self.closed = True


@pytest.fixture()
def loop(monkeypatch):
    L, PIPE = pkg("_lib"), pkg("pipeline")
    monkeypatch.setattr(L, "pinned_empty", lambda shape, dtype=np.float32: np.empty(shape, dtype))   # (page-locked memory needs a GPU)
    PIPE._POOL.clear()
    yield PIPE
    PIPE._POOL.clear()


def _sessions(tmp_path, n, wavio, mc_every=1):
    import pandas as pd
    rows, truth = [], {}
    rs = np.random.RandomState(3)
    for i in range(n):
        is_mc = (i % mc_every == 0) if mc_every > 1 else True
        nch = 7 if is_mc else 1
        ns = 48000 + 256 * int(rs.randint(0, 60)) + 512
        names, planes = [], []
        for c in range(nch):
            x = rs.randint(-3000, 3000, ns).astype(np.int16)
            p = str(tmp_path / f"s{i:02d}_c{c}.wav")
            wavio.write_pcm16_samples(p, x, 16000)
            names.append(p)
            planes.append(x)
        rows.append({"wav_file_names": names, "session_id": f"s{i:02d}", "is_mc": is_mc})
        truth[f"s{i:02d}"] = planes
    return pd.DataFrame(rows), truth


def _check_files(out, truth, wavio, desc_n_out):
    for _, row in out.iterrows():
        planes = truth[row.session_id]
        assert len(row.sep_wav_file_names) == 3
        for i, f in enumerate(row.sep_wav_file_names):
            assert os.path.basename(f) == f"sep_stream{i}.wav" and os.path.basename(os.path.dirname(f)) == row.session_id
            y, sr = wavio.read_wav_pcm16(f)
            src = planes[min(i, len(planes) - 1)]
            assert sr == 16000 and np.array_equal(y, -src[:len(y)]), (row.session_id, i, y[:4], src[:4])
        mix, sr = wavio.read_wav_pcm16(os.path.join(os.path.dirname(row.sep_wav_file_names[0]), "input_mixture.wav"))
        assert sr == 16000 and len(mix) == len(planes[0])


@pytest.mark.parametrize("n,depth", [(5, 12), (30, 4), (61, 3), (9, 1)])
def test_rolling_window_releases_only_what_was_waited_for(tmp_path, loop, n, depth):
    CSS, W, WIO = pkg("css"), pkg("weights"), pkg("wavio")
    df, truth = _sessions(tmp_path, n, WIO)
    log = []
    sep = _FakeSeparator(log, "mc", W.ModelDesc.mc_v1())
    stats = {}
    out = loop.css_sessions(str(tmp_path / "out"), "unused", df, CSS.CssCfg(activity_th=0.3, show_progressbar=False), separators={True: sep},
                            queue_depth=depth, io_threads=3, stats=stats)
    assert list(out.session_id) == list(df.session_id) and not sep.closed          # (a borrowed separator is not closed)
    _check_files(out, truth, WIO, None)
    # the window: never more than 2 x depth sessions un-released; every wait_sessions asks for sessions that exist; a full wait at the end
    enq = rel = 0
    for who, what, k in log:
        if what == "enqueue":
            enq += 1
            assert enq - rel <= 2 * depth
        elif what == "wait_sessions":
            assert rel < k <= enq
            rel = k
        else:
            rel, enq = 0, 0
    assert log[-1][1] == "wait" and sum(1 for _, w, _ in log if w == "enqueue") == n
    if n > 2 * depth:
        assert any(w == "wait_sessions" for _, w, _ in log)
    # the last window leaves in steps: the final full wait finds at most depth // 3 (>= 1) sessions that were not yet waited for
    tail = [(w, k) for _, w, k in log if w != "enqueue"]
    if len(tail) >= 2 and tail[-2][0] == "wait_sessions":
        assert tail[-1][1] - tail[-2][1] <= max(depth // 3, 1)
    assert stats["sessions"] == n and stats["total_s"] > 0


def test_model_kinds_alternate_and_ranks_stride(tmp_path, loop):
    CSS, W, WIO = pkg("css"), pkg("weights"), pkg("wavio")
    df, truth = _sessions(tmp_path, 14, WIO, mc_every=3)
    log = []
    seps = {True: _FakeSeparator(log, "mc", W.ModelDesc.mc_v1()), False: _FakeSeparator(log, "sc", W.ModelDesc.sc_v1())}
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    out = loop.css_sessions(str(tmp_path / "out"), "unused", df, cfg, separators=seps, queue_depth=2, io_threads=2)
    _check_files(out, truth, WIO, None)
    # a handle's queue is closed (full wait) before the other kind's opens: between two enqueues of different handles lies a wait of the first
    last = None
    for who, what, k in log:
        if what == "enqueue":
            if last is not None and last != who:
                assert closed == last, log
            last = who
        elif what == "wait":
            closed = who
    # ranks take every world-th session
    out1 = loop.css_sessions(str(tmp_path / "out_r1"), "unused", df, cfg, separators=seps, queue_depth=2, io_threads=2, rank=1, world=3)
    assert list(out1.session_id) == [f"s{i:02d}" for i in range(1, 14, 3)]
    _check_files(out1, truth, WIO, None)
    # the cache rule of css.py:79-82: a second call with fetch_from_cache answers from the directory, nothing is queued
    n_before = len(log)
    again = loop.css_sessions(str(tmp_path / "out"), "unused", df, cfg, fetch_from_cache=True, separators=seps, queue_depth=2, io_threads=2)
    assert len(log) == n_before and [len(x) for x in again.sep_wav_file_names] == [3] * 14

"""The three GEMM kernels of the path as stand-alone Linear layers (css_linear_host) against a float64 product:
error envelope over operand magnitudes 1e-6 .. 1e3 and K in {512, 1024, 1824}, bit equality of every tile layout of a
kernel (rows are independent of the launch shape: that is what makes the path batch-, lane- and shard-invariant), and
the operand range of the split-f16 format (nothing is clamped; an out-of-range pass is repeated in float32).
Needs an MI355X."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu

KERNELS = {0: ("split-f16, weights direct", (0, 32, 64, 96, 4, 128, 65)),   # 65: the 64-row tile on 64-bit global loads (operands of 2 GiB and more)
           1: ("split-f16, LDS staged", (0, 8, 4, 64)),
           2: ("exact float32", (0, 2, 8, 4, 64, 11, 12, 13, 14))}   # 0: gemm_f32.hip (persistent, balanced); 2 / 8 / 4 / 64: gemm.hip; 11..14: gemm_f32.hip, tiles of at most 32..128 rows


@pytest.fixture(scope="module")
def handle():
    L = pkg("_lib")
    if L.load().css_device_count() < 1:
        pytest.fail("no HIP device visible")
    w = pkg("weights")
    desc = w.ModelDesc(num_blocks=1)
    sep = pkg("separator").HipSeparator(w.apply_golden_recipe(w.portable_state_dict(desc, 5)), None, device=0)
    yield sep.handle
    sep.close()


def _case(rs, m, n, k, sx, sw):
    x = (rs.standard_normal((m, k)) * sx).astype(np.float32)
    w = (rs.standard_normal((n, k)) * sw).astype(np.float32)
    b = (rs.standard_normal(n) * sx * sw).astype(np.float32)
    y64 = x.astype(np.float64) @ w.astype(np.float64).T + b
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)   # sum of |terms| per output
    return x, w, b, y64, scale


@pytest.mark.parametrize("k", [512, 1024, 1824])
def test_error_envelope_vs_float64(handle, k):
    """|y - y64| relative to the sum of the absolute terms: float32 accumulation leaves ~sqrt(K) 2^-24; the split
    kernels must sit in the same envelope as the exact float32 kernel (their operands carry 22 bits, the dropped
    lo x lo term is 2^-22 of a product) wherever the operands are inside the format's full-precision range
    2^-14 <= |x| <= 65504 -- which every operand of the path is (LayerNorm outputs, features, activations, weights are
    O(1); the level-dependent spectra of the synthesis transform are scaled into it, split_f16.hpp level_gain)."""
    rs = np.random.RandomState(k)
    worst = {}
    for sx, sw in [(1e-2, 1.0), (1.0, 1e-2), (1.0, 1.0), (1.0, 30.0), (1e3, 1.0), (1e4, 1e-2)]:
        x, w, b, y64, scale = _case(rs, 333, 640, k, sx, sw)
        for kern in KERNELS:
            y = handle.linear(x, w, b, kernel=kern)
            assert np.isfinite(y).all(), (kern, sx, sw)
            worst[kern] = max(worst.get(kern, 0.0), float((np.abs(y - y64) / scale).max()))
    print(f"K={k}: max |err| / sum|terms|:", {KERNELS[kk][0]: f"{v:.2e}" for kk, v in worst.items()})
    for kern, v in worst.items():
        assert v < 1.5e-6, (KERNELS[kern][0], v)
    assert worst[0] < 3 * worst[2] + 1e-7 and worst[1] < 3 * worst[2] + 1e-7


def test_small_magnitude_regime_is_bounded(handle):
    """Below 2^-14 an operand is carried by its low part alone: 11 significant bits, i.e. an ABSOLUTE error of at most
    2^-26 per operand (3e-8 against the O(1) terms of the path).  With every operand down there the relative error is
    2^-11-grade -- the documented limit, never reached on the path -- and the exact kernel is unaffected."""
    rs = np.random.RandomState(3)
    x, w, b, y64, scale = _case(rs, 200, 256, 512, 1e-6, 1.0)
    e_split = float((np.abs(handle.linear(x, w, b, kernel=0) - y64) / scale).max())
    e_exact = float((np.abs(handle.linear(x, w, b, kernel=2) - y64) / scale).max())
    print(f"all operands ~1e-6: split {e_split:.2e}, exact {e_exact:.2e}")
    assert e_exact < 1.5e-6 and e_split < 2.5e-4
    absolute = float(np.abs(handle.linear(x, w, b, kernel=0) - y64).max())
    assert absolute < 512 * 2.0 ** -26 * 4.5       # K operands x 2^-26 x max |w|


@pytest.mark.parametrize("shape", [(100, 512, 512), (7440, 1536, 512), (23808, 512, 1024), (1028, 22506, 512),
                                   (28125, 512, 544), (777, 1028, 1824), (7440, 512, 512), (7440, 512, 32), (7440, 512, 64),
                                   (7441, 512, 96), (129, 128, 1024)])
def test_tile_layouts_give_the_same_bits(handle, shape):
    """Every tile layout of a kernel, and the launcher's own choice at this shape, writes the same bits -- including
    the large launches (long meetings, 128-segment batches) where the launcher switches layouts."""
    m, n, k = shape
    rs = np.random.RandomState(m + n)
    x, w, b, y64, scale = _case(rs, m, n, k, 1.0, 0.5)
    split = None
    for kern, (name, layouts) in KERNELS.items():
        if kern == 0 and n % 32:
            continue   # the weights-direct kernel takes whole 32-column weight tiles (every Linear layer of the encoder has them)
        ref = None
        for lay in layouts:
            y = handle.linear(x, w, b, kernel=kern, layout=lay)
            if ref is None:
                ref = y
                assert float((np.abs(y - y64) / scale).max()) < 1.5e-6, (name, lay)
            else:
                assert np.array_equal(y, ref), (name, lay, float(np.abs(y - ref).max()))
        if kern in (0, 1):   # the two split kernels accumulate in the same order
            if split is None:
                split = ref
            elif n % 32 == 0:
                assert np.array_equal(ref, split)


def test_split_operand_range_is_not_clamped(handle):
    """An operand beyond the float16 range must not pass silently: the split kernels return non-finite values (which
    css_run* detects at the end of a pass), the exact kernel the right answer."""
    rs = np.random.RandomState(1)
    x, w, b, y64, scale = _case(rs, 64, 128, 512, 1.0, 1.0)
    x[3, 17] = 1.0e5
    y64 = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert not np.isfinite(handle.linear(x, w, b, kernel=0)[3]).any()
    assert not np.isfinite(handle.linear(x, w, b, kernel=1)[3]).any()
    y = handle.linear(x, w, b, kernel=2)
    assert np.isfinite(y).all() and np.abs(y - y64).max() < 0.5


def test_recording_level_does_not_reach_the_operand_range(mc_state, mix60):
    """Integer-scaled PCM (x 32768) and a recording 48 dB down give the same streams as the unit-scale recording up
    to that factor: the features are level-invariant, the beamformer linear, and the one level-dependent split operand
    (the stitched spectra) is brought to unit peak by a power of two -- no overflow, no 11-bit regime, no fallback."""
    CSS = pkg("css")
    sep = pkg("separator").HipSeparator(mc_state[0], None, device=0, linear_mode="split_f16")   # (the mode with an operand range)
    try:
        h = sep.handle
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
        mix = np.ascontiguousarray(mix60[0, :12 * 16000])
        ref = h.run(mix, run_cfg).astype(np.float64)
        for gain in (32768.0, 2.0 ** -8):
            got = h.run(mix * np.float32(gain), run_cfg).astype(np.float64) / gain
            assert h.range_status() == (0, False)
            rel = np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
            assert rel < 1e-3, (gain, rel)   # winner-take-all flips at float32 rounding level aside (features' eps clamps)
    finally:
        sep.close()


def test_out_of_range_pass_is_repeated_in_float32(mc_state, mix60):
    """A model whose first feed-forward layer drives its ReLU outputs past 65504 (weights themselves in range): the
    split-f16 operand of the next Linear layer overflows -- to inf, not to a clamp --, that GEMM raises the range flag,
    and the pass is repeated on the exact float32 kernels: the result is the exact mode's, bit for bit; with the
    fallback off the same call fails with CSS_ERR_RANGE."""
    L, CSS = pkg("_lib"), pkg("css")
    st = dict(mc_state[0])
    key = pkg("weights").PREFIX + "conformer.encoders.0.feed_forward_in.net.0.weight"
    st[key] = np.asarray(st[key], np.float32) * np.float32(3e5)
    sep = pkg("separator").HipSeparator(st, None, device=0, linear_mode="split_f16")
    try:
        h = sep.handle
        run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
        mix = np.ascontiguousarray(mix60[0, :12 * 16000])
        assert h.range_status() == (0, False) and h.linear_mode() == "split_f16"
        got = h.run(mix, run_cfg)
        assert h.range_status() == (1, True) and np.isfinite(got).all() and h.linear_mode() == "split_f16"
        h.set_linear_mode("exact_f32")
        ref = h.run(mix, run_cfg)
        h.set_linear_mode("split_f16")
        assert np.array_equal(got, ref)
        h.set_range_fallback(False)
        with pytest.raises(L.CssError) as e:
            h.run(mix, run_cfg)
        assert e.value.code == L.CSS_ERR_RANGE
        # queued passes follow the same rule: css_wait repeats every pass queued since the last css_wait on the exact float32
        # kernels (from the caller's buffers) and reports it; with the fallback off it returns CSS_ERR_RANGE
        h.set_range_fallback(True)
        before = h.range_status()[0]
        pin = L.pinned_copy(mix)
        out, out2 = L.pinned_empty(ref.shape, np.float32), np.empty(ref.shape, np.float32)
        h.run_enqueue(pin, run_cfg, out)
        h.run_enqueue(pin, run_cfg, out2)        # (pageable output: the queue changes mode in between)
        h.wait()
        assert h.range_status() == (before + 2, True) and h.linear_mode() == "split_f16"
        assert np.array_equal(out, ref) and np.array_equal(out2, ref)
        h.set_range_fallback(False)
        h.run_enqueue(pin, run_cfg, out)
        with pytest.raises(L.CssError) as e:
            h.wait()
        assert e.value.code == L.CSS_ERR_RANGE
        h.set_range_fallback(True)
        assert np.array_equal(h.run(mix, run_cfg), ref)
    finally:
        sep.close()

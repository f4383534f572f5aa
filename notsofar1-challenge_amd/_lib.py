"""ctypes binding of libcss_mi355.so (the C ABI declared in include/css_mi355.h).

The product path has NO CPU fallback: if the shared library is missing or cannot be loaded, every
compute entry point raises ``CssLibraryError`` (build it with ``python __graft_entry__.py`` or
``make -C notsofar1-challenge_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CSS_MI355_LIBRARY: another build of the SAME library -- the AddressSanitizer build of tools/asan_tests.sh; nothing else)
LIB_PATH = os.environ.get("CSS_MI355_LIBRARY") or os.path.join(_HERE, "libcss_mi355.so")


class CssLibraryError(RuntimeError):
    pass


class CssError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"css_mi355 error {code}: {text}")
        self.code = code
        self.text = text


# status codes (include/css_mi355.h: css_status)
CSS_OK, CSS_ERR_INVALID_ARG, CSS_ERR_HIP, CSS_ERR_ZERO_WEIGHT, CSS_ERR_MASK_FLOOR = 0, -1, -2, -3, -4
CSS_ERR_SHAPE, CSS_ERR_STATE, CSS_ERR_NO_DEVICE, CSS_ERR_WEIGHT_WINDOW, CSS_ERR_RANGE = -5, -6, -7, -8, -9
ANALYSIS_WINDOWS = {"hann": 0, "sqrt_hann": 1}   # CSS_WINDOW_* (css_set_analysis_window)
MAX_SEGMENT_FRAMES = 16384                       # CSS_MAX_SEGMENT_FRAMES

# buffer ids (css_buffer)
(BUF_X, BUF_FEATURES, BUF_MASKS, BUF_SCM, BUF_BFW, BUF_SEP, BUF_PIT_COST, BUF_PERMS, BUF_MASK_ST, BUF_ACTIVITY,
 BUF_ACT_B, BUF_ACT_FINAL, BUF_Y, BUF_WAV, BUF_HIDDEN, BUF_WTA_OVERRIDE, BUF_LEVEL) = range(17)
_BUF_DTYPES = {BUF_SCM: np.float64, BUF_BFW: np.float64, BUF_PIT_COST: np.float64, BUF_PERMS: np.int32,
               BUF_ACT_B: np.uint8, BUF_ACT_FINAL: np.uint8, BUF_WTA_OVERRIDE: np.uint8}


class CssModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_mics", "num_bins", "in_features", "attention_dim", "attention_heads", "linear_units", "num_blocks",
        "kernel_size", "num_spks", "num_nois", "frame_len", "frame_hop", "maxlen")]


class CssFeatureCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("log_spectrogram", "mvn_spectrogram", "ipd_mean_normalize",
                                         "ipd_mean_normalize_version", "ipd_cos", "num_pairs")] + \
               [("pair_l", C.c_int32 * 16), ("pair_r", C.c_int32 * 16)]


class CssRunCfg(C.Structure):
    _fields_ = [("segment_frames", C.c_int32), ("hop_frames", C.c_int32), ("dilation_frames", C.c_int32),
                ("erosion_frames", C.c_int32), ("mc_mvdr", C.c_int32), ("stitching_loss", C.c_int32),
                ("stitching_input", C.c_int32), ("normalize_segment_power", C.c_int32),
                ("mask_floor", C.c_float), ("activity_th", C.c_float),
                ("w_first", C.POINTER(C.c_float)), ("w_mid", C.POINTER(C.c_float)), ("w_last", C.POINTER(C.c_float))]


class CssPlan(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("stft_frames", C.c_int64), ("mix_frames", C.c_int64),
                ("num_segments", C.c_int64), ("n_out", C.c_int64), ("last_valid", C.c_int32),
                ("zero_weight", C.c_int32)]


class CssCfgSeconds(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("segment_size_sec", "hop_size_sec", "seg_weight_m0_sec", "seg_weight_m1_sec",
                                           "activity_dilation_sec", "activity_erosion_sec", "activity_th", "mask_floor_db")] + \
               [(n, C.c_int32) for n in ("mc_mvdr", "stitching_loss", "stitching_input", "normalize_segment_power")]


class CssKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_float), ("launches", C.c_int32)]


class CssTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("upload", "stft", "features", "masknet", "mvdr", "stitch", "istft",
                                          "download", "total", "gemm_ms")] + \
               [("gemm_launches", C.c_int64), ("gemm_flops", C.c_double), ("host_enqueue", C.c_float), ("host_total", C.c_float)]


# name -> (restype, argtypes); exactly the symbols include/css_mi355.h declares
_P = C.c_void_p
_FP = C.POINTER(C.c_float)
SIGNATURES = {
    "css_version": (C.c_char_p, []),
    "css_last_error": (C.c_char_p, [_P]),
    "css_device_count": (C.c_int, []),
    "css_blob_num_floats": (C.c_int64, [C.POINTER(CssModelDesc)]),
    "css_create": (C.c_int, [C.POINTER(CssModelDesc), _P, C.c_int64, C.c_int, _P, C.c_int32, C.POINTER(_P)]),
    "css_destroy": (C.c_int, [_P]),
    "css_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "css_set_lanes": (C.c_int, [_P, C.c_int]),
    "css_get_lanes": (C.c_int, [_P]),
    "css_set_tuning": (C.c_int, [_P, C.c_int, C.c_int]),
    "css_set_queue_group": (C.c_int, [_P, C.c_int]),
    "css_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "css_host_free": (C.c_int, [_P]),
    "css_plan": (C.c_int, [C.POINTER(CssModelDesc), C.POINTER(CssRunCfg), C.c_int64, C.POINTER(CssPlan)]),
    "css_make_run_cfg": (C.c_int, [C.POINTER(CssModelDesc), C.POINTER(CssCfgSeconds), C.c_int32, C.POINTER(CssRunCfg), _FP, C.c_int64]),
    "css_pit_scan": (C.c_int, [_P, C.c_int64, C.c_int32, _P]),
    "css_run": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), _P, C.c_int64]),
    "css_run_enqueue": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), _P, C.c_int64]),
    "css_wait": (C.c_int, [_P]),
    "css_wait_sessions": (C.c_int, [_P, C.c_int64]),
    "css_run_device": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), _P, C.c_int64]),
    "css_run_pcm16": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), _P, C.c_int64, _P]),
    "css_run_enqueue_pcm16": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), _P, C.c_int64, _P]),
    "css_get_timings": (C.c_int, [_P, C.POINTER(CssTimings)]),
    "css_set_profile": (C.c_int, [_P, C.c_int]),
    "css_get_kernel_stats": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "css_set_linear_mode": (C.c_int, [_P, C.c_int]),
    "css_get_linear_mode": (C.c_int, [_P]),
    "css_set_range_fallback": (C.c_int, [_P, C.c_int]),
    "css_range_status": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "css_check_range": (C.c_int, [_P]),
    "css_linear_host": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "css_get_plan": (C.c_int, [_P, C.POINTER(CssPlan)]),
    "css_begin": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), C.c_int]),
    "css_begin_range": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(CssRunCfg), C.c_int64, C.c_int64]),
    "css_upload_range": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "css_stage_stft": (C.c_int, [_P]),
    "css_stage_stft_range": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_stitch_masks": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_stitch_gate": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_istft_partial": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64]),
    "css_stage_masknet": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_mvdr": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_pit_costs": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_pit_scan": (C.c_int, [_P]),
    "css_stage_stitch": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_istft": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_join_shards": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int64]),
    "css_stage_synthesis": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "css_stage_seam_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int32]),
    "css_stage_overlap_add": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64]),
    "css_sync": (C.c_int, [_P]),
    "css_stft_host": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int64]),
    "css_separate_host": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "css_set_feature_options": (C.c_int, [_P, C.POINTER(CssFeatureCfg)]),
    "css_set_analysis_window": (C.c_int, [_P, C.c_int32]),
    "css_forward_host": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "css_istft_host": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    "css_handoff_logmel": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64,
                                     C.POINTER(C.c_int64), _P, C.c_int32, C.POINTER(C.c_int32)]),
    "css_validation_loss_host": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_float, _P, _P, _P, C.POINTER(C.c_float)]),
    "css_comm_unique_id": (C.c_int, [_P]),
    "css_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "css_comm_destroy": (C.c_int, [_P]),
    "css_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "css_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_int64]),
    "css_buffer_dims": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "css_read_buffer": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "css_write_buffer": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "css_buffer_devptr": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
}

_lib: Optional[C.CDLL] = None


def _share_hip_runtime_with_torch() -> None:
    """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7, found through torch's RPATH under the name "libamdhip64.so"), while
    libcss_mi355.so asks the loader for "libamdhip64.so.7".  If our library is loaded first, the system
    runtime gets in, torch later adds its bundled one, and torch then sees no GPU.  Loading torch's copy
    first (without importing torch) makes the loader resolve our dependency to that same object by
    SONAME, whichever side is imported first.  Without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):  # pragma: no cover
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:  # pragma: no cover
            pass


def load() -> C.CDLL:
    """Load the library once; raise loudly (no fallback) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CssLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            f"(or `make -C notsofar1-challenge_amd/csrc`). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CssLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def check(handle, rc: int):
    if rc != CSS_OK:
        text = load().css_last_error(handle).decode("utf-8", "replace")
        if rc == CSS_ERR_ZERO_WEIGHT:
            # same exception type and text as the reference's assert (css/css.py:297)
            raise AssertionError(text or "zero weights found. check hop_size, segment_size or m0, m1")
        if rc == CSS_ERR_MASK_FLOOR:
            raise AssertionError(text)  # css/css.py:224
        raise CssError(rc, text)


def make_desc(d) -> CssModelDesc:
    return CssModelDesc(*[int(getattr(d, n)) for n, _ in CssModelDesc._fields_])


class RunCfg:
    """Owns the CssRunCfg struct together with the numpy weight windows it points to."""

    def __init__(self, segment_frames, hop_frames, dilation_frames, erosion_frames, mc_mvdr, stitching_loss,
                 stitching_input, normalize_segment_power, mask_floor, activity_th, w_first, w_mid, w_last):
        self._w = [np.ascontiguousarray(w, dtype=np.float32) for w in (w_first, w_mid, w_last)]
        for w in self._w:
            assert w.shape == (segment_frames,)
        self.c = CssRunCfg(int(segment_frames), int(hop_frames), int(dilation_frames), int(erosion_frames),
                           int(bool(mc_mvdr)), int(stitching_loss), int(stitching_input),
                           int(bool(normalize_segment_power)), float(mask_floor), float(activity_th),
                           self._w[0].ctypes.data_as(_FP), self._w[1].ctypes.data_as(_FP),
                           self._w[2].ctypes.data_as(_FP))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI: rank 0 calls it and hands the 128 bytes to every rank (over its own channel)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().css_comm_unique_id(buf)
    if rc != 0:
        raise CssError(rc, "css_comm_unique_id failed (is librccl.so loadable?)")
    return buf.raw


def plan(desc, run_cfg: RunCfg, n_samples: int) -> CssPlan:
    p = CssPlan()
    rc = load().css_plan(C.byref(make_desc(desc)), C.byref(run_cfg.c), int(n_samples), C.byref(p))
    if rc != CSS_OK:
        raise CssError(rc, "css_plan: bad configuration")
    return p


def pit_scan(costs: np.ndarray, num_spks: int) -> np.ndarray:
    costs = np.ascontiguousarray(costs, dtype=np.float64).reshape(-1, num_spks * num_spks)
    perms = np.zeros((costs.shape[0] + 1, num_spks), dtype=np.int32)
    rc = load().css_pit_scan(_np_ptr(costs), costs.shape[0], num_spks, _np_ptr(perms))
    if rc != CSS_OK:
        raise CssError(rc, "css_pit_scan")
    return perms


class _PinnedBlock:
    """Owner of one css_host_alloc block (page-locked host memory), freed when the last array over it dies."""

    def __init__(self, nbytes: int):
        self.lib = load()
        p = C.c_void_p()
        rc = self.lib.css_host_alloc(max(int(nbytes), 1), C.byref(p))
        if rc != CSS_OK or not p.value:
            raise CssError(rc, "css_host_alloc failed")
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.__array_interface__ = {"shape": (max(int(nbytes), 1),), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):  # pragma: no cover
        try:
            self.lib.css_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """numpy array in page-locked host memory (css_host_alloc): css_run* on such buffers moves the samples by DMA,
    asynchronously, overlapped with the first and last kernels of the pass."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    block = _PinnedBlock(n)
    return np.asarray(block)[:n].view(dt).reshape(shape)   # the view chain keeps `block` alive


def pinned_copy(a: np.ndarray) -> np.ndarray:
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Handle:
    """One model resident on one GPU (css_create .. css_destroy)."""

    def __init__(self, desc, blob: np.ndarray, device: int = 0, stream: int = 0, max_batch_segments: int = 64):
        lib = load()
        self.lib = lib
        self.desc = desc
        self._cdesc = make_desc(desc)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        need = lib.css_blob_num_floats(C.byref(self._cdesc))
        if need != blob.size:
            raise CssError(CSS_ERR_INVALID_ARG, f"weight blob has {blob.size} floats, library expects {need}")
        h = C.c_void_p()
        rc = lib.css_create(C.byref(self._cdesc), _np_ptr(blob), blob.size, int(device),
                            C.c_void_p(stream) if stream else None, int(max_batch_segments), C.byref(h))
        if rc != CSS_OK:
            raise CssError(rc, lib.css_last_error(None).decode("utf-8", "replace"))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.css_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def stream_ptr(self) -> int:
        """hipStream_t of the handle (wrap with torch.cuda.ExternalStream to order torch work with its kernels)."""
        p = C.c_void_p()
        check(self.h, self.lib.css_get_stream(self.h, C.byref(p)))
        return int(p.value or 0)

    def set_lanes(self, lanes: int):
        check(self.h, self.lib.css_set_lanes(self.h, int(lanes)))

    def set_tuning(self, which, value: int):
        """which: "tail_pieces" | "out_mapped" | "tail_per_unit" | "mvdr_on_lanes" | "pipeline_device" (include/css_mi355.h css_tuning)"""
        idx = {"tail_pieces": 0, "out_mapped": 1, "tail_per_unit": 2, "mvdr_on_lanes": 3, "pipeline_device": 4,
               "group_lanes": 5, "group_transform_on_main": 6, "group_mvdr_on_lanes": 7, "group_out_dma": 8, "f32_gemm": 9, "split_batch_rows": 10, "f32_lane_rows": 11}[which] if isinstance(which, str) else int(which)
        check(self.h, self.lib.css_set_tuning(self.h, idx, int(value)))

    def set_queue_group(self, max_sessions: int):
        """queued sessions (run_enqueue) merged into one mask-estimator batch: at most `max_sessions` (1 = none)"""
        check(self.h, self.lib.css_set_queue_group(self.h, int(max_sessions)))

    def lanes(self) -> int:
        return int(self.lib.css_get_lanes(self.h))

    # ---- fused path
    def run(self, pcm: np.ndarray, cfg: RunCfg, out: Optional[np.ndarray] = None) -> np.ndarray:
        """pcm [n, C] float32 host -> wav [S, n_out] float32 host (css/css.py:110).  `out`: an existing [S, >= n_out]
        float32 array to fill (page-locked buffers from pinned_empty make both PCIe legs asynchronous DMA)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n, c = pcm.shape
        p = plan(self.desc, cfg, n)
        if out is None:
            out = np.empty((self.desc.num_spks, p.n_out), dtype=np.float32)
        assert out.dtype == np.float32 and out.ndim == 2 and out.shape[0] == self.desc.num_spks and out.shape[1] >= p.n_out \
            and out.strides[1] == 4
        check(self.h, self.lib.css_run(self.h, _np_ptr(pcm), n, c, C.byref(cfg.c), _np_ptr(out), out.strides[0] // 4))
        return out[:, :p.n_out]

    def run_enqueue(self, pcm: np.ndarray, cfg: RunCfg, out: np.ndarray) -> np.ndarray:
        """css_run without the closing synchronisation (a queue of sessions): `pcm` (float32 [n, C], C-contiguous) must
        stay alive and `out` untouched until wait().  With page-locked buffers (pinned_empty / pinned_copy) consecutive
        queued passes overlap: a session's PCIe legs hide under its neighbours' kernels.  Returns the view of `out` that
        wait() makes valid."""
        assert pcm.dtype == np.float32 and pcm.flags.c_contiguous and pcm.ndim == 2
        n, c = pcm.shape
        p = plan(self.desc, cfg, n)
        assert out.dtype == np.float32 and out.ndim == 2 and out.shape[0] == self.desc.num_spks and out.shape[1] >= p.n_out \
            and out.strides[1] == 4
        self._queued_keep = getattr(self, "_queued_keep", []) + [(pcm, out, cfg)]
        check(self.h, self.lib.css_run_enqueue(self.h, _np_ptr(pcm), n, c, C.byref(cfg.c), _np_ptr(out), out.strides[0] // 4))
        return out[:, :p.n_out]

    def wait(self):
        """Blocks until every pass queued with run_enqueue has finished (CssError CSS_ERR_RANGE if one left the split-f16
        range: repeat those sessions with run())."""
        try:
            check(self.h, self.lib.css_wait(self.h))
        finally:
            self._queued_keep = []

    def wait_sessions(self, n: int):
        """Blocks until the first `n` sessions queued since the last wait() have finished (outputs in host memory); the later
        ones keep running.  Not a wait(): call that eventually (it releases the references this object keeps, and in
        "split_f16" mode it is where a session that left the operand range is repeated -- from its input buffers)."""
        check(self.h, self.lib.css_wait_sessions(self.h, int(n)))

    def run_device(self, pcm_ptr: int, n: int, c: int, cfg: RunCfg, wav_ptr: int, cap: int):
        check(self.h, self.lib.css_run_device(self.h, C.c_void_p(pcm_ptr), n, c, C.byref(cfg.c),
                                              C.c_void_p(wav_ptr), cap))

    def run_pcm16(self, planes, cfg: RunCfg):
        """planes: sequence of C mono int16 arrays of equal length (the session's wav payloads).
        -> (int16 [S, n_out] peak-normalised PCM16 samples of the separated streams, float32 [S] peaks)."""
        planes = [np.ascontiguousarray(p, dtype=np.int16) for p in planes]
        n = planes[0].shape[0]
        if any(p.ndim != 1 or p.shape[0] != n for p in planes):
            raise ValueError("all channel planes must be one-dimensional and of equal length")
        c = len(planes)
        pl = plan(self.desc, cfg, n)
        S = int(self.desc.num_spks)
        out = np.empty((S, pl.n_out), dtype=np.int16)
        peaks = np.empty((S,), dtype=np.float32)
        ptrs = (C.c_void_p * c)(*[p.ctypes.data for p in planes])
        check(self.h, self.lib.css_run_pcm16(self.h, ptrs, n, c, C.byref(cfg.c), out.ctypes.data_as(C.c_void_p), pl.n_out,
                                             peaks.ctypes.data_as(C.c_void_p)))
        return out, peaks

    def run_enqueue_pcm16(self, planes, cfg: RunCfg, out16: np.ndarray, peaks: Optional[np.ndarray] = None) -> np.ndarray:
        """run_pcm16 as a queued session (css_run_enqueue_pcm16): `planes` = C mono int16 arrays of equal length, `out16` an
        int16 [S, >= n_out] array, `peaks` None or a float32 [S] array -- all of them must stay alive and untouched until
        wait(); page-locked buffers (pinned_empty / pinned_copy) let the session share an estimator batch with its neighbours
        in the queue and hide its PCIe legs.  Returns the view of `out16` that wait() makes valid."""
        n = planes[0].shape[0]
        for p in planes:
            assert p.dtype == np.int16 and p.ndim == 1 and p.shape[0] == n and p.flags.c_contiguous
        c = len(planes)
        pl = plan(self.desc, cfg, n)
        S = int(self.desc.num_spks)
        assert out16.dtype == np.int16 and out16.ndim == 2 and out16.shape[0] == S and out16.shape[1] >= pl.n_out and out16.strides[1] == 2
        assert peaks is None or (peaks.dtype == np.float32 and peaks.shape == (S,) and peaks.flags.c_contiguous)
        ptrs = (C.c_void_p * c)(*[p.ctypes.data for p in planes])
        self._queued_keep = getattr(self, "_queued_keep", []) + [(planes, out16, peaks, cfg)]
        check(self.h, self.lib.css_run_enqueue_pcm16(self.h, ptrs, n, c, C.byref(cfg.c), out16.ctypes.data_as(C.c_void_p), out16.strides[0] // 2,
                                                     peaks.ctypes.data_as(C.c_void_p) if peaks is not None else None))
        return out16[:, :pl.n_out]

    def timings(self) -> dict:
        t = CssTimings()
        check(self.h, self.lib.css_get_timings(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in CssTimings._fields_}

    def kernel_stats(self) -> dict:
        """{family: (ms, launches)} of the last pass run with set_profile(True)"""
        arr = (CssKernelStat * 32)()
        cnt = C.c_int32()
        check(self.h, self.lib.css_get_kernel_stats(self.h, C.cast(arr, C.c_void_p), 32, C.byref(cnt)))
        return {arr[i].name.decode(): (float(arr[i].ms), int(arr[i].launches)) for i in range(min(cnt.value, 32))}

    def set_profile(self, enable: bool):
        check(self.h, self.lib.css_set_profile(self.h, int(enable)))

    def set_linear_mode(self, mode):
        """"exact_f32" (the default of a new handle: float32 operands, the reference's own precision) or "split_f16" (opt-in:
        22-bit operands as float16 pairs on the f16 matrix cores, ~2x the throughput)."""
        check(self.h, self.lib.css_set_linear_mode(self.h, {"split_f16": 0, "exact_f32": 1}[mode]))

    def set_feature_options(self, log_spectrogram=False, mvn_spectrogram=True, ipd_mean_normalize=True,
                            ipd_mean_normalize_version=1, ipd_cos=False, pairs=None):
        """ExtractorCfg options beyond the shipped configuration (css_set_feature_options); pairs: [(l, r), ...]"""
        c = CssFeatureCfg(int(log_spectrogram), int(mvn_spectrogram), int(ipd_mean_normalize), int(ipd_mean_normalize_version),
                          int(ipd_cos), 0)
        pairs = list(pairs) if pairs is not None else [(m, 0) for m in range(1, int(self.desc.num_mics))]
        c.num_pairs = len(pairs)
        for i, (l, r) in enumerate(pairs[:16]):
            c.pair_l[i], c.pair_r[i] = int(l), int(r)
        check(self.h, self.lib.css_set_feature_options(self.h, C.byref(c)))

    def set_analysis_window(self, window: str):
        """ExtractorCfg.window: 'hann' (default) or 'sqrt_hann' (css_set_analysis_window; feature.py:19-45)"""
        if window not in ANALYSIS_WINDOWS:
            raise RuntimeError("Now only support sqrt hanning window or hann window")   # feature.py:24-25
        check(self.h, self.lib.css_set_analysis_window(self.h, ANALYSIS_WINDOWS[window]))

    def set_range_fallback(self, enable: bool):
        check(self.h, self.lib.css_set_range_fallback(self.h, int(enable)))

    def range_status(self):
        """(passes repeated on the exact float32 kernels so far, whether the last pass was one)"""
        n, last = C.c_int64(), C.c_int32()
        check(self.h, self.lib.css_range_status(self.h, C.byref(n), C.byref(last)))
        return int(n.value), bool(last.value)

    def check_range(self):
        check(self.h, self.lib.css_check_range(self.h))

    def linear(self, x: np.ndarray, w: np.ndarray, bias=None, kernel: int = 0, layout: int = 0) -> np.ndarray:
        """y = x @ w.T + bias through one of the path's GEMM kernels (css_linear_host)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        w = np.ascontiguousarray(w, dtype=np.float32)
        m, k = x.shape
        n = w.shape[0]
        assert w.shape[1] == k
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        y = np.empty((m, n), dtype=np.float32)
        check(self.h, self.lib.css_linear_host(self.h, _np_ptr(x), _np_ptr(w), _np_ptr(b) if b is not None else None,
                                               m, n, k, int(kernel), int(layout), _np_ptr(y)))
        return y

    # ---- RCCL through the C ABI (css_comm_*): what a host in another language would call
    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        assert len(unique_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        check(self.h, self.lib.css_comm_init(self.h, buf, int(nranks), int(rank)))

    def comm_destroy(self):
        check(self.h, self.lib.css_comm_destroy(self.h))

    def comm_info(self):
        n, r, d, v = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self.h, self.lib.css_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d), C.byref(v)))
        return {"nranks": n.value, "rank": r.value, "device": d.value, "rccl_version_code": v.value}

    def comm_all_gather(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int):
        """device pointers; enqueued on the handle's stream"""
        check(self.h, self.lib.css_comm_all_gather(self.h, C.c_void_p(int(send_ptr)), C.c_void_p(int(recv_ptr)), int(bytes_per_rank)))

    def linear_mode(self):
        m = self.lib.css_get_linear_mode(self.h)
        if m < 0:
            check(self.h, m)
        return ("split_f16", "exact_f32")[m]

    def get_plan(self) -> CssPlan:
        p = CssPlan()
        check(self.h, self.lib.css_get_plan(self.h, C.byref(p)))
        return p

    # ---- stages
    def begin(self, pcm, n: int, c: int, cfg: RunCfg, device: bool = False):
        if device:
            ptr = C.c_void_p(int(pcm))
        else:
            self._pcm_keep = np.ascontiguousarray(pcm, dtype=np.float32)
            ptr = _np_ptr(self._pcm_keep)
        check(self.h, self.lib.css_begin(self.h, ptr, n, c, C.byref(cfg.c), int(device)))

    def begin_range(self, pcm: np.ndarray, n: int, c: int, cfg: RunCfg, s_lo: int, s_hi: int, slice_only: bool = False,
                    base_sample: Optional[int] = None):
        """Session over a host recording of n samples of which only samples [s_lo, s_hi) are uploaded (a rank's slice).
        slice_only: `pcm` holds just a slice starting at sample `base_sample` (default s_lo) of the recording (the library
        is handed the address sample 0 would have and never reads outside [s_lo, s_hi))."""
        assert pcm.dtype == np.float32 and pcm.flags.c_contiguous
        self._pcm_keep = pcm
        base = pcm.ctypes.data
        if slice_only:
            b0 = int(s_lo) if base_sample is None else int(base_sample)
            assert b0 <= s_lo and s_hi - b0 <= pcm.shape[0]
            base -= b0 * c * 4
        else:
            assert pcm.shape[0] == n
        check(self.h, self.lib.css_begin_range(self.h, C.c_void_p(base), n, c, C.byref(cfg.c), int(s_lo), int(s_hi)))

    def upload_range(self, pcm: np.ndarray, c: int, s_lo: int, s_hi: int, base_sample: int = 0):
        """Further samples [s_lo, s_hi) of the recording begin_range opened, asynchronously on the copy stream;
        `pcm[0]` is sample `base_sample` of the recording."""
        assert pcm.dtype == np.float32 and pcm.flags.c_contiguous and base_sample <= s_lo and s_hi - base_sample <= pcm.shape[0]
        base = pcm.ctypes.data - int(base_sample) * c * 4
        check(self.h, self.lib.css_upload_range(self.h, C.c_void_p(base), int(s_lo), int(s_hi)))

    def stage_stft(self):
        check(self.h, self.lib.css_stage_stft(self.h))

    def stage_stft_range(self, lo, hi):
        check(self.h, self.lib.css_stage_stft_range(self.h, lo, hi))

    def stage_stitch_masks(self, lo, hi):
        check(self.h, self.lib.css_stage_stitch_masks(self.h, lo, hi))

    def stage_stitch_gate(self, lo, hi):
        check(self.h, self.lib.css_stage_stitch_gate(self.h, lo, hi))

    def stage_istft_partial(self, lo, hi, shard_ptr: int, shard_ld: int):
        check(self.h, self.lib.css_stage_istft_partial(self.h, lo, hi, C.c_void_p(shard_ptr), shard_ld))

    def stage_synthesis(self, lo, hi):
        check(self.h, self.lib.css_stage_synthesis(self.h, lo, hi))

    def stage_seam_rows(self, lo, hi, rows_ptr: int, write: bool):
        check(self.h, self.lib.css_stage_seam_rows(self.h, lo, hi, C.c_void_p(rows_ptr), int(bool(write))))

    def stage_overlap_add(self, f_lo, f_hi, q_lo, q_hi, out_ptr: int, out_ld: int, out_q0: int):
        check(self.h, self.lib.css_stage_overlap_add(self.h, f_lo, f_hi, q_lo, q_hi, C.c_void_p(out_ptr), int(out_ld), int(out_q0)))

    def stage_join_shards(self, gathered_ptr: int, world: int, shard_ld: int, t_lo, t_hi, out_ptr: int, out_ld: int):
        lo = np.ascontiguousarray(t_lo, dtype=np.int64)
        hi = np.ascontiguousarray(t_hi, dtype=np.int64)
        check(self.h, self.lib.css_stage_join_shards(self.h, C.c_void_p(gathered_ptr), int(world), int(shard_ld), _np_ptr(lo), _np_ptr(hi),
                                                     C.c_void_p(out_ptr), int(out_ld)))

    def stage_masknet(self, lo, hi):
        check(self.h, self.lib.css_stage_masknet(self.h, lo, hi))

    def stage_mvdr(self, lo, hi):
        check(self.h, self.lib.css_stage_mvdr(self.h, lo, hi))

    def stage_pit_costs(self, lo, hi):
        check(self.h, self.lib.css_stage_pit_costs(self.h, lo, hi))

    def stage_pit_scan(self):
        check(self.h, self.lib.css_stage_pit_scan(self.h))

    def stage_stitch(self, lo, hi):
        check(self.h, self.lib.css_stage_stitch(self.h, lo, hi))

    def stage_istft(self, lo, hi):
        check(self.h, self.lib.css_stage_istft(self.h, lo, hi))

    def sync(self):
        check(self.h, self.lib.css_sync(self.h))

    # ---- separator protocol helpers
    def stft_host(self, pcm: np.ndarray) -> np.ndarray:
        """pcm [n, C] -> planes [C, 2F, T]."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n, c = pcm.shape
        t = 0 if n < self.desc.frame_len else (n - self.desc.frame_len) // self.desc.frame_hop + 1
        out = np.zeros((c, 2 * self.desc.num_bins, t), dtype=np.float32)
        check(self.h, self.lib.css_stft_host(self.h, _np_ptr(pcm), n, c, _np_ptr(out), t))
        return out

    def separate_host(self, planes: np.ndarray, batch: int) -> np.ndarray:
        """planes [C, 2F, B*T] (B segments back to back) -> masks [(S+1)F, B*T]."""
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        c, f2, tt = planes.shape
        assert c == self.desc.num_mics and f2 == 2 * self.desc.num_bins and tt % batch == 0
        out = np.empty(((self.desc.num_spks + self.desc.num_nois) * self.desc.num_bins, tt), dtype=np.float32)
        check(self.h, self.lib.css_separate_host(self.h, _np_ptr(planes), batch, tt // batch, _np_ptr(out)))
        return out

    def forward_host(self, pcm: np.ndarray) -> np.ndarray:
        """pcm [B, n, C] (equally long clips) -> masks [(S+1)F, B*T'], clip b in columns [b T', (b+1) T')."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        b, n, c = pcm.shape
        t = (n - self.desc.frame_len) // self.desc.frame_hop + 1
        out = np.empty(((self.desc.num_spks + self.desc.num_nois) * self.desc.num_bins, b * max(t, 0)), dtype=np.float32)
        check(self.h, self.lib.css_forward_host(self.h, _np_ptr(pcm), b, n, c, _np_ptr(out)))
        return out

    def handoff_logmel(self, wav_ptr: int, wav_ld: int, stream: int, n_mels: int = 80, pad_frames: int = 8,
                       drop_silence: bool = True, max_regions: int = 4096):
        """After run_device: (log-mel [n_mels, frames] of the stream's active regions, regions [n, 2] sample ranges)."""
        p = self.get_plan()
        cap = int(p.n_out) // 160 + 1
        mel = np.empty((n_mels, cap), dtype=np.float32)
        regions = np.zeros((max_regions, 2), dtype=np.int64)
        nfr, nreg = C.c_int64(), C.c_int32()
        flat = np.empty(n_mels * cap, dtype=np.float32)
        check(self.h, self.lib.css_handoff_logmel(self.h, C.c_void_p(wav_ptr), int(wav_ld), int(stream), int(n_mels), int(pad_frames),
                                                  int(bool(drop_silence)), _np_ptr(flat), cap, C.byref(nfr), _np_ptr(regions),
                                                  max_regions, C.byref(nreg)))
        del mel
        return flat[:n_mels * nfr.value].reshape(n_mels, nfr.value).copy(), regions[:nreg.value].copy()

    def validation_loss(self, mix: np.ndarray, gt_spk0: np.ndarray, gt_noise0: np.ndarray, loss_name: str = "masked_mag",
                        base_loss: str = "mse", clip_gt_to_mixture: bool = False, noise_weight: float = 1.0):
        """train.py:411 _calc_loss for a validation batch: mix [B, n, C], gt_spk0 [B, S, n], gt_noise0 [B, n] (ground
        truths at microphone 0) -> (loss, spk_loss [B], noise_loss [B], perms [B, S])."""
        mix = np.ascontiguousarray(mix, dtype=np.float32)
        gs = np.ascontiguousarray(gt_spk0, dtype=np.float32)
        gn = np.ascontiguousarray(gt_noise0, dtype=np.float32)
        b, n, c = mix.shape
        s = int(self.desc.num_spks)
        assert gs.shape == (b, s, n) and gn.shape == (b, n)
        spk, noi = np.empty(b, np.float32), np.empty(b, np.float32)
        perms = np.empty((b, s), np.int32)
        loss = C.c_float()
        check(self.h, self.lib.css_validation_loss_host(
            self.h, _np_ptr(mix), _np_ptr(gs), _np_ptr(gn), b, n, c, {"masked_mag": 0, "mask": 1}[loss_name],
            {"l1": 0, "mse": 1}[base_loss], int(bool(clip_gt_to_mixture)), float(noise_weight), _np_ptr(spk), _np_ptr(noi),
            _np_ptr(perms), C.byref(loss)))
        return float(loss.value), spk, noi, perms

    def istft_host(self, planes: np.ndarray) -> np.ndarray:
        """planes [B, 2F, T] -> wav [B, (T-1)*hop + frame_len]."""
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        b, _, t = planes.shape
        out = np.empty((b, (t - 1) * self.desc.frame_hop + self.desc.frame_len), dtype=np.float32)
        check(self.h, self.lib.css_istft_host(self.h, _np_ptr(planes), b, t, _np_ptr(out)))
        return out

    # ---- buffers
    def buffer_dims(self, which: int):
        dims = (C.c_int64 * 4)()
        el = C.c_int32()
        check(self.h, self.lib.css_buffer_dims(self.h, which, dims, C.byref(el)))
        shape = [int(x) for x in dims]
        rank = {BUF_X: 3, BUF_SCM: 4, BUF_BFW: 4, BUF_SEP: 4, BUF_MASK_ST: 3, BUF_Y: 3, BUF_WTA_OVERRIDE: 3, BUF_LEVEL: 1}.get(which, 2)
        return tuple(shape[:rank]), int(el.value)

    def read(self, which: int) -> np.ndarray:
        dims, el = self.buffer_dims(which)
        dt = _BUF_DTYPES.get(which, np.float32)
        out = np.empty(dims, dtype=dt)
        assert out.itemsize == el
        check(self.h, self.lib.css_read_buffer(self.h, which, _np_ptr(out), out.nbytes))
        return out

    def write(self, which: int, arr: np.ndarray):
        dims, el = self.buffer_dims(which)
        dt = _BUF_DTYPES.get(which, np.float32)
        arr = np.ascontiguousarray(arr, dtype=dt)
        assert arr.size == int(np.prod(dims)), (arr.shape, dims)
        check(self.h, self.lib.css_write_buffer(self.h, which, _np_ptr(arr), arr.nbytes))

    def devptr(self, which: int) -> int:
        p = C.c_void_p()
        check(self.h, self.lib.css_buffer_devptr(self.h, which, C.byref(p)))
        return int(p.value or 0)

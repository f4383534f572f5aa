"""css-mi355: MI355X-native continuous speech separation (CSS) front end.

Drop-in for the NOTSOFAR baseline's ``css/css.py`` hot path (``css_inference`` /
``separate_and_stitch`` / the ``stft``-``separate``-``istft`` separator protocol), computed by
hand-written gfx950 HIP kernels behind the C ABI declared in ``include/css_mi355.h``.

Submodules are imported lazily so that pure-host helpers (weights, synthetic meetings, sharding
plans, wav I/O) work on a box without a GPU; anything that computes goes through ``_lib`` and fails
loudly when ``libcss_mi355.so`` is missing.
"""
__all__ = ["weights", "synth"]

"""The ISA of gemm_f32.hip -- the only unit with 128-bit buffer stores -- must not hold the pattern round 6 found twice: a buffer
store of more than 64 bits whose `soffset` is an SGPR, directly followed by a VALU write of one of its data registers.  LLVM pads
that slot only when soffset is an immediate (GCNHazardRecognizer::createsVALUHazard); gfx950 needs the wait state either way --
inside this kernel: behind the transposed-accumulator epilogue (F32_TRANSPOSED=1, 42 such stores) and behind the SHIPPED epilogue
as `-O1 -g` schedules it (2 such stores, the sanitizer build of round 6: 28 of 40 launch shapes wrong on the device).  One wait
state glued to every such store (F32_STORE_GUARD, default on) makes both builds write the right bits
(profiles/r06_store_guard.txt).  Here: the shipped build has no such pair, at either optimisation level; and the scanner does
see the pair in both builds with the guard taken out.  CPU only: hipcc cross-compiles to assembly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNIT = os.path.join(ROOT, "notsofar1-challenge_amd", "csrc", "gemm_f32.hip")


def _scan(flags=""):
    env = dict(os.environ, SCAN_FLAGS=flags)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_war.py"), UNIT], capture_output=True, text=True, env=env, timeout=600)
    total = int(out.stdout.strip().splitlines()[-1].split(":")[1])
    return total, out.stdout


def test_shipped_kernel_has_no_store_followed_by_an_overwrite_of_its_data():
    total, text = _scan()
    assert total == 0, text
    assert "wide buffer stores" in text and " 0 wide buffer stores" not in text      # (the scanner saw the epilogue's stores)


def test_no_such_pair_at_the_sanitizer_builds_optimisation_level_either():
    total, text = _scan("-O1 -g")
    assert total == 0, text


def test_scanner_finds_the_pattern_in_the_builds_that_have_it():
    raw, text = _scan("-DF32_TRANSPOSED=1 -DF32_STORE_GUARD=0")
    assert raw > 0 and "soffset: sgpr" in text, text
    o1, text = _scan("-O1 -g -DF32_STORE_GUARD=0")
    assert o1 > 0 and "soffset: sgpr" in text, text
    guarded, _ = _scan("-DF32_TRANSPOSED=1")
    assert guarded == 0

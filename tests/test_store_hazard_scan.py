"""The ISA of gemm_f32.hip -- the only unit with 128-bit buffer stores -- must not hold the pattern round 6 found behind the
transposed-accumulator epilogue: a buffer store of more than 64 bits whose `soffset` is an SGPR, directly followed by a VALU
write of one of its data registers.  LLVM pads that slot only when soffset is an immediate (GCNHazardRecognizer::
createsVALUHazard); gfx950 needs the wait state either way (tools/store_war_probe.hip, profiles/r06_store_war_probe.txt).  The
scanner is checked against the variant that has the pattern (F32_TRANSPOSED=1: 42 such stores, wrong results on the device).
CPU only: hipcc cross-compiles to assembly."""
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


def test_scanner_finds_the_pattern_in_the_variant_that_has_it():
    raw, text = _scan("-DF32_TRANSPOSED=1")
    padded, _ = _scan("-DF32_TRANSPOSED=2")
    assert raw > 0 and "soffset: sgpr" in text, text
    assert padded == 0

#!/usr/bin/env python3
"""Scan gfx950 ISA for the hazard round 6 found behind the transposed-accumulator epilogue of gemm_f32.hip (tools only).

    python tools/scan_store_war.py [file.hip ...]        (default: every unit of notsofar1-challenge_amd/csrc)

A buffer store of more than 64 bits reads its data registers AFTER it has issued; a VALU instruction that writes one of them in
the very next slot corrupts what is stored.  LLVM pads that slot (GCNHazardRecognizer::createsVALUHazard) -- except when the
store's `soffset` operand is an SGPR, where it assumes there is no hazard.  On gfx950 there is, inside gemm_f32.hip's epilogue:
`buffer_store_dwordx4 v[16:19], v64, s[56:59], s8 offen` directly followed by `v_add_f32 v16, ...` made F32_TRANSPOSED=1 write a
few hundred wrong elements per launch, and `buffer_store_dwordx4 v[4:7], v9, s[44:47], s8 offen` -> `v_max_f32 v4, 0, v0` did
the same to the shipped epilogue as -O1 -g schedules it (profiles/r06_store_guard.txt).  The kernel now glues one wait state to
every such store (F32_STORE_GUARD).  This script compiles each unit to assembly and lists every such store whose data registers
are written by the instruction(s) within `WINDOW` slots behind it (s_nop N counts N + 1 slots).  SCAN_FLAGS adds compiler flags
(e.g. "-O1 -g -DF32_STORE_GUARD=0")."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "notsofar1-challenge_amd", "csrc")
WINDOW = 1          # wait states the hardware needs (tools/store_war_probe.hip: 1 is enough)

STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*(v_\w+|buffer_load\w*|global_load\w*|ds_read\w*|ds_load\w*|scratch_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")


def regs_written(line):
    m = VDST.match(line)
    if not m:
        return None
    if m.group(3) is not None:
        return set(range(int(m.group(3)), int(m.group(4)) + 1)), m.group(1)
    return {int(m.group(5))}, m.group(1)


def scan(asm_text, name):
    lines = [l for l in asm_text.splitlines() if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.rstrip().endswith(":")]
    found = []
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        soffset = m.group(5).rstrip(",")
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        slots = 0
        j = i + 1
        while j < len(lines) and slots < WINDOW:
            n = NOP.match(lines[j])
            if n:
                slots += int(n.group(1)) + 1
                j += 1
                continue
            w = regs_written(lines[j])
            # only VALU results land within a slot or two; a load's data arrives hundreds of cycles later
            if w and w[1].startswith("v_") and (w[0] & data):
                found.append((name, l.strip(), lines[j].strip(), "sgpr" if soffset.startswith("s") else "imm", slots))
                break
            slots += 1
            j += 1
    return found


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    extra = os.environ.get("SCAN_FLAGS", "").split()
    total = 0
    for f in files:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + CSRC,
                              "-Wno-unused-value", "-Wno-inline-asm"] + extra + [f, "-o", "-"], capture_output=True, text=True)
        if out.returncode != 0:
            print(f"{os.path.basename(f)}: did not compile\n{out.stderr[-400:]}")
            continue
        hits = scan(out.stdout, os.path.basename(f))
        stores = len([1 for l in out.stdout.splitlines() if STORE.match(l)])
        print(f"{os.path.basename(f):24s} {stores:4d} wide buffer stores, {len(hits):3d} with a VALU write of their data within {WINDOW} slot(s)")
        for h in hits[:6]:
            print(f"      {h[1]}   ->   {h[2]}   (soffset: {h[3]}, {h[4]} slot(s) between)")
        total += len(hits)
    print("total:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())

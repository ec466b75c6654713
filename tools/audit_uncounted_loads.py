#!/usr/bin/env python3
"""Audit of the inline-asm global loads that hipcc does not count (csrc/conv_pw_ares.hip: load_uncounted).

An asm `global_load_dword vN, ...` returns at once; vN holds its data only after the kernel's own asm `s_waitcnt vmcnt(N)`
(whether N is right is the kernel's protocol -- the loads are issued BEFORE the operations the count leaves in flight).
hipcc treats vN as written at the end of the asm statement, so it MAY copy, spill or reuse it before the data has landed
(cdna_hip_programming.md 5.7, item 1) -- silently wrong values.  This script compiles the source to gfx950 assembly and
checks, for every kernel, that in program text no instruction between such a load and the next `s_waitcnt vmcnt(0)`
mentions the destination register (alone or inside a register range), and that every branch in between targets a label
inside that same stretch of text (wave-uniform `if`s around the LDS-DMA requests: both arms are then covered by the scan).

    tools/audit_uncounted_loads.py [file.hip ...]        exit status 1 and a report on a finding
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import List, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "peanut_amd", "csrc")
DEFAULT = [os.path.join(CSRC, "conv_pw_ares.hip"), os.path.join(CSRC, "conv_pw256p.hip")]

LOAD = re.compile(r"^\s*global_load_dword\s+v(\d+)\s*,")
NOP = re.compile(r"^\s*s_nop\b")
WAIT0 = re.compile(r"^\s*s_waitcnt\s+vmcnt\(\d+\)")      # the kernel's own counted wait (inside an asm statement)
RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
SINGLE = re.compile(r"\bv(\d+)\b")
LABEL = re.compile(r"^\.LBB\d+_\d+:")
BRANCH = re.compile(r"^\s*s_(c?branch|setpc|call)")


def assembly(src: str) -> str:
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
               "-S", "--cuda-device-only", "-o", out, src]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(r.stdout.decode(errors="replace"))
        with open(out) as fh:
            return fh.read()


def mentions(line: str, reg: int) -> bool:
    code = line.split(";")[0]
    for a, b in RANGE.findall(code):
        if int(a) <= reg <= int(b):
            return True
    return any(int(n) == reg for n in SINGLE.findall(RANGE.sub(" ", code)))


def audit(text: str) -> Tuple[int, List[str]]:
    """(number of uncounted loads seen, findings)"""
    lines = text.splitlines()
    labels = {}
    for no, line in enumerate(lines, 1):
        m = LABEL.match(line.strip())
        if m:
            labels[line.strip().split(":")[0]] = no
    findings: List[str] = []
    n_loads = 0
    in_asm = False
    pending = {}       # destination register -> line number of its load
    targets: List[Tuple[int, str]] = []
    for no, line in enumerate(lines, 1):
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(".amdhsa_kernel") or s.startswith("s_endpgm"):
            for reg, at in pending.items():
                findings.append(f"line {at}: load into v{reg} is never followed by s_waitcnt vmcnt(0)")
            pending, targets = {}, []
            continue
        if not s or s.startswith(";") or (s.startswith(".") and not LABEL.match(s)):
            continue
        m = LOAD.match(line) if in_asm else None
        if m:
            reg = int(m.group(1))
            if reg in pending:
                findings.append(f"line {no}: v{reg} loaded again before the wait for the load of line {pending[reg]}")
            pending[reg] = no
            n_loads += 1
            continue
        if in_asm and WAIT0.match(line):
            first = min(pending.values()) if pending else no
            for at, tgt in targets:
                where = labels.get(tgt)
                if where is None or not (first < where < no):
                    findings.append(f"line {at}: branch to {tgt} leaves the stretch between an uncounted load (line {first}) and its wait (line {no})")
            pending, targets = {}, []
            continue
        if not pending:
            continue
        if LABEL.match(s):
            continue
        if BRANCH.match(line):
            targets.append((no, s.split()[-1]))
            continue
        for reg, at in pending.items():
            if mentions(line, reg):
                findings.append(f"line {no}: `{s}` touches v{reg} before the wait for its load (line {at})")
    return n_loads, findings


def main(argv: List[str]) -> int:
    rc = 0
    for src in (argv or DEFAULT):
        n, findings = audit(assembly(src))
        print(f"{os.path.basename(src)}: {n} uncounted loads, {len(findings)} findings")
        for f in findings[:40]:
            print("  " + f)
        rc |= bool(findings)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

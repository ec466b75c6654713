#!/usr/bin/env python3
"""Register / scratch / LDS budget of every gfx950 kernel in the built library, read from the code objects' own metadata
notes (what `-Rpass-analysis=kernel-resource-usage` prints at compile time, without recompiling): VGPRs, AGPRs, SGPRs,
spilled VGPRs / SGPRs, private-segment (scratch) bytes, static LDS bytes and the waves per SIMD the VGPR count allows.

    tools/kernel_resources.py [--all] [--json]       default: kernels with spills or scratch, and the MFMA kernels
    tools/kernel_resources.py --packed               packed fp32 instructions of the form that is not safe next to fp16 / bf16 MFMAs

tests/test_abi.py holds the hot kernels to "no spilled VGPR, no scratch" with this table, so that a spill shows up in the CPU
suite and not in somebody's disassembly.
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "peanut_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def _run(cmd: List[str]) -> str:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stderr.decode(errors="replace"))
    return r.stdout.decode(errors="replace")


def waves_per_simd(vgprs: int, agprs: int) -> int:
    """gfx950: 512 unified VGPR/AGPR entries per lane per SIMD, allocated in blocks of 8; at most 8 waves."""
    total = max(vgprs + agprs, 1)
    total = (total + 7) // 8 * 8
    return max(1, min(8, 512 // total))


def object_kernels(obj: str) -> List[Dict]:
    """Kernel descriptors' metadata of one host object (.hip_fatbin section -> gfx950 code object -> AMDGPU metadata note)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        try:
            _run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj])
        except RuntimeError as e:
            if "not found" in str(e):          # a host-only object (csrc/comm.hip: RCCL calls, no kernel)
                return []
            raise
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        _run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"])
        notes = _run([os.path.join(LLVM, "llvm-readelf"), "--notes", co])
    kernels, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"^\s*(- )?\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash and key in ("agpr_count", "args"):       # first key of a kernel entry (.args for kernels with arguments)
            cur = {}
            kernels.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["mangled"] = val.strip()
        elif key in FIELDS:
            cur[key] = int(val)
    kernels = [k for k in kernels if "mangled" in k and "vgpr_count" in k]
    if kernels:
        names = _run(["c++filt"] + [k["mangled"] for k in kernels]).splitlines()
        for k, n in zip(kernels, names):
            k["name"] = re.sub(r"\(anonymous namespace\)::", "", n).replace("peanut::", "")
            k["name"] = re.sub(r"\(peanut::.*\)$|\(ConvKParams\)$", "", k["name"])
            k["file"] = os.path.basename(obj)[:-2] + ".hip"
            k["waves_per_simd"] = waves_per_simd(k["vgpr_count"], k.get("agpr_count", 0))
    return kernels


def object_disassembly(obj: str) -> str:
    """llvm-objdump -d of the gfx950 code object inside one host object ('' for a host-only object)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        try:
            _run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj])
        except RuntimeError as e:
            if "not found" in str(e):
                return ""
            raise
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""
        _run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"])
        return _run([os.path.join(LLVM, "llvm-objdump"), "-d", co])


_PK_F32 = re.compile(r"\b(v_pk_\w+_f32)\b")
_OP_SEL = re.compile(r"op_sel:\[([01,]+)\]")


def risky_packed_fp32(build_dir: str = BUILD) -> List[Dict]:
    """Packed fp32 instructions whose LOW half takes an operand from the HIGH register of a pair (`op_sel:[..1..]`), per kernel.
    On gfx950 that form returns wrong low halves while a wave on the same SIMD issues fp16 / bf16 MFMAs (csrc/common.h,
    PEANUT_NO_PK_F32; profiles/r9r): the library must not contain it.  `op_sel_hi` alone (the high half reading a low register) and
    unmodified packed instructions were measured exact and are not reported."""
    out = []
    for f in sorted(os.listdir(build_dir)):
        if not f.endswith(".o"):
            continue
        cur = None
        for ln in object_disassembly(os.path.join(build_dir, f)).splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
            if m:
                cur = m.group(1)
                continue
            m = _PK_F32.search(ln)
            if cur and m:
                sel = _OP_SEL.search(ln)
                if sel and "1" in sel.group(1):
                    out.append({"file": f[:-2] + ".hip", "mangled": cur, "instruction": ln.split("//")[0].strip()})
    if out:
        names = _run(["c++filt"] + [r["mangled"] for r in out]).splitlines()
        for r, n in zip(out, names):
            r["name"] = re.sub(r"\(anonymous namespace\)::", "", n).replace("peanut::", "")
    return out


def library_kernels(build_dir: str = BUILD) -> List[Dict]:
    out = []
    for f in sorted(os.listdir(build_dir)):
        if f.endswith(".o"):
            out += object_kernels(os.path.join(build_dir, f))
    return out


def main():
    if "--packed" in sys.argv:
        rows = risky_packed_fp32()
        for r in rows:
            print(f"{r['file']:18s} {r['instruction']:70s} {r['name'][:80]}")
        print(f"{len(rows)} packed fp32 instructions with op_sel (low half from a high register)")
        return
    ks = library_kernels()
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
        return
    show_all = "--all" in sys.argv
    print(f"{'file':18s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'w/simd':>6s}  kernel")
    for k in ks:
        flagged = k["vgpr_spill_count"] or k["private_segment_fixed_size"] or k["sgpr_spill_count"]
        if show_all or flagged or k["vgpr_count"] >= 128:
            print(f"{k['file']:18s} {k['vgpr_count']:4d} {k.get('agpr_count', 0):4d} {k['sgpr_count']:4d} {k['vgpr_spill_count']:6d} "
                  f"{k['sgpr_spill_count']:6d} {k['private_segment_fixed_size']:7d} {k.get('group_segment_fixed_size', 0):6d} "
                  f"{k['waves_per_simd']:6d}  {k['name']}")
    print(f"{len(ks)} kernels; {sum(1 for k in ks if k['vgpr_spill_count'])} with spilled VGPRs, "
          f"{sum(1 for k in ks if k['private_segment_fixed_size'])} with scratch, {sum(1 for k in ks if k['sgpr_spill_count'])} with spilled SGPRs")


if __name__ == "__main__":
    main()

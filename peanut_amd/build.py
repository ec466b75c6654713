"""In-tree build of libpeanut_hip.so (explicit hipcc, gfx950 only; no JIT cache, no cmake).

The shared object is written next to this file so that it travels with the repo snapshot to the
GPU box (``*.so`` is git-ignored but not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "libpeanut_hip.so")
ARCH = "gfx950"


def sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps() -> List[str]:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return sources() + hdrs + [os.path.abspath(__file__)]


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the names and contents of every source and header the library is built from.
    Compiled into the library (peanut_source_hash()): the binding refuses -- or rebuilds -- a library whose sources have
    changed since it was built, whatever the file times say (a repo snapshot copied to another box keeps no useful
    mtimes)."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _built_hash() -> str:
    try:
        with open(LIB_PATH + ".srchash") as fh:
            return fh.read().strip()
    except OSError:
        return ""


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if _built_hash() != source_hash():
        return True
    t = os.path.getmtime(LIB_PATH)
    return os.path.getmtime(os.path.abspath(__file__)) > t


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source for gfx950 and link libpeanut_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    src_hash = source_hash()
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
              "-Wall", "-Wno-unused-function"]
    stamp = os.path.join(objdir, "srchash")          # objects are reused only while they were built from these very sources
    try:
        with open(stamp) as fh:
            reuse = fh.read().strip() == src_hash
    except OSError:
        reuse = False
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not reuse or not os.path.exists(obj):
            cmd = common + ([f'-DPEANUT_SOURCE_HASH="{src_hash}"'] if os.path.basename(src) == "pred_api.hip" else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    tmp = LIB_PATH + ".tmp"
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    os.replace(tmp, LIB_PATH)
    for path in (stamp, LIB_PATH + ".srchash"):
        with open(path, "w") as fh:
            fh.write(src_hash + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

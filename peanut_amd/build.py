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


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


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
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
              "-Wall", "-Wno-unused-function"]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(p) > os.path.getmtime(obj) for p in _deps()):
            cmd = common + ["-c", src, "-o", obj]
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
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

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
    """Compile every .hip source for gfx950 and link libpeanut_hip.so; returns its path.

    Safe against concurrent callers (torchrun ranks importing after a source edit): the whole build runs under an
    exclusive ``flock`` on ``build/.lock``; a caller that had to wait re-checks staleness once it holds the lock and
    returns the library the first one produced.  Objects carry a stamp of the sources they were compiled from PER OBJECT
    (the stamp is removed before a compile starts and written only after it succeeded), the shared object is linked
    under a per-process temporary name and moved into place atomically."""
    import fcntl
    if not force and not is_stale():
        return LIB_PATH
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():          # another process built it while this one waited
                return LIB_PATH
            return _build_locked(objdir, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _file_hash(paths: List[str]) -> str:
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _build_locked(objdir: str, force: bool, verbose: bool) -> str:
    cc = hipcc()
    src_hash = source_hash()
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
              "-Wall", "-Wno-unused-function"]
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        stamp = obj + ".srchash"
        objs.append(obj)
        embeds_hash = os.path.basename(src) == "pred_api.hip"     # carries PEANUT_SOURCE_HASH: depends on every source
        want = src_hash if embeds_hash else _file_hash([src] + headers + [os.path.abspath(__file__)])
        try:
            with open(stamp) as fh:
                have = fh.read().strip()
        except OSError:
            have = ""
        if force or have != want or not os.path.exists(obj):
            if os.path.exists(stamp):
                os.remove(stamp)                      # a failed or interrupted compile must not leave a valid stamp
            cmd = common + ([f'-DPEANUT_SOURCE_HASH="{src_hash}"'] if embeds_hash else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, stamp, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, stamp, want, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
            continue
        with open(stamp, "w") as fh:
            fh.write(want + "\n")
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if failed:
        raise RuntimeError("\n".join(failed))
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".srchash", "w") as fh:
        fh.write(src_hash + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

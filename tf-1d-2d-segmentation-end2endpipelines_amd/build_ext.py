"""Builds libseg_hip.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

Every csrc/*.hip is its own translation unit: the units compile in parallel (the two big ones take about a minute each)
and only the ones whose source, a header or the compiler flags changed (by content hash, kept beside each object) are
rebuilt; the objects live in csrc/_build/ (git-ignored)."""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "_build")
OUT = os.path.join(HERE, "libseg_hip.so")
STAMP = OUT + ".srchash"          # fingerprint of the sources the shipped .so was built from
BASE_CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
ABI_HEADER = os.path.join(ROOT, "include", "seg_hip.h")


def abi_version() -> int:
    """The C ABI's version = 28 bits of the content hash of include/seg_hip.h: every signature or struct change is a new version
    without anybody remembering to bump a number.  Compiled into the library (-DSEG_ABI_VERSION, returned by seg_abi_version())
    and recomputed from the header by _ffi.lib(), which refuses a library built against another header (a SEG_LIB=<older build>
    used for an A/B would otherwise resolve every symbol and read shifted arguments)."""
    with open(ABI_HEADER, "rb") as fh:
        return int(hashlib.sha256(fh.read()).hexdigest()[:7], 16)


def _cflags():
    return BASE_CFLAGS + [f"-DSEG_ABI_VERSION={abi_version()}"]


CFLAGS = _cflags()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "seg_hip.h")]


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _obj_key(src, hdrs) -> str:
    """What an object file depends on: its source, every header and the compiler flags, by content (a flags-only change or a
    copy that does not keep modification times must recompile, an unchanged unit must not)."""
    h = hashlib.sha256()
    for f in [src] + list(hdrs):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_cflags()).encode())
    return h.hexdigest()


def _stale(obj, key) -> bool:
    stamp = obj + ".key"
    if not (os.path.exists(obj) and os.path.exists(stamp)):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != key


def _fingerprint() -> str:
    """Content hash of every source and header (the snapshot copy to the GPU box does not keep modification times)."""
    h = hashlib.sha256()
    for f in sources() + headers():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_cflags()).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _fingerprint()


def build(force: bool = False, verbose: bool = True) -> str:
    srcs, hdrs = sources(), headers()
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJDIR, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    jobs = []
    for s in srcs:
        key = _obj_key(s, hdrs)
        if force or _stale(_obj(s), key):
            if os.path.exists(_obj(s) + ".key"):
                os.remove(_obj(s) + ".key")
            cmd = [hipcc] + _cflags() + inc + ["-c", s, "-o", _obj(s)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((s, key, subprocess.Popen(cmd)))
    failed = []
    for s, key, p in jobs:
        if p.wait() != 0:
            failed.append(s)
        else:
            with open(_obj(s) + ".key", "w") as fh:
                fh.write(key + "\n")
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(os.path.basename(f) for f in failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in srcs] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(_fingerprint() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Builds libseg_hip.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

Every csrc/*.hip is its own translation unit: the units compile in parallel (the two big ones take about a minute each)
and only the ones whose source - or a header - changed are rebuilt; the objects live in csrc/_build/ (git-ignored)."""
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
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "seg_hip.h")]


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _fingerprint() -> str:
    """Content hash of every source and header (the snapshot copy to the GPU box does not keep modification times)."""
    h = hashlib.sha256()
    for f in sources() + headers():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CFLAGS).encode())
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
        if force or _stale(_obj(s), [s] + hdrs):
            cmd = [hipcc] + CFLAGS + inc + ["-c", s, "-o", _obj(s)]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in jobs if p.wait() != 0]
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

"""Builds libseg_hip.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

Every csrc/*.hip is its own translation unit: the units compile in parallel (the two big ones take about a minute each)
and only the ones whose source, a header or the compiler flags changed (by content hash, kept beside each object) are
rebuilt; the objects live in csrc/_build/ (git-ignored)."""
import glob
import hashlib
import os
import re
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


ABI_STAMP = os.path.join(HERE, "_abi.py")   # written by build(): the ABI version the shipped library was compiled against


def _abi_text(src: str) -> str:
    """The header without comments and with white space collapsed: an edit to a comment is not a new ABI."""
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    return " ".join(src.split())


def abi_version() -> int:
    """The C ABI's version = 28 bits of the content hash of include/seg_hip.h (comments and white space excluded): every signature or
    struct change is a new version without anybody remembering to bump a number.  Compiled into the library (-DSEG_ABI_VERSION,
    returned by seg_abi_version()) and recomputed from the header by _ffi.lib(), which refuses a library built against another
    header (a SEG_LIB=<older build> used for an A/B would otherwise resolve every symbol and read shifted arguments).  A deployment
    that ships only the package (no include/) compares against the version build() recorded in _abi.py instead."""
    if not os.path.exists(ABI_HEADER):
        return recorded_abi_version()
    with open(ABI_HEADER, "r", encoding="utf-8") as fh:
        return int(hashlib.sha256(_abi_text(fh.read()).encode()).hexdigest()[:7], 16)


def recorded_abi_version() -> int:
    if not os.path.exists(ABI_STAMP):
        raise ImportError(f"cannot establish the C ABI version: neither {ABI_HEADER} (the header the version is derived from) nor "
                          f"{ABI_STAMP} (written by the build) exists; ship include/seg_hip.h or rebuild with "
                          "`python -c 'import __graft_entry__ as g; g.build()'`")
    ns = {}
    with open(ABI_STAMP) as fh:
        exec(fh.read(), ns)
    return int(ns["ABI_VERSION"])


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
    if not os.path.exists(OUT) or not os.path.exists(STAMP) or not os.path.exists(ABI_STAMP):
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
    with open(ABI_STAMP, "w") as fh:
        fh.write(f"# generated by build_ext.build(): the ABI version libseg_hip.so was compiled against\nABI_VERSION = {abi_version()}\n")
    return OUT


def build_variant(out: str, extra_cflags) -> str:
    """An alternative build for same-box A/Bs (tools/ab_bench.sh, SEG_LIB=<out>): every unit compiled with extra flags into its
    own object directory beside `out`; the in-tree library and its objects are not touched."""
    out = os.path.abspath(out)
    objdir = out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs, jobs = [], []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        jobs.append((s, subprocess.Popen([hipcc] + _cflags() + list(extra_cflags) + inc + ["-c", s, "-o", o])))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(os.path.basename(f) for f in failed))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python build_ext.py --variant ab/name.so -DFLAG ...
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(force="--force" in sys.argv)

"""Builds libseg_hip.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = ["igemm.hip", "elementwise.hip"]
OUT = os.path.join(HERE, "libseg_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, "csrc", f) for f in SRCS + ["seg_common.h"]] + [os.path.join(ROOT, "include", "seg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")] + \
          [os.path.join(HERE, "csrc", f) for f in SRCS] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

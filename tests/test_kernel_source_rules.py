"""Source-level rules of the HIP kernels that no functional test sees (CPU).

Rule 1 - ring slots are released only after the releasing wave's own LDS reads have RETURNED.  Every raw `s_barrier` of the
ring kernels hands an LDS slot back to whoever refills it; gfx950 puts no implicit wait in front of `s_barrier` and the
compiler may leave the last MFMAs of a step (with the wait for their fragment reads) behind it.  Without an explicit
`s_waitcnt lgkmcnt(0)` in front of the barrier a queued read can return the next step's bytes when the LDS is backed up by
a co-resident kernel - one 32 x 32 tile a K step off in 1-10 % of the C3 training steps until the end of round 2
(DESIGN.md section 3).  Barriers of pure loader waves are tagged in the source.

Rule 2 - the library is built without compiler-formed packed fp32 arithmetic (`-fno-slp-vectorize`): the BN sum of squares
of the conv epilogue went wrong through it."""
import glob
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "tf-1d-2d-segmentation-end2endpipelines_amd"
CSRC = os.path.join(ROOT, PKG, "csrc")


def test_every_raw_barrier_waits_for_the_waves_own_lds_reads():
    bad = []
    n = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        lines = open(path).read().splitlines()
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            if "__builtin_amdgcn_s_barrier()" not in code:
                continue
            n += 1
            if "loader waves: no LDS reads" in line:
                continue
            before = "\n".join(l.split("//")[0] for l in lines[max(0, i - 8):i])
            if not re.search(r"s_waitcnt[^\"]*lgkmcnt\(0\)", before):
                bad.append(f"{os.path.basename(path)}:{i + 1}")
    assert n >= 8, "the ring kernels' barriers were not found - has the idiom been renamed?"
    assert not bad, f"s_barrier without a preceding s_waitcnt lgkmcnt(0): {bad}"


def test_library_is_built_without_slp_vectorisation():
    be = importlib.import_module(PKG + ".build_ext")
    assert "-fno-slp-vectorize" in be.CFLAGS

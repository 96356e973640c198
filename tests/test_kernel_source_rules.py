"""Source-level rules of the HIP kernels that no functional test sees (CPU).

Rule 1 - ring slots are released only after the releasing wave's own LDS reads have RETURNED.  Every barrier of the ring
kernels hands an LDS slot back to whoever refills it; gfx950 puts no implicit wait in front of `s_barrier` and the
compiler may leave the last MFMAs of a step (with the wait for their fragment reads) behind it.  Without an explicit
`s_waitcnt lgkmcnt(0)` in front of the barrier a queued read can return the next step's bytes when the LDS is backed up by
a co-resident kernel - one 32 x 32 tile a K step off in 1-10 % of the C3 training steps until the end of round 2
(docs/experiments.md section 3).  The wait and the barrier are ONE helper (igemm_common.h: ring_release_barrier); pure loader waves
have their own (loader_barrier).

Rule 2 - the library is built without compiler-formed packed fp32 arithmetic (`-fno-slp-vectorize`): the BN sum of squares
of the conv epilogue went wrong through it (docs/experiments.md section 3)."""
import glob
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "tf-1d-2d-segmentation-end2endpipelines_amd"
CSRC = os.path.join(ROOT, PKG, "csrc")


def test_every_raw_barrier_waits_for_the_waves_own_lds_reads():
    """`s_barrier` is executed in exactly two helpers of igemm_common.h: ring_release_barrier() (wait for the wave's own LDS reads,
    then the barrier) and loader_barrier() (waves that read nothing from LDS).  No kernel spells the barrier out itself, so the
    wait cannot be separated from it by an edit; the ring kernels all go through the helper."""
    raw, users = [], 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        text = open(path).read()
        code = "\n".join(l.split("//")[0] for l in text.splitlines())
        users += len(re.findall(r"\bring_release_barrier\(", code))
        for m in re.finditer(r"__builtin_amdgcn_s_barrier\(\)|\bs_barrier\b", code):
            raw.append((os.path.basename(path), code[:m.start()].count("\n") + 1))
    assert [f for f, _l in raw] == ["igemm_common.h", "igemm_common.h"], raw
    common = open(os.path.join(CSRC, "igemm_common.h")).read()
    body = common[common.index("void ring_release_barrier("):common.index("void loader_barrier(")]
    assert body.index("s_waitcnt lgkmcnt(0)") < body.index("__builtin_amdgcn_s_barrier()")
    assert users >= 9, "the ring kernels no longer go through ring_release_barrier()"      # (1 definition + 8 kernels)


def test_library_is_built_without_slp_vectorisation():
    be = importlib.import_module(PKG + ".build_ext")
    assert "-fno-slp-vectorize" in be.CFLAGS

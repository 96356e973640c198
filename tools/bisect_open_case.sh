#!/bin/bash
# The open round-5 finding (DESIGN section 3 "R5"; tests/test_gpu_fuzz.py::test_the_open_big_case_of_round_five): case 129 of
# `fuzz_models.py --big --seed 21` against the oracle with the engine's decisions imposed, once per switch - a minute each on an MI355X.
# The line to read is "grad(arith)": 2.6e-03 = the finding, ~3e-04 = gone.  FUZZ_WATCH lists the two sums that must be EQUAL
# (sc/bn/beta and bn_cat/beta of a MultiResBlock are both the per-channel sum of the add's gradient) next to the worst eight.
#     bash tools/bisect_open_case.sh > gpurun_out/bisect_open_case.log 2>&1
cd "$(dirname "$0")/.."
for v in "X=1" "SEG_TAIL_LAZY_INPUTS=0" "SEG_TAIL_SHARED_GRAD=0" "SEG_TAIL_VIRTUAL_CAT=0" "SEG_TAIL_LAZY_SUMS=0" "SEG_TAIL_PIECE_STATS=0" \
         "SEG_TAIL_PAIR_APPLY=0" "SEG_FWD_HALO=0" "SEG_WGRAD_HALO=0" "SEG_FWD_SPLITK=0" "SEG_FUSE_ACT_BWD=0"; do
  echo "== $v"
  env $v FUZZ_WATCH="enc2/mrb/sc/bn/beta,enc2/mrb/bn_cat/beta,enc1/mrb/sc/bn/beta,enc1/mrb/bn_cat/beta" timeout 240 \
    python tools/fuzz_models.py --seed 21 --big --n 400 --cases 129 2>&1 | grep -E "^      out|conditioning" | cut -c1-190
done

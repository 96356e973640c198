#!/usr/bin/env python3
"""Engine against engine on the open round-5 case (DESIGN section 3 'R5': MultiResUNet 256 x 192, W = 32, alpha = 1.67, 9-class softmax,
batch 8): parameter gradients of the model built with ONE switch of flags.py turned off at a time, against the model with the
BN(ReLU(a + b)) tails unfused (which the oracle confirms to 3e-4) - same weights, same batch, no oracle run, seconds per variant.
CAVEAT, learnt on the first run: every build takes its OWN ReLU / pool decisions, and two builds whose forward values differ in the
last bits decide a few hundred boundary elements differently - at this size that alone moves small-norm gradients by 3e-3 .. 1e-2
(default build against unfused: 7.1e-3; TAIL_SHARED_GRAD or TAIL_VIRTUAL_CAT off: 3.0e-3; FWD_HALO off: 1.3e-2), the size of the
effect being looked for.  Engine against engine is only conclusive with the decisions of one build imposed on the other, which the
engine cannot do; the oracle can (`tools/fuzz_models.py --big --seed 21 --cases 129` under one `SEG_*=0` at a time, a minute each).

    python tools/tail_bisect.py

Build-authored test tooling; needs an MI355X."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import segamd
from segamd import flags as FL

VARIANTS = [(), ("TAIL_LAZY_INPUTS",), ("TAIL_SHARED_GRAD",), ("TAIL_VIRTUAL_CAT",), ("TAIL_LAZY_SUMS",), ("TAIL_PIECE_STATS",), ("TAIL_PAIR_APPLY",),
            ("FWD_HALO",), ("WGRAD_HALO",), ("FUSE_ACT_BWD",), ("HEAD_FUSE_FWD", "HEAD_INJECT"), ("FWD_SPLITK",)]


def grads(off, weights, x, t):
    old = {k: getattr(FL, k) for k in off}
    for k in off:
        setattr(FL, k, False)
    try:
        m = segamd.unet_model_builder('MultiResUNet', 256, 192, 32, 2, num_channels=3, output_nums=9, ds=0, ae=0, ag=0, lstm=0, dense_loop=1,
                                      feature_number=50, is_transconv=False, alpha=1.67, final_activation='softmax', train_mode="from_scratch").ResNet50()
        m.build(8, dtype="f32")
    finally:
        for k, v in old.items():
            setattr(FL, k, v)
    if weights is None:
        weights = m.get_weights()
    else:
        m.set_weights(weights)
    m.forward(x, training=True)
    m.compute_loss([t], "cce")
    m.backward(None)
    g = m.get_grads()
    torch.cuda.synchronize()
    return g, weights


def main():
    torch.manual_seed(21)
    x = torch.rand(8, 256, 192, 3)
    t = torch.nn.functional.one_hot(torch.randint(0, 9, (8, 256, 192)), 9).float()
    ref, w = grads(("FUSE_BN_TAILS",), None, x, t)
    gn = max(float(np.linalg.norm(v.ravel())) for v in ref.values())
    for off in VARIANTS:
        g, _ = grads(off, w, x, t)
        errs = sorted(((float(np.linalg.norm((g[k] - ref[k]).ravel())) / (float(np.linalg.norm(ref[k].ravel())) + 1e-3 * gn), k) for k in ref), reverse=True)
        print(f"off: {','.join(off) or '(nothing: the default build)':40s} worst {errs[0][0]:.1e} {errs[0][1]:32s} then {errs[1][0]:.1e} {errs[1][1]:28s} {errs[2][0]:.1e} {errs[2][1]}", flush=True)


if __name__ == "__main__":
    main()

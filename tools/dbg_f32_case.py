"""Debug aid: per-tensor errors of one parity-mode model case against the oracle (tests/gpu_util.compare_model)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import gpu_util as U
from test_gpu_models import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "unet_w64_bce"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
torch.manual_seed(7)
model, x, targets, loss = CASES[name]()
model.build(x.shape[0], dtype=dtype)
r = U.compare_model(model, x, targets, loss, emulate_fp16=False)
print("outs", r["outs"], "loss", r["loss"])
worst = sorted(((e / max(1.0, m), n) for n, e, m in r["nodes"]), reverse=True)[:8]
print("worst nodes", worst)
worst = sorted(((e / (m + 1e-30), n, m) for n, e, m in r["agrads"]), reverse=True)[:8]
print("worst agrads", worst)
gn = max(n for _e, n in r["gl2"].values())
for k, (e, n) in sorted(r["gl2"].items(), key=lambda kv: -kv[1][0] / (kv[1][1] + 1e-3 * gn))[:10]:
    print(f"gl2 {k:40s} err {e:.3e} norm {n:.3e} rel {e / (n + 1e-30):.3e}  cos {r['gcos'][k][0]:.8f}")
# how concentrated is the activation-gradient error?  (a ReLU whose input is within the forward error of zero flips: that
# element's gradient is off by 100 %, everything downstream of it by a little)
S = r["S"]
vals = U.oracle_run(model.graph, model.get_weights(), x, targets, loss)["vals"]
for op in model.ops:
    if isinstance(op, U.M.ConvOp) and getattr(op, "live", False) and op.node.name in vals and vals[op.node.name].grad is not None:
        ref = vals[op.node.name].grad
        got = op.out.grad_values().cpu() / S
        d = (got - ref).abs()
        mx = float(ref.abs().max())
        big = int((d > 0.01 * mx).sum())
        print(f"{op.node.name:24s} n {ref.numel():9d} elements off by > 1% of max: {big:6d}  L2 rel {float(d.norm() / ref.norm()):.2e}  "
              f"L2 rel without them {float((d * (d <= 0.01 * mx)).norm() / ref.norm()):.2e}")

#!/usr/bin/env python3
"""Where is an activation gradient off?  Builds one builder call (parity mode, SEG_BACKWARD as set), runs forward + loss + backward on
the engine and on the oracle with the engine's decisions imposed, and lists for every activation-gradient tensor how many elements are
off by more than 1e-3 of the tensor's maximum and where they sit (image, row, column, channel histograms) - the view that tells a
border / padding-channel / single-image defect from rounding.  Build-authored analysis tooling (uses the oracle).

    python tools/dbg_agrad.py "UNet" 128 80 64 3 --kw "num_channels=5,output_nums=2,ds=1,lstm=1,dense_loop=2,is_transconv=False,final_activation='softmax'" --batch 2 --loss logcosh
"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import segamd
import gpu_util as U
import fuzz_models as FZ
import random

ap = argparse.ArgumentParser()
ap.add_argument("dec", nargs="?"); ap.add_argument("H", type=int, nargs="?"); ap.add_argument("W", type=int, nargs="?")
ap.add_argument("width", type=int, nargs="?"); ap.add_argument("depth", type=int, nargs="?")
ap.add_argument("--kw", default=""); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--loss", default="mse"); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--fuzz-case", default="", help="SEED:INDEX[:big] - exactly that case of tools/fuzz_models.py (same weights, input, targets)")
a = ap.parse_args()
if a.fuzz_case:
    parts = a.fuzz_case.split(":")
    seed, index = int(parts[0]), int(parts[1])
    FZ.BIG = len(parts) > 2 and parts[2] == "big"
    rng = random.Random(seed)
    for i in range(index + 1):
        two_d = True if FZ.BIG else rng.random() < 0.65
        call, make, xf, loss_name = (FZ.sample_2d if two_d else FZ.sample_1d)(rng)
    print(call)
    torch.manual_seed(seed * 100003 + index)
    m, x = make(), xf()
    m.build(x.shape[0], dtype="f32")
    a.loss = loss_name
    tg = FZ.targets_for(m, x, a.loss, rng)
else:
    kw = eval("dict(" + a.kw + ")")
    torch.manual_seed(a.seed)
    m = segamd.unet_model_builder(a.dec, a.H, a.W, a.width, a.depth, train_mode="from_scratch", **kw).ResNet50()
    x = torch.rand(a.batch, a.H, a.W, kw.get("num_channels", 3))
    m.build(a.batch, dtype="f32")
    tg = FZ.targets_for(m, x, a.loss, random.Random(a.seed))
r = U.compare_model(m, x, tg, a.loss, impose_decisions=True)
orb, _dec = r["imposed_oracle"]
vals = orb["vals"]
S = r["S"]
print("outs", r["outs"], "loss", r["loss"], "backward", m.backward_spec, "loss scale S", S, "grad_target", m.grad_target)
for h, t in zip(m.out_heads, tg):
    print(f"head {h.node.name}: output {tuple(h.y.shape)} min {float(h.y.min()):.3e} max {float(h.y.max()):.3e}; target unique {torch.unique(t)[:6].tolist()}")
for op in m.ops:
    t = getattr(op, "out", None)
    if t is None or t.grad is None or t.name not in vals or vals[t.name].grad is None:
        continue
    ref = vals[op.node.name].grad if hasattr(op, "node") and op.node.name in vals and vals[op.node.name].grad is not None else vals[t.name].grad
    got = t.grad_values().cpu() / S
    if got.shape != ref.shape:
        continue
    err = (got - ref).abs()
    mx = float(ref.abs().max())
    bad = err > 1e-3 * mx
    nb = int(bad.sum())
    line = f"{'d/' + t.name:28s} max|ref| {mx:.3e} worst {float(err.max()) / mx:.2e} off {nb} of {bad.numel()}"
    if nb:
        idx = bad.nonzero()
        line += f" | images {sorted(set(idx[:, 0].tolist()))} rows {int(idx[:, 1].min())}..{int(idx[:, 1].max())} cols {int(idx[:, 2].min())}..{int(idx[:, 2].max())} channels {sorted(set(idx[:, 3].tolist()))[:12]}"
        j = int(err.flatten().argmax()); pos = np.unravel_index(j, tuple(err.shape))
        line += f" | worst at {tuple(int(v) for v in pos)}: got {float(got[pos]):.4e} ref {float(ref[pos]):.4e}"
    print(line)
# ConvLSTM gates: pre-activations sitting on a hard-sigmoid knee (|z| within 1e-6 of 2.5) - where the engine's rule (slope iff |z| <= 2.5)
# and a harness that re-derives the decision from 0.2 z + 0.5 in fp32 can disagree
for name, t in m.tensors.items():
    if name.endswith(":z") and (name[:-2] + "") in vals and m.graph.nodes[[n.name for n in m.graph.nodes].index(name[:-2])].kind == "convlstm":
        z = t.values().cpu()
        zi, _zf, _zc, zo = torch.chunk(z, 4, dim=-1)
        for gname, v in (("i", zi), ("o", zo)):
            near = ((v.abs() - 2.5).abs() < 1e-6).nonzero()
            for pos in near[:8]:
                val = float(v[tuple(pos)])
                f32 = torch.tensor(val, dtype=torch.float32)
                hs = float(0.2 * f32 + 0.5)
                print(f"{name} gate {gname} at {tuple(int(p) for p in pos)}: z = {val!r}, fp32 0.2 z + 0.5 = {hs!r}, engine slope {abs(val) <= 2.5}, harness low/high knee {hs <= 0.0}/{hs >= 1.0}")

#!/usr/bin/env python3
"""What two MFMAs per product in the BACKWARD GEMMs would cost in gradient accuracy (docs/experiments.md section 7, item 0), emulated in the oracle
on the CPU: the data gradient sees the incoming gradient rounded to fp16 in the loss-scaled domain (its hi half), the weight gradient
the activations rounded to fp16 (or, second variant, the incoming gradient again); everything else fp32.  Prints the worst and the
median parameter-gradient error (relative L2 with the fuzzer's floor of 1e-3 of the largest gradient norm) for three small models at
loss-scale targets 32 and 1 024.  Round 5's run: UNet++ worst 5.4e-4 / median 2.8e-5; MultiResUNet 9-class 2.6e-3 .. 3.8e-3 / 3.9e-4;
UNet with attention gates and ConvLSTM 9e-4 / 4e-4 - the size of TF32's rounding, two orders above the three-MFMA mode's 8e-6.

    python tools/emulate_2mfma_backward.py                  # CPU only, about a minute
    python tools/emulate_2mfma_backward.py --trajectory     # Adam steps on one batch: how far predict() moves (chaos: at 20 steps the
                                                            # 22-bit backward pass is as far from the exact one as the 11-bit one)

Build-authored analysis tooling (uses the oracle: not part of the product path)."""
import sys, numpy as np, torch, torch.nn.functional as F
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import segamd
from oracle import keras_ref as K
from oracle.keras_ref import init_param
from ir_interp import run_graph
torch.set_num_threads(8)

MODE = {"on": False, "dz_only": False}
R = {"bits": 11}


def r16(t):
    """The operand at 11 significant bits (its fp16 hi half) or, for comparison, perturbed at 2^-bits (22: the split pair's own precision)."""
    return t.half().float() if R["bits"] == 11 else t * (1.0 + 2.0 ** -R["bits"] * torch.randn_like(t))

class Conv2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xin, w, bias, stride, transposed):
        ctx.stride, ctx.transposed = stride, transposed
        ctx.save_for_backward(xin, w, bias)
        with torch.backends.mkldnn.flags(enabled=False):
            return (F.conv_transpose2d if transposed else F.conv2d)(xin, w, bias, stride=stride)
    @staticmethod
    def backward(ctx, gy):
        xin, w, bias = ctx.saved_tensors
        f = F.conv_transpose2d if ctx.transposed else F.conv2d
        gy = gy.contiguous()
        with torch.enable_grad(), torch.backends.mkldnn.flags(enabled=False):
            xa = xin.detach().requires_grad_(True)
            gx = torch.autograd.grad(f(xa, w.detach(), None, stride=ctx.stride), xa, r16(gy) if MODE["on"] else gy)[0] if ctx.needs_input_grad[0] else None
            wa = w.detach().requires_grad_(True)
            xw = xin.detach() if (not MODE["on"] or MODE["dz_only"]) else r16(xin.detach())
            gyw = r16(gy) if (MODE["on"] and MODE["dz_only"]) else gy
            gw = torch.autograd.grad(f(xw, wa, None, stride=ctx.stride), wa, gyw)[0]
        gb = gy.sum((0, 2, 3)) if bias is not None else None
        return gx, gw, gb, None, None

def conv(xin, w, bias, stride, transposed=False):
    return Conv2.apply(xin.contiguous(), w.contiguous(), bias, tuple(stride), transposed)
K._conv = conv

def grads(g, W, x, tgt, loss, scale):
    w = {k: torch.from_numpy(np.asarray(v)).requires_grad_(not (k.endswith("moving_mean") or k.endswith("moving_variance"))) for k, v in W.items()}
    outs = run_graph(g, x, w, training=True)
    if loss == "mse":
        L = sum(((o - t) ** 2).mean() for o, t in zip(outs, tgt))
    else:
        L = sum(-(t * torch.log(o.clamp_min(1e-7))).sum(-1).mean() for o, t in zip(outs, tgt))
    (L * scale).backward()
    return {k: v.grad.double() / scale for k, v in w.items() if v.grad is not None}, outs

def gradients():
    cases = [
        ("UNet++ d3 w16 64x64 bs4 +DS, mse", segamd.unet_model_builder("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1, train_mode="from_scratch").graph(), "mse"),
        ("MultiResUNet d2 w16 alpha 1.67 64x64 bs4, 9-class cce", segamd.unet_model_builder("MultiResUNet", 64, 64, 16, 2, num_channels=3, output_nums=9, alpha=1.67, final_activation="softmax", is_transconv=False, train_mode="from_scratch").graph(), "cce"),
        ("UNet d3 w16 ag + lstm 64x64 bs4, mse", segamd.unet_model_builder("UNet", 64, 64, 16, 3, num_channels=1, ag=1, lstm=1, train_mode="from_scratch").graph(), "mse"),
    ]
    for name, g, loss in cases:
        W = {p.name: init_param(p.name, p.shape, p.init, p.fan, 3) for p in g.param_specs()}
        torch.manual_seed(0)
        H, Wd, C = g.nodes[g.inputs[0]].shape
        x = torch.rand(4, H, Wd, C)
        tgt = []
        for o in g.outputs:
            shp = (4,) + tuple(g.nodes[o].shape)
            tgt.append(torch.nn.functional.one_hot(torch.randint(0, shp[-1], shp[:-1]), shp[-1]).float() if loss == "cce" else (torch.rand(shp) > 0.5).float())
        MODE["on"] = False
        ref, outs = grads(g, W, x, tgt, loss, 1.0)
        # loss scale: the largest |dLoss/dOut| at 32 (flags.GRAD_TARGET_F32)
        n_out = sum(o.numel() for o in outs)
        dmax = max(float((2 * (o - t) / o.numel()).abs().max()) if loss == "mse" else float((t / o.clamp_min(1e-7) / (o.numel() / o.shape[-1])).abs().max()) for o, t in zip(outs, tgt))
        gn = max(float(v.norm()) for v in ref.values())
        for target in (32.0, 1024.0):
            S = 2.0 ** np.floor(np.log2(target / dmax))
            for dz_only in (False, True):
                MODE["on"], MODE["dz_only"] = True, dz_only
                got, _ = grads(g, W, x, tgt, loss, S)
                errs = sorted(((float((got[k] - ref[k]).norm()) / (float(ref[k].norm()) + 1e-3 * gn), k) for k in ref), reverse=True)
                med = np.median([e for e, _ in errs])
                print(f"{name:58s} target {int(target):5d} {'dz hi-only in both' if dz_only else 'dgrad: dz hi / wgrad: x hi'}: worst {errs[0][0]:.1e} ({errs[0][1]}), median {med:.1e}")
        MODE["on"] = False


def trajectory():
    """Adam steps on one batch with the exact backward pass, with 11-bit and with 22-bit backward operands: how far predict() moves."""
    from oracle.train_ref import KerasAdam

    def train(g, W0, x, tgt, steps, on):
        MODE["on"], MODE["dz_only"] = on, False
        W = {k: torch.from_numpy(np.array(v)).requires_grad_(not (k.endswith("moving_mean") or k.endswith("moving_variance"))) for k, v in W0.items()}
        keys = [k for k in W if W[k].requires_grad]
        opt = KerasAdam({k: W[k] for k in keys}, lr=1e-3)
        for _ in range(steps):
            for k in keys:
                W[k].grad = None
            outs = run_graph(g, x, W, training=True)
            loss = sum(((o - t) ** 2).mean() for o, t in zip(outs, tgt))
            (loss * 4096.0).backward()
            opt.step({k: W[k].grad / 4096.0 for k in keys})
        MODE["on"] = False
        with torch.no_grad():
            return [o.clone() for o in run_graph(g, x, W, training=False)]

    for name, g in (("UNet++ d3 w16 64x64 bs4 +DS", segamd.unet_model_builder("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1, train_mode="from_scratch").graph()),
                    ("MultiResUNet d2 w16 ag 64x64 bs4", segamd.unet_model_builder("MultiResUNet", 64, 64, 16, 2, num_channels=3, ag=1, alpha=1.67, train_mode="from_scratch").graph())):
        W0 = {p.name: init_param(p.name, p.shape, p.init, p.fan, 3) for p in g.param_specs()}
        torch.manual_seed(0)
        H, Wd, C = g.nodes[g.inputs[0]].shape
        x = torch.rand(4, H, Wd, C)
        tgt = [(torch.rand((4,) + tuple(g.nodes[o].shape)) > 0.5).float() for o in g.outputs]
        for steps in (1, 5, 20):
            a = train(g, W0, x, tgt, steps, False)
            row = []
            for bits in (11, 22):
                R["bits"] = bits
                torch.manual_seed(1)
                b = train(g, W0, x, tgt, steps, True)
                row.append(max(float((p - q).abs().max()) for p, q in zip(a, b)))
            R["bits"] = 11
            print(f"{name}: after {steps:2d} Adam steps: max |predict difference| against the exact backward pass: backward operands at 11 bits {row[0]:.1e}, at 22 bits {row[1]:.1e}")


if __name__ == "__main__":
    if "--trajectory" in sys.argv:
        trajectory()
    else:
        gradients()

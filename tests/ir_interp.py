"""Test helper: run the product's layer-graph IR on torch-CPU with the ORACLE's primitives.

Lives under tests/ on purpose - it is a checker for the host-side graph builder (wiring, names, shapes),
not a CPU fallback of the product."""
from typing import Dict, List

import torch

from oracle import keras_ref as K


class _R16(torch.autograd.Function):
    """round to fp16 and back, straight-through gradient: emulates the storage precision of the HIP pipeline."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _pool_take(x, size, idx):
    """max-pool of x (NHWC) with the arg-max positions IMPOSED (idx: flat h * W + w per (n, c, ph, pw), as F.max_pool2d returns)."""
    n, h, w, c = x.shape
    xc = x.permute(0, 3, 1, 2).reshape(n, c, h * w)
    out = xc.gather(2, idx.reshape(n, c, -1)).reshape(n, c, h // size[0], w // size[1])
    return out.permute(0, 2, 3, 1)


def run_graph(g, x: torch.Tensor, weights: Dict[str, torch.Tensor], training=True, return_all=False, emulate_fp16=False, impose=None,
              update_moving=False):
    """emulate_fp16: round every materialised tensor (and the MFMA weight operands) to fp16 exactly where the HIP
    engine does - model input, conv / convT / ConvLSTM weights, every layer output except the fp32 heads - so that
    activation masks (ReLU, max-pool arg-max, hard-sigmoid knees) agree with the GPU run and kernel logic can be
    checked tightly, separately from the fp16-vs-fp32 precision question.
    impose: {node name: decision} - the BRANCH DECISIONS of another run of the same graph (the GPU's, gpu_util.gpu_decisions)
    are taken instead of this run's own: a bool mask "positive branch" for a ReLU / LeakyReLU node, arg-max positions for a
    max-pool node, (low, high) knee masks of the three gates for a ConvLSTM node.  A value within the forward error (1e-6) of
    a branch point takes either branch legitimately, and the gradient behind it then differs by its whole value; with the
    decisions imposed what is left between the two runs' gradients is arithmetic alone (tests/test_gpu_fullsize.py)."""
    impose = impose or {}
    def hs(v, knees):
        if knees is None:
            return K.hard_sigmoid(v)
        lo, hi = knees
        return torch.where(hi, torch.ones_like(v), torch.where(lo, torch.zeros_like(v), 0.2 * v + 0.5))
    if g.ndim == 1 and x.dim() == 3:
        x = x.unsqueeze(1)
    R = _R16.apply if emulate_fp16 else (lambda t: t)
    outset = set(g.outputs)
    vals = {}
    for i in g.live_nodes():
        n = g.nodes[i]
        ins = [vals[j] for j in n.inputs]
        a = n.attrs
        if n.kind == "input":
            v = R(x)
        elif n.kind == "conv":
            b = weights.get(f"{n.name}/bias") if a["use_bias"] else None
            head = i in outset and a["kernel"] == (1, 1)          # fp32 head kernel: fp32 weights, fp32 output
            kern = weights[f"{n.name}/kernel"] if head else R(weights[f"{n.name}/kernel"])
            v = K.activation(K.conv2d_same(ins[0], kern, b, a["strides"]), a["activation"])
            if not head:
                v = R(v)
        elif n.kind == "convT":
            v = R(K.conv2d_transpose_same(ins[0], R(weights[f"{n.name}/kernel"]), weights[f"{n.name}/bias"], a["strides"]))
        elif n.kind == "bn":
            v, nm, nv = K.batchnorm(ins[0], weights[f"{n.name}/gamma"], weights[f"{n.name}/beta"],
                                    weights[f"{n.name}/moving_mean"], weights[f"{n.name}/moving_variance"], training,
                                    fused_4d=(g.ndim == 2), eps=a["eps"], momentum=a["momentum"])
            if update_moving and training:          # (a training loop: the layer's moving statistics follow, as Keras updates them)
                with torch.no_grad():
                    weights[f"{n.name}/moving_mean"].copy_(nm)
                    weights[f"{n.name}/moving_variance"].copy_(nv)
            v = R(v)
        elif n.kind == "act":
            if n.name in impose and a["fn"] in ("relu", "leaky_relu"):
                slope = a["alpha"] if a["fn"] == "leaky_relu" else 0.0
                v = R(torch.where(impose[n.name], ins[0], ins[0] * slope))
            else:
                v = R(K.activation(ins[0], a["fn"], a["alpha"]))
        elif n.kind == "maxpool":
            v = _pool_take(ins[0], a["size"], impose[n.name]) if n.name in impose else K.maxpool(ins[0], a["size"])
        elif n.kind == "upsample":
            v = R(K.upsample(ins[0], a["size"], a["interpolation"]))
        elif n.kind == "concat":
            v = torch.cat(ins, -1)
        elif n.kind == "add":
            v = R(sum(ins[1:], ins[0]))
        elif n.kind == "mul":
            v = R(ins[0] * ins[1])
        elif n.kind == "convlstm":
            z = R(K.conv2d_same(ins[0], R(weights[f"{n.name}/kernel"]), weights[f"{n.name}/bias"]))
            zi, zf, zc, zo = torch.chunk(z, 4, dim=-1)
            kn = impose.get(n.name)
            v = R(hs(zo, kn["o"] if kn else None) * torch.tanh(hs(zi, kn["i"] if kn else None) * torch.tanh(zc)))
        elif n.kind == "dense_ae":
            N, H, W, C = ins[0].shape
            z = ins[0].reshape(N, -1) @ weights[f"{n.name}/features/kernel"] + weights[f"{n.name}/features/bias"]
            v = (z @ weights[f"{n.name}/expand/kernel"] + weights[f"{n.name}/expand/bias"]).reshape(N, H, W, -1)
        else:
            raise NotImplementedError(n.kind)
        assert tuple(v.shape[1:]) == tuple(n.shape), (n.name, v.shape, n.shape)
        vals[i] = v
    outs = [vals[i] for i in g.outputs]
    if g.ndim == 1:
        outs = [o.squeeze(1) for o in outs]
    if return_all:
        return outs, {g.nodes[i].name: v for i, v in vals.items()}
    return outs

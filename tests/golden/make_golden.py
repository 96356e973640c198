#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz.

These are REGRESSION vectors produced by the oracle (oracle/keras_ref.py, oracle/models_ref.py, oracle/train_ref.py)
- a CPU restatement of the reference's Keras-2.x semantics - on fixed seeded inputs.  They are NOT outputs of the
reference: TensorFlow cannot be installed here and the reference ships no tests or golden vectors, so parity stays
"unpinned" (docs/experiments.md section 3).  What the file pins is the oracle itself: a later edit of the oracle (or of the
product) that changes any of these numbers is caught by tests/test_golden.py, on the CPU (oracle vs file) and on the
GPU (HIP path in exact-fp32 mode vs file).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import keras_ref as K
from oracle.keras_ref import Layers
from oracle.models_ref import unet2d_forward
from oracle.train_ref import KerasAdam, loss_and_grads

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")


def tensors(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda *shape: torch.randn(shape, generator=g)


def primitive_cases():
    r = tensors(7)
    out = {}
    x, k, b = r(2, 6, 5, 3), r(3, 3, 3, 4) * 0.3, r(4) * 0.1
    out["conv3x3_s1"] = (dict(x=x, kernel=k, bias=b), K.conv2d_same(x, k, b, (1, 1)))
    out["conv1x1_s2"] = (dict(x=x, kernel=r(1, 1, 3, 2)), None)
    out["conv1x1_s2"] = (out["conv1x1_s2"][0], K.conv2d_same(x, out["conv1x1_s2"][0]["kernel"], None, (2, 2)))
    kt, bt = r(4, 4, 2, 3) * 0.3, r(2) * 0.1
    out["convT4x4_s2"] = (dict(x=x, kernel=kt, bias=bt), K.conv2d_transpose_same(x, kt, bt, (2, 2)))
    xb, ga, be = r(3, 4, 4, 5) * 2 + 0.5, torch.rand(5) + 0.5, r(5) * 0.2
    mm, mv = torch.zeros(5), torch.ones(5)
    y, nm, nv = K.batchnorm(xb, ga, be, mm, mv, True)
    out["bn_train_4d"] = (dict(x=xb, gamma=ga, beta=be), torch.cat([y.reshape(-1), nm, nv]))
    out["maxpool2"] = (dict(x=x[:, :4, :4]), K.maxpool(x[:, :4, :4], (2, 2)))
    out["upsample_bilinear2"] = (dict(x=x), K.upsample(x, (2, 2), "bilinear"))
    kl, bl = r(3, 3, 3, 8) * 0.3, r(8) * 0.1
    out["convlstm_t1"] = (dict(x=x, kernel=kl, bias=bl), K.convlstm_t1(x, kl, bl))
    return out


def model_case():
    torch.manual_seed(0)
    L = Layers(2, seed=11)
    x = torch.rand(2, 16, 16, 3, generator=torch.Generator().manual_seed(3))
    fwd = lambda: unet2d_forward(L, x, "UNetPP", 8, 2, ds=1)
    with torch.no_grad():
        L.update_moving = False
        outs = fwd()
    tg = [(torch.rand(o.shape, generator=torch.Generator().manual_seed(5 + i)) > 0.5).float() for i, o in enumerate(outs)]
    loss, _per, outs2, grads = loss_and_grads(L, fwd, tg, "bce")
    w0 = {k: v.detach().clone() for k, v in L.params.items()}
    opt = KerasAdam({k: v for k, v in L.params.items() if L.trainable[k]}, lr=1e-3)
    opt.step(grads)
    names = sorted(grads)
    pick = [names[0], names[len(names) // 2], names[-1]]
    d = {"x": x, "loss": torch.tensor(float(loss))}
    for i, (o, t) in enumerate(zip(outs2, tg)):
        d[f"out{i}"], d[f"target{i}"] = o.detach(), t
    for n in pick:
        d["grad:" + n] = grads[n]
        d["adam1:" + n] = L.params[n].detach()
    for k, v in w0.items():
        d["w:" + k] = v
    return d


def main():
    z = {}
    for name, (inp, out) in primitive_cases().items():
        for k, v in inp.items():
            z[f"{name}/in/{k}"] = v.numpy()
        z[f"{name}/out"] = out.detach().numpy()
    for k, v in model_case().items():
        z["unetpp_w8_d2_ds/" + k] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(OUT, **z)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(z), "arrays")


if __name__ == "__main__":
    main()

"""tests/golden/oracle_vectors.npz: regression vectors of the oracle (see tests/golden/make_golden.py - oracle outputs,
not reference outputs; parity stays unpinned).  CPU: the oracle still reproduces them.  GPU: the HIP path in the parity
mode reproduces the model-level vectors within the north-star tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import keras_ref as K
from oracle.keras_ref import Layers
from oracle.models_ref import unet2d_forward
from oracle.train_ref import loss_and_grads

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
T = lambda k: torch.from_numpy(Z[k])
PRE = "unetpp_w8_d2_ds/"


def close(a, b, tol):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_oracle_primitives_reproduce_the_vectors():
    i = lambda n, k: T(f"{n}/in/{k}")
    assert close(K.conv2d_same(i("conv3x3_s1", "x"), i("conv3x3_s1", "kernel"), i("conv3x3_s1", "bias"), (1, 1)), T("conv3x3_s1/out"), 1e-6)
    assert close(K.conv2d_same(i("conv1x1_s2", "x"), i("conv1x1_s2", "kernel"), None, (2, 2)), T("conv1x1_s2/out"), 1e-6)
    assert close(K.conv2d_transpose_same(i("convT4x4_s2", "x"), i("convT4x4_s2", "kernel"), i("convT4x4_s2", "bias"), (2, 2)),
                 T("convT4x4_s2/out"), 1e-6)
    y, nm, nv = K.batchnorm(i("bn_train_4d", "x"), i("bn_train_4d", "gamma"), i("bn_train_4d", "beta"), torch.zeros(5), torch.ones(5), True)
    assert close(torch.cat([y.reshape(-1), nm, nv]), T("bn_train_4d/out"), 1e-6)
    assert close(K.maxpool(i("maxpool2", "x"), (2, 2)), T("maxpool2/out"), 0.0)
    assert close(K.upsample(i("upsample_bilinear2", "x"), (2, 2), "bilinear"), T("upsample_bilinear2/out"), 1e-6)
    assert close(K.convlstm_t1(i("convlstm_t1", "x"), i("convlstm_t1", "kernel"), i("convlstm_t1", "bias")), T("convlstm_t1/out"), 1e-6)


def _weights():
    return {k[len(PRE) + 2:]: Z[k] for k in Z.files if k.startswith(PRE + "w:")}


def _targets():
    return [T(f"{PRE}target{i}") for i in range(sum(1 for k in Z.files if k.startswith(PRE + "target")))]


def test_oracle_model_reproduces_the_vectors():
    L = Layers(2, seed=99)
    x = T(PRE + "x")
    fwd = lambda: unet2d_forward(L, x, "UNetPP", 8, 2, ds=1)
    with torch.no_grad():
        L.update_moving = False
        fwd()                      # creates the parameters
    L.set_weights(_weights())
    loss, _per, outs, grads = loss_and_grads(L, fwd, _targets(), "bce")
    assert abs(float(loss) - float(Z[PRE + "loss"])) < 1e-5
    for i, o in enumerate(outs):
        assert close(o.detach(), T(f"{PRE}out{i}"), 1e-5)
    for k in Z.files:
        if k.startswith(PRE + "grad:"):
            assert close(grads[k[len(PRE) + 5:]], T(k), 1e-5)


@pytest.mark.gpu
def test_hip_fp32_mode_reproduces_the_model_vectors():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import segamd
    m = segamd.unet_model_builder("UNetPP", 16, 16, 8, 2, ds=1, num_channels=3, train_mode="from_scratch").ResNet50()
    m.build(2, dtype="f32")
    m.compile(loss="bce", lr=1e-3)
    m.set_weights(_weights())
    x, tg = T(PRE + "x"), _targets()
    outs = m.forward(x, training=True)
    loss = float(m.compute_loss(tg, "bce").cpu())
    m.backward(None)
    grads = m.get_grads()
    m.optimizer_step()
    after = m.get_weights()
    assert abs(loss - float(Z[PRE + "loss"])) < 1e-4
    for i, o in enumerate(outs):
        assert float((o.cpu() - T(f"{PRE}out{i}")).abs().max()) < 1e-3            # the north-star tolerance (measured ~1e-6)
        assert float((o.cpu() - T(f"{PRE}out{i}")).abs().max()) < 1e-4
    for k in Z.files:
        if k.startswith(PRE + "grad:"):
            n = k[len(PRE) + 5:]
            ref = Z[k]
            assert np.linalg.norm(grads[n] - ref) <= 5e-3 * np.linalg.norm(ref) + 1e-7
        if k.startswith(PRE + "adam1:"):
            n = k[len(PRE) + 6:]
            # the first Adam step moves every weight by ~ lr * sign(g) = +-1e-3: only weights whose gradient is numerically
            # zero may land on the other side
            d = np.abs(after[n] - Z[k])
            assert d.max() < 2.1e-3 and d.mean() < 5e-5, (n, d.max(), d.mean())

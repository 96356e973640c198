"""GPU parity tests, model level, through the reference's builder API (unet_model_builder / UNet1D) and the C ABI.

Two references per model (both from the oracle's primitives, tests/ir_interp.py):
  * fp32 oracle  - the north-star comparison, here for the f16-MFMA mode (the parity mode, dtype f32, is held to 1e-3 /
    bit-exact masks in test_gpu_fp32.py).  fp16 has a 2^-11 relative rounding step per stored activation, so the
    bound is 1e-2 x max(1, |ref|) on every output (measured: 1e-3 .. 2e-3 on sigmoid/softmax maps of the 2D
    models, up to 1.3e-2 on un-normalised linear 1D regression heads); masks (threshold 0.5 / channel arg-max,
    2DCNN/Test.py:176) must agree on >= 99 % of the pixels and on every pixel that is not within 1e-2 of a
    decision boundary ("near-tie accounting").
  * fp16-emulating oracle (same rounding points as the kernels) - checks kernel logic: loss to 1e-4 relative.
Gradients: a single fp16 ulp upstream flips ReLU masks, so element-wise maxima are meaningless at model level
(the op-level tests in test_gpu_ops.py are exact); here the per-parameter cosine similarity with the fp32
oracle's gradient must be >= 0.97 and the norm ratio within 10 %."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import segamd
import gpu_util as U
from oracle.keras_ref import Layers
from oracle.models_ref import unet2d_forward
from oracle.train_ref import KerasOptimizer, KerasAdam, loss_and_grads

OUT_TOL = 1e-2
# MultiRes models stack BatchNormalization on BatchNormalization over 5..16-channel tensors; channels whose batch
# mean is large against their standard deviation amplify the fp16 rounding of the stored pre-BN value by mean/sigma
# (docs/experiments.md section 3), so the f16-mode bound is looser there.  The parity mode (test_gpu_fp32.py) is unaffected.
OUT_TOL_CASE = {"multiresunet_ag_bce": 3e-2, "multiresunet_alpha167_ds_mse": 3e-2, "multiresunet1d_mse": 3e-2, "multiresunet_40x24_ag_bce": 3e-2}


def B(*a, **k):
    return segamd.unet_model_builder(*a, train_mode="from_scratch", **k).ResNet50()


def one_hot(shape, ncls):
    return F.one_hot(torch.randint(0, ncls, shape), ncls).float()


def _with_targets(model, x, loss):
    """Random targets shaped like the model's outputs ([out, level1, ...]; 1D outputs without the H = 1 axis)."""
    g = model.graph
    tg = []
    for o in g.outputs:
        H, W, C = g.nodes[o].shape
        tg.append(torch.rand(x.shape[0], W, C) if g.ndim == 1 else torch.rand(x.shape[0], H, W, C))
    return model, x, tg, loss


CASES = {
    "unet_softmax_cce": lambda: (B("UNet", 32, 32, 16, 2, num_channels=3, output_nums=4, final_activation="softmax"),
                                 torch.rand(2, 32, 32, 3), [one_hot((2, 32, 32), 4)], "cce"),
    "unet_21class_softmax_cce": lambda: (B("UNet", 32, 32, 16, 2, num_channels=3, output_nums=21, final_activation="softmax"),
                                         torch.rand(2, 32, 32, 3), [one_hot((2, 32, 32), 21)], "cce"),
    "unet_12label_sigmoid_bce": lambda: (B("UNet", 32, 32, 16, 2, num_channels=3, output_nums=12, final_activation="sigmoid"),
                                         torch.rand(2, 32, 32, 3), [(torch.rand(2, 32, 32, 12) > 0.5).float()], "bce"),
    "unet_autoencoder_mse": lambda: (B("UNet", 32, 32, 16, 2, num_channels=3, ae=1, feature_number=24), torch.rand(2, 32, 32, 3),
                                     [torch.rand(2, 32, 32, 1)], "mse"),
    "unet1d_autoencoder_mse": lambda: (segamd.UNet1D(128, 2, 1, 16, 3, ds=0, ae=1, feature_number=40).UNet(), torch.rand(4, 128, 1),
                                       [torch.rand(4, 128, 1)], "mse"),
    "unetpp_ds_mse": lambda: (B("UNetPP", 32, 32, 16, 3, num_channels=3, ds=1), torch.rand(2, 32, 32, 3),
                              [(torch.rand(2, 32, 32, 1) > 0.5).float()] * 4, "mse"),
    "unet_convlstm_bce": lambda: (B("UNet", 32, 32, 16, 2, num_channels=1, lstm=1), torch.rand(2, 32, 32, 1),
                                  [(torch.rand(2, 32, 32, 1) > 0.5).float()], "bce"),
    "unet_ag_ds_mse": lambda: (B("UNet", 32, 32, 16, 2, num_channels=3, ag=1, ds=1), torch.rand(2, 32, 32, 3),
                               [torch.rand(2, 32, 32, 1), torch.rand(2, 16, 16, 1), torch.rand(2, 8, 8, 1)], "mse"),
    "unet_w64_bce": lambda: (B("UNet", 64, 64, 64, 3, num_channels=3), torch.rand(2, 64, 64, 3),
                             [(torch.rand(2, 64, 64, 1) > 0.5).float()], "bce"),
    "unete_upsampling_bce": lambda: (B("UNetE", 32, 32, 16, 2, num_channels=3, is_transconv=False), torch.rand(2, 32, 32, 3),
                                     [(torch.rand(2, 32, 32, 1) > 0.5).float()], "bce"),
    "unet3p_mse": lambda: (B("UNet3P", 32, 32, 8, 2, num_channels=3), torch.rand(2, 32, 32, 3), [torch.rand(2, 32, 32, 1)], "mse"),
    "multiresunet_ag_bce": lambda: (B("MultiResUNet", 32, 32, 32, 2, num_channels=3, ag=1, alpha=1.0), torch.rand(2, 32, 32, 3),
                                    [(torch.rand(2, 32, 32, 1) > 0.5).float()], "bce"),
    "multiresunet_alpha167_ds_mse": lambda: (B("MultiResUNet", 32, 32, 12, 2, num_channels=1, ds=1, alpha=1.67), torch.rand(2, 32, 32, 1),
                                             [torch.rand(2, 32, 32, 1), torch.rand(2, 16, 16, 1), torch.rand(2, 8, 8, 1)], "mse"),
    "multiresunet1d_mse": lambda: (segamd.UNet1D(128, 2, 1, 16, 3, ds=0, alpha=1).MultiResUNet(), torch.rand(4, 128, 1),
                                   [torch.rand(4, 128, 1)], "mse"),
    "unet1d_softmax_cce": lambda: (segamd.UNet1D(256, 2, 1, 16, 3, "Classification", output_nums=2, ds=0).UNet(),
                                   torch.rand(4, 256, 1), [one_hot((4, 256), 2)], "cce"),
    "unetpp1d_ds_mse": lambda: (segamd.UNet1D(128, 2, 1, 16, 3, ds=1).UNetPP(), torch.rand(4, 128, 1),
                                [torch.rand(4, 128, 1), torch.rand(4, 128, 1), torch.rand(4, 128, 1)], "mse"),
    # the remaining same-kernel variants (2DCNN unet_variants.py:217-274; 1DCNN BCDUNet.py:111-174)
    "unetp_bce": lambda: (B("UNetP", 32, 32, 16, 2, num_channels=3), torch.rand(2, 32, 32, 3),
                          [(torch.rand(2, 32, 32, 1) > 0.5).float()], "bce"),
    "unetp_ag_ds_mse": lambda: _with_targets(B("UNetP", 32, 32, 16, 2, num_channels=3, ag=1, ds=1), torch.rand(2, 32, 32, 3), "mse"),
    "bcdunet1d_lstm_dense2_mse": lambda: _with_targets(segamd.BCDUNet(128, 2, 1, 16, 3, lstm=1, dense_loop=2, ds=1).BCDUNet(),
                                                      torch.rand(4, 128, 1), "mse"),
    # geometries the halo patches do not tile (the reference takes any size divisible by 2^depth): non-square images, widths that are
    # not a multiple of 16, odd batch, a 1D length whose deep levels are odd - per-tap kernels with ragged tiles inside whole models
    "unet_24x40_batch3_bce": lambda: (B("UNet", 24, 40, 16, 2, num_channels=3), torch.rand(3, 24, 40, 3),
                                      [(torch.rand(3, 24, 40, 1) > 0.5).float()], "bce"),
    "unetpp_48x32_ds_mse": lambda: _with_targets(B("UNetPP", 48, 32, 16, 3, num_channels=3, ds=1), torch.rand(2, 48, 32, 3), "mse"),
    "multiresunet_40x24_ag_bce": lambda: (B("MultiResUNet", 40, 24, 16, 2, num_channels=3, ag=1, alpha=1.0), torch.rand(2, 40, 24, 3),
                                          [(torch.rand(2, 40, 24, 1) > 0.5).float()], "bce"),
    "unet1d_len200_mse": lambda: (segamd.UNet1D(200, 2, 1, 16, 3, ds=0).UNet(), torch.rand(4, 200, 1), [torch.rand(4, 200, 1)], "mse"),
}


# f16 mode, gradient of every well-conditioned parameter tensor against the ROUNDING-EMULATING oracle with the GPU's branch decisions
# imposed: measured 1.0e-3 .. 4.0e-3 (fp16 storage of the activation gradients), 1.0e-2 .. 2.0e-2 on the MultiRes / BCD-1D stacks
F16_KERNEL_GRAD_L2 = 1.5e-2
F16_KERNEL_GRAD_L2_FRAGILE = 6e-2


def masks(p):
    return (p >= 0.5) if p.shape[-1] == 1 else p.argmax(-1, keepdim=True)


def margin(p):
    if p.shape[-1] == 1:
        return (p - 0.5).abs()
    top = p.topk(2, -1).values
    return (top[..., :1] - top[..., 1:])


@pytest.mark.parametrize("name", sorted(CASES))
def test_model_forward_backward_parity(name):
    torch.manual_seed(7)
    model, x, targets, loss = CASES[name]()
    model.build(x.shape[0], dtype="f16")
    r32 = U.compare_model(model, x, targets, loss, emulate_fp16=False)
    r16 = U.compare_model(model, x, targets, loss, emulate_fp16=True, impose_decisions=True)
    # ---- forward, vs the fp32 oracle
    tol = OUT_TOL_CASE.get(name, OUT_TOL)
    for e, m in zip(r32["outs"], r32["out_mag"]):
        assert e < tol * max(1.0, m), (r32["outs"], r32["out_mag"])
    assert abs(r32["loss"][0] - r32["loss"][1]) < (tol / 5) * max(1.0, abs(r32["loss"][1]))
    # ---- forward, vs the rounding-emulating oracle (kernel logic)
    assert abs(r16["loss"][0] - r16["loss"][1]) < max(2e-4, tol / 20) * max(1.0, abs(r16["loss"][1])), r16["loss"]
    assert max(e / max(1.0, m) for _n, e, m in r16["nodes"]) < 4e-2    # a few fp16 ulps on O(1..8) activations
    # ---- gradients vs the fp32 oracle: direction and size per parameter
    for k, (cos, ratio, mag) in r32["gcos"].items():
        if mag < 1e-3 * r32["gmax"]:
            continue                                        # numerically-zero gradients (conv bias under BN)
        lim = 0.97 if name not in OUT_TOL_CASE else 0.80
        assert cos > lim, (k, cos)
        # (the fragile MultiRes cases: two equivalent lowerings of the same graph - BN tails fused or not, SEG_FUSE_BN_TAILS - differ
        # by this much between themselves in f16 mode: relative error 0.60 against 0.32 on enc1/mrb/c3/bn/beta, tools/dbg_f32_case.py)
        assert (0.90 if name not in OUT_TOL_CASE else 0.60) < ratio < (1.10 if name not in OUT_TOL_CASE else 1.60), (k, ratio)
    # ---- gradients vs the ROUNDING-EMULATING oracle with the GPU's branch decisions imposed (kernel logic of the backward pass: what
    # the wide band above cannot see in the fragile cases - a missing ReLU mask or c2 term in a fused tail moves these by O(1))
    gn = max(n for _e, n in r16["gl2_imposed"].values())
    worst = max(((e / n, k) for k, (e, n) in r16["gl2_imposed"].items() if n > 1e-2 * gn and k not in r16["zero_grad_params"]), default=(0.0, ""))
    print(f"[{name}] f16 gradient rel-L2 vs the rounding-emulating oracle, decisions imposed: worst {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < (F16_KERNEL_GRAD_L2_FRAGILE if (name in OUT_TOL_CASE or name.startswith('bcdunet1d')) else F16_KERNEL_GRAD_L2), worst


@pytest.mark.parametrize("name", ["unet_softmax_cce", "unet_w64_bce", "unet1d_softmax_cce"])
def test_masks_match_reference_rule(name):
    """2DCNN/Test.py:161-176: predict (inference-mode BN) -> threshold 0.5; multi-class: channel arg-max."""
    torch.manual_seed(3)
    model, x, targets, loss = CASES[name]()
    model.build(x.shape[0], dtype="f16")
    for _ in range(2):                                       # make the moving statistics non-trivial
        model.train_step(x, targets, loss)
    assert model.nonfinite_updates() == 0                    # the optimiser skipped nothing (no fp16 overflow in the backward pass)
    w = {k: torch.from_numpy(v) for k, v in model.get_weights().items()}
    from ir_interp import run_graph
    ref = run_graph(model.graph, x, w, training=False)[0]
    got = torch.from_numpy(model.predict(x) if len(model.output_names) == 1 else model.predict(x)[0])
    assert float((got - ref).abs().max()) < OUT_TOL
    same = masks(got) == masks(ref)
    assert float(same.float().mean()) >= 0.99
    assert bool(same[margin(ref) > OUT_TOL].all())          # every pixel that is not a near-tie agrees exactly


def test_train_steps_follow_the_oracle():
    """Three optimiser steps (forward, loss, backward, Keras Adam) on GPU vs the oracle's train step."""
    torch.manual_seed(11)
    kw = dict(ds=1)
    model = B("UNetPP", 32, 32, 16, 2, num_channels=3, **kw)
    model.build(2, dtype="f16")
    model.compile(loss="mse", lr=1e-3)
    x = torch.rand(2, 32, 32, 3)
    tg = [(torch.rand(2, 32, 32, 1) > 0.5).float()] * 3
    L = Layers(2, seed=0)
    L.set_weights(model.get_weights())
    opt = KerasAdam({k: v for k, v in L.params.items() if L.trainable[k]}, lr=1e-3)
    fwd = lambda: unet2d_forward(L, x, "UNetPP", 16, 2, **kw)
    gl, rl = [], []
    for _ in range(3):
        gl.append(float(model.train_step(x, tg).cpu()))
        l, _per, _o, grads = loss_and_grads(L, fwd, tg, "mse")
        opt.step(grads)
        rl.append(l)
    assert all(abs(a - b) < 5e-3 * max(1.0, abs(b)) for a, b in zip(gl, rl)), (gl, rl)
    assert gl[-1] < gl[0]                                    # the loss goes down
    wg, wr = model.get_weights(), L.get_weights()
    for k in ("X01/conv/kernel", "enc1/bn/moving_mean", "enc1/bn/moving_variance", "out/kernel"):
        d = np.abs(wg[k] - wr[k]).max()
        assert d < 5e-3 * max(1.0, np.abs(wr[k]).max()), (k, d)


def test_registry_loss_with_loss_weights_follows_the_oracle():
    """compile(loss='MeanSquaredLogarithmicError' - the Train_Configs.ini default -, loss_weights=1 - 0.1 k as the 1D
    notebook passes them): two optimiser steps against the oracle's weighted sum."""
    torch.manual_seed(5)
    kw = dict(ds=1)
    model = B("UNetPP", 32, 32, 16, 2, num_channels=3, **kw)
    model.build(2, dtype="f16")
    lw = [1.0, 0.9, 0.8]
    model.compile(loss="MeanSquaredLogarithmicError", lr=1e-3, loss_weights=lw)
    x = torch.rand(2, 32, 32, 3)
    tg = [torch.rand(2, 32, 32, 1)] * 3
    L = Layers(2, seed=0)
    L.set_weights(model.get_weights())
    opt = KerasAdam({k: v for k, v in L.params.items() if L.trainable[k]}, lr=1e-3)
    fwd = lambda: unet2d_forward(L, x, "UNetPP", 16, 2, **kw)
    for _ in range(2):
        gl = float(model.train_step(x, tg).cpu())
        l, per, _o, grads = loss_and_grads(L, fwd, tg, "msle", lw)
        opt.step(grads)
        assert abs(gl - l) < 5e-3 * max(1.0, abs(l)), (gl, l)
        assert abs(l - sum(w * p for w, p in zip(lw, per))) < 1e-6
    with pytest.raises(ValueError):
        model.compile(loss="Dice")
    with pytest.raises(ValueError):
        model.compile(loss="mse", loss_weights=[1.0])
    model.compile(loss="mae", loss_weights={"out": 2.0})
    assert model.loss_weights == [2.0, 1.0, 1.0]


@pytest.mark.parametrize("name", ["SGD", "RMSprop", "Adagrad", "Adadelta", "Adamax", "Nadam", "FTRL"])
def test_registry_optimizer_training_follows_the_oracle(name):
    """compile(optimizer=<a name of utils/tf_optimizers.py>): three steps of the whole model against the oracle with the
    same optimiser (fp32 mode, so the comparison is about the update rule and its state, not fp16 rounding)."""
    torch.manual_seed(3)
    kw = dict(ds=1)
    model = B("UNetPP", 32, 32, 16, 2, num_channels=3, **kw)
    model.build(2, dtype="f32")
    lr = {"Adadelta": 1.0, "SGD": 1e-2, "FTRL": 1e-2}.get(name, 1e-3)
    model.compile(loss="mse", optimizer=name, lr=lr)
    x = torch.rand(2, 32, 32, 3)
    tg = [(torch.rand(2, 32, 32, 1) > 0.5).float()] * 3
    L = Layers(2, seed=0)
    L.set_weights(model.get_weights())
    opt = KerasOptimizer(name, {k: v for k, v in L.params.items() if L.trainable[k]}, lr=lr)
    fwd = lambda: unet2d_forward(L, x, "UNetPP", 16, 2, **kw)
    for _ in range(3):
        gl = float(model.train_step(x, tg).cpu())
        l, _per, _o, grads = loss_and_grads(L, fwd, tg, "mse")
        opt.step(grads)
        assert abs(gl - l) < 2e-3 * max(1.0, abs(l)), (name, gl, l)
    wg, wr = model.get_weights(), L.get_weights()
    for k in ("X01/conv/kernel", "enc1/conv/kernel", "out/kernel", "enc1/bn/gamma"):
        d = np.abs(wg[k] - wr[k]).max()
        assert d < 2e-3 * max(1.0, np.abs(wr[k]).max()), (name, k, d)
    with pytest.raises(ValueError):
        model.compile(loss="mse", optimizer="Lion")


@pytest.mark.parametrize("kw", [dict(ae=1, feature_number=24), dict(output_nums=12, final_activation="sigmoid")],
                         ids=["autoencoder", "wide_head"])
def test_multi_op_nodes_train_like_the_oracle(kw):
    """Graph nodes that lower to several ops (Dense bottleneck = reshape + 2 GEMMs + reshape; a wide head = GEMM +
    activation) over three optimiser steps: their parameters must be updated before the next forward reads them."""
    torch.manual_seed(2)
    model = B("UNet", 32, 32, 16, 2, num_channels=3, **kw)
    model.build(2, dtype="f32")
    model.compile(loss="mse", lr=1e-3)
    x = torch.rand(2, 32, 32, 3)
    tg = [torch.rand(2, 32, 32, kw.get("output_nums", 1))]
    L = Layers(2, seed=0)
    L.set_weights(model.get_weights())
    opt = KerasAdam({k: v for k, v in L.params.items() if L.trainable[k]}, lr=1e-3)
    fwd = lambda: unet2d_forward(L, x, "UNet", 16, 2, **kw)
    for _ in range(3):
        gl = float(model.train_step(x, tg).cpu())
        l, _per, _o, grads = loss_and_grads(L, fwd, tg, "mse")
        opt.step(grads)
        assert abs(gl - l) < 1e-3 * max(1.0, abs(l)), (gl, l)
    wg, wr = model.get_weights(), L.get_weights()
    worst = max(float(np.abs(wg[k] - wr[k]).max()) for k in wr if L.trainable.get(k, False))
    assert worst < 2e-3, worst


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("make,xshape", [
    (lambda: B("UNet3P", 16, 16, 12, 2, num_channels=3), (2, 16, 16, 3)),                       # sigmoid on 12-channel upsampled branches
    (lambda: B("MultiResUNet", 16, 16, 12, 2, num_channels=3, ag=1, alpha=1.67), (2, 16, 16, 3)),  # attention gates: sigmoid(psi), 1 channel
    (lambda: B("UNet", 16, 16, 12, 2, num_channels=5, lstm=1, ae=1, feature_number=10), (2, 16, 16, 5)),   # ConvLSTM F = 12, Dense bottleneck
    (lambda: segamd.BCDUNet(40, 2, 1, 12, 3, lstm=1, ag=1, dense_loop=2).BCDUNet(), (3, 40, 1)),
], ids=["unet3p_w12", "multires_ag_w12", "unet_lstm_ae_w12", "bcdunet1d_w12"])
def test_padding_weights_stay_zero_through_training(make, xshape, dtype):
    """Tensors are padded to 8 channels and GEMM operands to 32 K columns; the padding weights only ever meet padding activations,
    which every kernel keeps at zero - so their gradient is zero and they never move.  Round 5's fuzzer found the exception:
    Activation('sigmoid') wrote sigmoid(0) = 0.5 into the padding channels, the next conv's padding weights moved, and every model with
    a sigmoid on a tensor whose width is not a multiple of 8 (UNet 3+; the attention gate's 1-channel map in front of a transposed
    conv) drifted from the reference from the second optimiser step on.  Three steps with a large learning rate, then every parameter
    must round-trip through its Keras layout unchanged."""
    torch.manual_seed(4)
    m = make()
    x = torch.rand(*xshape)
    m.build(x.shape[0], dtype=dtype)
    m.compile(loss="mse", optimizer="SGD", lr=5e-2)
    _m, _x, tg, _l = _with_targets(m, x, "mse")
    assert U.padding_residues(m) == {}
    for _ in range(3):
        m.train_step(x, tg)
    assert m.nonfinite_updates() == 0
    assert U.padding_residues(m) == {}


"""MI355X: the two-MFMA forms of the parity-mode BACKWARD GEMMs (SegModel.build(backward="2mfma")).
A data-gradient launch reads its A operand - the incoming gradient dz - as the fp16 hi half of the split pair
(SEG_VARIANT_A_HI: a b ~= ah bh + ah bl / 2048), a weight-gradient launch its INPUT operand (SEG_VARIANT_IN_HI: ah bh + al bh / 2048).
What defines them exactly: the result IS the three-MFMA launch on that operand rounded to fp16 (a pair whose lo half is zero) - bit for
bit, nothing else may change - and therefore within 2^-11 elementwise rounding of the operand of the fp32 convolution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import gpu_util as U
from oracle import keras_ref as K

ffi, M = U.ffi, U.M


def rel(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("N,H,W,cins,cout,kind", [
    (2, 32, 32, [64], 64, "conv"), (1, 64, 64, [32], 17, "conv"), (2, 16, 32, [40], 96, "conv"), (2, 16, 32, [24, 40], 72, "conv"),
    (2, 32, 16, [64], 64, "conv"),                       # 16 x 16 patches
])
def test_hi_only_gradient_operand(N, H, W, cins, cout, kind):
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    xs = [torch.randn(N, H, W, c) * 3.0 for c in cins]
    cin = sum(cins)
    tile = (256, 64 if cout > 32 else 32)
    kern = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    run = lambda xs_, v: U.igemm_conv(xs_, kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=v)
    exact = K.conv2d_same(torch.cat(xs, -1), torch.from_numpy(kern), None, (1, 1))
    exact_hi = K.conv2d_same(torch.cat([U.h16(x) for x in xs], -1), torch.from_numpy(kern), None, (1, 1))
    full, _s, pad0 = run(xs, 3)
    half, _s, pad1 = run(xs, 3 | ffi.SEG_VARIANT_A_HI)
    full_of_hi, _s, _p = run([U.h16(x) for x in xs], 3)
    assert pad0 == 0.0 and pad1 == 0.0
    assert rel(full, exact) < 2e-6
    assert torch.equal(half, full_of_hi)                 # the definition: the three-MFMA launch on the operand's hi half
    assert rel(half, exact_hi) < 2e-6
    e = rel(half, exact)
    print(f"A operand hi only against the fp32 convolution: {e:.1e}")
    assert 1e-6 < e < 5e-4                               # 2^-12 rms per element of the operand, averaged over the contraction
    d = half - full
    assert abs(float(d.mean())) < 0.1 * float(d.abs().mean())       # unbiased
    # SEG_VARIANT_B_HI: the weight operand as its hi half instead (what a data-gradient launch of backward="b2w*" does)
    kern_hi = U.h16(torch.from_numpy(kern)).numpy()
    halfb, _s, padb = U.igemm_conv(xs, kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=3 | ffi.SEG_VARIANT_B_HI)
    full_of_bhi, _s, _p = U.igemm_conv(xs, kern_hi, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=3)
    assert padb == 0.0 and torch.equal(halfb, full_of_bhi)
    eb = rel(halfb, exact)
    print(f"B operand hi only against the fp32 convolution: {eb:.1e}")
    assert 1e-6 < eb < 5e-4


@pytest.mark.parametrize("N,H,W,cin,cout,kind", [(2, 32, 32, 64, 64, "conv"), (2, 16, 32, 40, 96, "conv"), (2, 16, 16, 64, 64, "convT"),
                                                 (2, 16, 32, 48, 72, "convT")])
def test_hi_only_input_operand_in_the_weight_gradient(N, H, W, cin, cout, kind):
    torch.manual_seed(1)
    x = torch.randn(N, H, W, cin) * 2.0
    k = (3, 3) if kind == "conv" else (4, 4)
    s = (1, 1) if kind == "conv" else (2, 2)
    kshape = (k[0], k[1], cin, cout) if kind == "conv" else (k[0], k[1], cout, cin)
    Ho, Wo = (H, W) if kind == "conv" else (2 * H, 2 * W)
    dz = torch.randn(N, Ho, Wo, cout)
    full = U.igemm_wgrad([x], dz, kshape, kind, s, 2, None, None, f32=True)
    half = U.igemm_wgrad([x], dz, kshape, kind, s, 2 | ffi.SEG_VARIANT_A_HI, None, None, f32=True)
    full_of_hi = U.igemm_wgrad([U.h16(x)], dz, kshape, kind, s, 2, None, None, f32=True)
    # the definition: the three-MFMA launch on the input's hi half (to the summation order of the pixel split's atomics)
    assert float(np.abs(half - full_of_hi).max() / np.abs(full).max()) < 1e-6
    # one MFMA per product: dz as its hi half as well
    one = U.igemm_wgrad([x], dz, kshape, kind, s, 2 | ffi.SEG_VARIANT_IN_HI | ffi.SEG_VARIANT_DZ_HI, None, None, f32=True)
    full_of_both_hi = U.igemm_wgrad([U.h16(x)], U.h16(dz), kshape, kind, s, 2, None, None, f32=True)
    e1 = float(np.abs(one - full_of_both_hi).max() / np.abs(full).max())
    assert e1 < 1e-6, e1                                 # (one accumulator instead of two: the same products in another summation order)
    e2 = float(np.abs(one - full).max() / np.abs(full).max())
    print(f"weight gradient, both operands hi only against three MFMAs: {e2:.1e}")
    assert 1e-6 < e2 < 7e-4
    e = float(np.abs(half - full).max() / np.abs(full).max())
    print(f"weight gradient, input operand hi only against three MFMAs: {e:.1e}")
    assert 1e-6 < e < 5e-4
    d = half - full
    assert abs(float(d.mean())) < 0.1 * float(np.abs(d).mean())


def test_the_backward_mode_changes_the_backward_pass_only():
    """build(backward="2mfma") against build(backward="3mfma") on the same weights: the forward pass (outputs, loss, every stored tensor)
    is the same bits, the parameter gradients agree to the operand rounding, and `predict` after an optimiser step stays within the
    north star's tolerance of the strict mode's."""
    import segamd
    torch.manual_seed(3)
    x = torch.rand(2, 64, 64, 3)
    res = {}
    for mode in ("3mfma", "2mfma"):
        m = segamd.unet_model_builder("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1, train_mode="from_scratch").ResNet50()
        m.build(2, dtype="f32", backward=mode)
        assert m.backward_mode == mode and (m.bwd_d, m.bwd_w) == ((2, 2) if mode == "2mfma" else (3, 3))
        if res:
            m.set_weights(res["3mfma"][3])
        w = m.get_weights()
        tg = [(torch.rand(*h.y.shape) > 0.5).float() for h in m.out_heads] if not res else res["3mfma"][4]
        outs = [o.cpu().clone() for o in m.forward(x, training=True)]
        loss = float(m.compute_loss(tg, "mse").cpu())
        m.backward(None)
        m._join_side()
        g = {k: np.array(v) for k, v in m.get_grads().items()}
        n_hi = sum(1 for op in m.ops if isinstance(op, M.ConvOp) for a in [op.wg_args] if a.variant & ffi.SEG_VARIANT_A_HI)
        res[mode] = (outs, loss, g, w, tg, n_hi)
    o3, l3, g3, _w, _t, n3 = res["3mfma"]
    o2, l2, g2, _w, _t, n2 = res["2mfma"]
    assert n3 == 0 and n2 > 0
    assert all(torch.equal(a, b) for a, b in zip(o3, o2)) and abs(l3 - l2) < 1e-6 * abs(l3)       # (the loss is summed with float atomics: order)
    gn = max(float(np.linalg.norm(v)) for v in g3.values())
    worst = max(float(np.linalg.norm(g2[k] - g3[k])) / (float(np.linalg.norm(g3[k])) + 1e-3 * gn) for k in g3)
    assert 1e-7 < worst < 1e-3, worst


@pytest.mark.parametrize("N,H,W,cins,cout,k,tile", [
    (2, 8, 16, [256, 160], 136, 3, (128, 64)),      # a deep level: M = 256 pixels, K = 3 744
    (2, 16, 16, [96, 40], 64, 1, (256, 64)),        # 1x1
    (1, 16, 16, [192], 128, 3, (128, 128)),
])
def test_hi_only_weight_operand_on_the_per_tap_kernel(N, H, W, cins, cout, k, tile):
    """SEG_VARIANT_B_HI on variant 0 (the per-tap LDS-DMA kernel: deep levels, 1x1 convs; what SEG_BWD_B2_PER_TAP=1 adds to the b2 modes -
    off by default, flags.py): the three-MFMA launch on the weights' hi halves, bit for bit."""
    torch.manual_seed(2)
    rng = np.random.default_rng(2)
    xs = [torch.randn(N, H, W, c) for c in cins]
    cin = sum(cins)
    kern = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    kern_hi = U.h16(torch.from_numpy(kern)).numpy()
    full, _s, _p = U.igemm_conv(xs, kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=0)
    half, _s, pad = U.igemm_conv(xs, kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=0 | ffi.SEG_VARIANT_B_HI)
    want, _s, _p = U.igemm_conv(xs, kern_hi, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=0)
    assert pad == 0.0 and torch.equal(half, want)
    e = rel(half, full)
    assert 1e-6 < e < 5e-4

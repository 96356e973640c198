"""GPU parity tests in the parity mode (dtype='f32'): tensors are stored as split fp16 pairs (x = hi + lo / 2048, ~22
significant bits, csrc/seg_common.h) and every contraction is three fp16 MFMAs per product into fp32 accumulators.  The north
star's tolerance is asserted as written - every model output within 1e-3 of the oracle (fp32 torch-CPU restatement of the
reference) and masks (threshold 0.5 / channel arg-max, 2DCNN/Test.py:176) identical; measured forward errors are 1e-6 .. 5e-6,
so the tests hold a tighter 1e-4 as well.
Gradients: a ReLU whose input lies within the forward error of zero takes the other branch than the oracle's - measured on
`unet_w64_bce` (tools/dbg_f32_case.py): 1 .. 200 of 10^5 .. 10^6 activation-gradient elements per layer are off by their
whole value, everything else by 6e-6 relative; the flipped elements then spread through the data gradients below them.  Per
parameter tensor the relative L2 error is therefore bounded at 2e-2 (measured 2e-3 .. 8e-3 with flips, 1e-5 without) and no
element may be off by more than 5 % of the tensor's largest gradient: bounds a kernel defect cannot hide under, while the
arithmetic itself is held to 2e-6 .. 1e-5 at the operator level (the first two tests)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import gpu_util as U
from gpu_util import h16
from oracle import keras_ref as K
from test_gpu_models import CASES, masks

NORTH_STAR_TOL = 1e-3
NOISE_FLOOR = 1e-6       # absolute summation noise allowed on any gradient tensor, relative to the model's largest gradient norm
ZERO_FLOOR = 2e-4        # summation noise of a gradient that is mathematically zero, relative to the largest gradient norm of the model
ARITH_L2 = 2e-4          # relative L2 of a parameter gradient with the GPU run's branch decisions imposed on the oracle (arithmetic alone)
rng = np.random.default_rng(5)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("tile", [(128, 128), (64, 64), (128, 64), (256, 64)])
def test_conv_fwd_parity_mode_per_tap_kernels(tile):
    xs = [torch.randn(2, 16, 16, 24), torch.randn(2, 16, 16, 40)]
    kern = rng.standard_normal((3, 3, 64, 72)).astype(np.float32) * 0.1
    b = rng.standard_normal(72).astype(np.float32)
    out, st, pad = U.igemm_conv(xs, kern, b, "conv", (1, 1), "relu", 0.0, True, tile, f32=True)
    pre = K.conv2d_same(torch.cat(xs, -1), torch.from_numpy(kern), torch.from_numpy(b))
    assert rel(out, torch.relu(pre)) < 2e-6 and pad == 0.0
    assert rel(st[:, :72], torch.stack([pre.sum((0, 1, 2)), (pre * pre).sum((0, 1, 2))])) < 1e-5


@pytest.mark.parametrize("kind,k,s,shape,cin,cout", [("convT", (4, 4), (2, 2), (2, 8, 8), 40, 24), ("conv", (1, 3), (1, 1), (4, 1, 64), 16, 32),
                                                     ("conv", (1, 1), (2, 2), (2, 16, 16), 64, 8), ("convT", (1, 2), (1, 2), (4, 1, 32), 32, 16)])
def test_igemm_fp32_other_geometries(kind, k, s, shape, cin, cout):
    N, H, W = shape
    xs = [torch.randn(N, H, W, cin)]
    kshape = (*k, cout, cin) if kind == "convT" else (*k, cin, cout)
    kern = rng.standard_normal(kshape).astype(np.float32) * 0.1
    out, _, _ = U.igemm_conv(xs, kern, None, kind, s, None, 0.0, False, None, f32=True)
    f = K.conv2d_transpose_same if kind == "convT" else K.conv2d_same
    assert rel(out, f(xs[0], torch.from_numpy(kern), None, s)) < 2e-6
    kz = torch.zeros(kshape, requires_grad=True)
    y = f(xs[0], kz, None, s)
    dz = torch.randn_like(y)
    (y * dz).sum().backward()
    got = U.igemm_wgrad(xs, dz, kshape, kind, s, 1, 2, None, f32=True)
    assert rel(torch.from_numpy(got), kz.grad) < 1e-5


@pytest.mark.parametrize("name", sorted(CASES))
def test_model_parity_fp32_mode(name):
    torch.manual_seed(7)
    model, x, targets, loss = CASES[name]()
    model.build(x.shape[0], dtype="f32")
    r = U.compare_model(model, x, targets, loss, emulate_fp16=False, impose_decisions=True)
    for e, m in zip(r["outs"], r["out_mag"]):
        assert e < NORTH_STAR_TOL                              # the north-star bound, as written
        assert e < 1e-4 * max(1.0, m)                          # what the kernels actually deliver
    assert abs(r["loss"][0] - r["loss"][1]) < 1e-5 * max(1.0, abs(r["loss"][1]))
    assert max(e / max(1.0, m) for _n, e, m in r["nodes"]) < 1e-4
    gn = max(n for _e, n in r["gl2"].values())
    # gradients, two readings.  (1) ARITHMETIC: against the oracle run with the GPU's branch decisions imposed (ReLU signs,
    # pool arg-maxes, gate knees - a pre-activation within the forward error of zero takes either branch legitimately): every
    # parameter tensor to ARITH_L2 relative L2.  (2) as the oracle decides on its own: the flipped elements carry their whole
    # value into everything upstream - the looser bound, with the number of flipped decisions held small.
    worst = 0.0
    for k, (e, n) in r["gl2_imposed"].items():
        if k in r["zero_grad_params"]:      # conv bias in front of BatchNormalization: an exactly-zero gradient, summation noise on both sides
            assert e < ZERO_FLOOR * gn, (k, e, gn)
            continue
        # (NOISE_FLOOR x the model's largest gradient norm: a gradient that is a small difference of large sums - the gamma of a
        # BatchNormalization that feeds another one - carries summation noise that is not relative to its own size)
        if n > 1e-2 * gn:
            worst = max(worst, e / n)
        assert e < ARITH_L2 * n + NOISE_FLOOR * gn, (k, e, n, gn)
    flipped = sum(a for a, _b in r["flips"].values())
    total = sum(b for _a, b in r["flips"].values())
    print(f"[{name}] gradient rel-L2, decisions imposed: worst {worst:.2e}; flipped decisions {flipped} of {total}; "
          f"as decided by the oracle: worst {max(e / (n + 1e-3 * gn) for e, n in r['gl2'].values()):.2e}")
    assert flipped <= 1e-3 * total
    for k, (e, n) in r["gl2"].items():
        assert e < 2e-2 * (n + 1e-3 * gn), (k, e, n)
    assert U.grad_rel_err(r, floor=1e-3) < 5e-2


@pytest.mark.parametrize("name", ["unet_softmax_cce", "unet_w64_bce", "unet1d_softmax_cce", "unetpp_ds_mse", "unet_21class_softmax_cce"])
def test_masks_bit_exact_fp32_mode(name):
    torch.manual_seed(3)
    model, x, targets, loss = CASES[name]()
    model.build(x.shape[0], dtype="f32")
    for _ in range(2):
        model.train_step(x, targets, loss)
    w = {k: torch.from_numpy(v) for k, v in model.get_weights().items()}
    from ir_interp import run_graph
    ref = run_graph(model.graph, x, w, training=False)[0]
    p = model.predict(x)
    got = torch.from_numpy(p if len(model.output_names) == 1 else p[0])
    assert float((got - ref).abs().max()) < 1e-4
    assert torch.equal(masks(got), masks(ref))               # bit-exact masks


@pytest.mark.parametrize("name,dtype,tol", [("unetpp_ds_mse", "f32", 5e-6), ("unet_w64_bce", "f32", 5e-6), ("unetpp_ds_mse", "f16", 3e-3)])
def test_fused_inference_equals_the_three_pass_form(name, dtype, tol, monkeypatch):
    """model.predict runs conv -> BatchNormalization(moving statistics) -> ReLU as ONE launch (the affine folded into the GEMM
    epilogue, SegModel ConvOp.fwd_args_infer); SEG_FUSE_INFERENCE_BN=0 keeps the conv / bn_finalize / bn_act_apply passes.  Same
    weights, same input: the two agree to rounding (the fused form skips the storage rounding of the pre-BN tensor)."""
    M = U.M
    torch.manual_seed(3)
    model, x, targets, loss = CASES[name]()
    model.build(x.shape[0], dtype=dtype)
    for _ in range(2):
        model.train_step(x, targets, loss)                 # moving statistics away from their initial values
    assert any(getattr(op, "fwd_args_infer", None) is not None for op in model.ops)
    fused = model.predict(x)
    w = model.get_weights()
    monkeypatch.setattr(M.F, "FUSE_INFERENCE_BN", False)
    m2, _x, _t, _l = CASES[name]()
    m2.build(x.shape[0], dtype=dtype)
    m2.set_weights(w)
    assert all(getattr(op, "fwd_args_infer", None) is None for op in m2.ops)
    plain = m2.predict(x)
    fused, plain = (fused if isinstance(fused, list) else [fused]), (plain if isinstance(plain, list) else [plain])
    for a, b in zip(fused, plain):
        assert float(np.abs(a - b).max()) < tol * max(1.0, float(np.abs(b).max()))


def test_range_of_the_split_pair_format_is_loud_not_silent():
    """The parity mode's storage (csrc/seg_common.h, split_t): ~22 significant bits inside fp16's range, an absolute error floor
    of ~2^-36 below 2^-14, and NO silent garbage above 65504 - a value that does not fit becomes Inf / NaN in the output (and a
    non-finite gradient is skipped and counted by the optimiser kernels).  Large-but-representable and tiny operands, measured:"""
    kern = rng.standard_normal((3, 3, 32, 64)).astype(np.float32)
    # (a) operands of magnitude 3e4 (representable), weights 1e-4: results ~1e1 .. 1e2, full precision
    x = torch.randn(2, 16, 32, 32) * 1e4
    out, _, _ = U.igemm_conv([x], kern * 1e-4, None, "conv", (1, 1), None, 0.0, False, (256, 64), f32=True, variant=3)
    ref = K.conv2d_same(x, torch.from_numpy(kern * 1e-4), None)
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < 1e-5
    # (b) an input element beyond fp16's range: it becomes Inf in the stored operand, every output it touches is non-finite
    x2 = x.clone()
    x2[0, 8, 16, 3] = 1.0e5
    out2, _, _ = U.igemm_conv([x2], kern * 1e-4, None, "conv", (1, 1), None, 0.0, False, (256, 64), f32=True, variant=3)
    bad = ~torch.isfinite(out2)
    assert bool(bad[0, 7:10, 15:18].all()) and int(bad.sum()) == 9 * 64
    # (c) tiny operands (1e-6): below 2^-14 the pair keeps an absolute error of ~2^-36 per element, i.e. ~1e-5 relative here
    x3 = torch.randn(2, 16, 32, 32) * 1e-6
    out3, _, _ = U.igemm_conv([x3], kern, None, "conv", (1, 1), None, 0.0, False, (256, 64), f32=True, variant=3)
    ref3 = K.conv2d_same(x3, torch.from_numpy(kern), None)
    assert rel(out3, ref3) < 2e-4


@pytest.mark.parametrize("tile", [(256, 64), (256, 32)])
@pytest.mark.parametrize("case", [
    dict(kind="conv", shape=(2, 16, 32), cins=[24, 40], cout=72),       # 8x32 patches, virtual concat, two N blocks (the second ragged)
    dict(kind="conv", shape=(2, 32, 16), cins=[64], cout=64),           # 16x16 patches
    dict(kind="convT", shape=(2, 16, 32), cins=[40], cout=24),          # four sub-pixel classes of 2x2 taps
    dict(kind="conv", shape=(2, 16, 32), cins=[8, 17], cout=26),        # MultiResUNet widths: one ragged 32-channel tile
    dict(kind="conv", shape=(1, 16, 16), cins=[51], cout=8),
], ids=lambda c: f"{c['kind']}-{c['shape']}-{c['cins']}->{c['cout']}")
def test_halo_kernels_in_the_parity_mode(case, tile):
    """variant 3 with split-pair tensors (256 x 64 tile; 256 x 32: one 32-channel tile per wave, what pick_fwd takes for layers of at
    most 32 output channels - here also on wider ones): bias, ReLU / LeakyReLU, BN partial statistics, accumulate.
    (A two-patches-per-block form - 8 waves against one weight stream, 512 x 64 - was built and measured in round 3: correct, but
    31.9 instead of 30.4 ms over the forward / data-gradient launches of a C3 step and 54.1 instead of 50.9 ms per step: one
    barrier domain of 8 waves loses more than the halved weight traffic gains.  Not kept.)"""
    N, H, W = case["shape"]
    xs = [torch.randn(N, H, W, c) for c in case["cins"]]
    cin, cout = sum(case["cins"]), case["cout"]
    tconv = case["kind"] == "convT"
    kshape = (4, 4, cout, cin) if tconv else (3, 3, cin, cout)
    kern = rng.standard_normal(kshape).astype(np.float32) * 0.1
    b = rng.standard_normal(cout).astype(np.float32)
    s = (2, 2) if tconv else (1, 1)
    f = K.conv2d_transpose_same if tconv else K.conv2d_same
    pre = f(torch.cat(xs, -1), torch.from_numpy(kern), torch.from_numpy(b), s)
    act, alpha = ("leaky_relu", 0.3) if tconv else ("relu", 0.0)
    out, st, pad = U.igemm_conv(xs, kern, b, case["kind"], s, act, alpha, True, tile, f32=True, variant=3)
    want = torch.nn.functional.leaky_relu(pre, 0.3) if tconv else torch.relu(pre)
    assert rel(out, want) < 2e-6 and pad == 0.0
    assert rel(st[:, :cout], torch.stack([pre.sum((0, 1, 2)), (pre * pre).sum((0, 1, 2))])) < 1e-5
    if not tconv:
        base = torch.randn(N, H, W, U.r8(cout))
        out2, _, _ = U.igemm_conv(xs, kern, None, "conv", s, None, 0.0, False, tile, accumulate_into=base, f32=True, variant=3)
        assert rel(out2, f(torch.cat(xs, -1), torch.from_numpy(kern), None, s) + base[..., :cout]) < 2e-6


@pytest.mark.parametrize("f32", [True, False], ids=["parity", "fp16"])
@pytest.mark.parametrize("case", [
    dict(kind="conv", k=3, shape=(1, 8, 16), cins=[256, 160], cout=136, tile=(128, 64), sk=3),      # deep level: M = 128, K = 3 744
    dict(kind="conv", k=3, shape=(2, 8, 8), cins=[512], cout=72, tile=(128, 128), sk=4),
    dict(kind="conv", k=1, shape=(2, 16, 16), cins=[96, 40], cout=64, tile=(256, 64), sk=2),
    dict(kind="convT", k=4, shape=(1, 8, 8), cins=[192], cout=96, tile=(128, 64), sk=5),            # four classes; 6 chunks over 5 slices: an empty slice
], ids=lambda c: f"{c['kind']}{c['k']}-{c['shape']}-{c['cins']}->{c['cout']}-sk{c['sk']}")
def test_split_reduction_of_the_per_tap_kernel(case, f32):
    """seg_igemm_fwd(_f32) with splitk > 1 (variant 0, LDS-DMA kernel): slices of the K chunks leave accumulator images in the
    workspace (NaN-filled here: no initialisation needed), a second launch adds them in order and runs the epilogue - bias,
    activation, BN partial statistics, accumulate - with the same result as the unsplit launch, bit for bit between runs."""
    N, H, W = case["shape"]
    xs = [torch.randn(N, H, W, c) for c in case["cins"]]
    cin, cout, k = sum(case["cins"]), case["cout"], case["k"]
    tconv = case["kind"] == "convT"
    kern = rng.standard_normal((k, k, cout, cin) if tconv else (k, k, cin, cout)).astype(np.float32) * 0.05
    b = rng.standard_normal(cout).astype(np.float32)
    s = (2, 2) if tconv else (1, 1)
    kw = dict(f32=f32, variant=0)
    ref, st0, _ = U.igemm_conv(xs, kern, b, case["kind"], s, "relu", 0.0, True, case["tile"], **kw)
    out, st, pad = U.igemm_conv(xs, kern, b, case["kind"], s, "relu", 0.0, True, case["tile"], splitk=case["sk"], **kw)
    out2, _, _ = U.igemm_conv(xs, kern, b, case["kind"], s, "relu", 0.0, True, case["tile"], splitk=case["sk"], **kw)
    assert torch.equal(out, out2) and pad == 0.0
    tol = 2e-6 if f32 else 2e-3
    assert rel(out, ref) < tol and rel(st, st0) < 1e-5
    if f32:
        f = K.conv2d_transpose_same if tconv else K.conv2d_same
        want = torch.relu(f(torch.cat(xs, -1), torch.from_numpy(kern), torch.from_numpy(b), s))
        assert rel(out, want) < 5e-6          # (K = 4 608: the fp32 CPU reference's own summation order is in this figure)
    if not tconv:
        base = torch.randn(N, H, W, U.r8(cout))
        acc, _, _ = U.igemm_conv(xs, kern, None, "conv", s, None, 0.0, False, case["tile"], accumulate_into=base, splitk=case["sk"], **kw)
        acc0, _, _ = U.igemm_conv(xs, kern, None, "conv", s, None, 0.0, False, case["tile"], accumulate_into=base, **kw)
        assert rel(acc, acc0) < tol


@pytest.mark.parametrize("shape,cins,cout,sk,partial", [
    ((2, 16, 32), [24, 40], 72, 3, True),       # 4x16 patches, two concat segments, two row tiles (the second ragged)
    ((2, 16, 32), [8, 17], 26, 2, True),        # <= 32 output channels: the four waves split the taps (NARROW)
    ((1, 8, 32), [51], 8, 1, False),            # ... one patch range, accumulating into dw
    ((2, 8, 8), [40], 17, 2, True),             # 8x8 patches
    ((1, 6, 32), [64], 32, 1, False),           # 2x32 patches
])
def test_halo_weight_gradient_in_the_parity_mode(shape, cins, cout, sk, partial):
    """Variant 2 on a 3x3 / stride-1 conv with split-pair tensors: 64 output channels x one 32-channel plane x all taps per block,
    waves = (row tile) x (tap half); at most 32 output channels: waves = four tap groups, tap 8 shared by K blocks."""
    N, H, W = shape
    xs = [torch.randn(N, H, W, c) for c in cins]
    kshape = (3, 3, sum(cins), cout)
    kz = torch.zeros(kshape, requires_grad=True)
    y = K.conv2d_same(torch.cat(xs, -1), kz, None)
    dz = torch.randn_like(y)
    (y * dz).sum().backward()
    got = U.igemm_wgrad(xs, dz, kshape, "conv", (1, 1), 2, sk, None, f32=True, partial=partial, prefill=0.0 if partial else 0.25)
    assert rel(torch.from_numpy(got), kz.grad + (0.0 if partial else 0.25)) < 1e-5


@pytest.mark.parametrize("shape,cin,cout,sk", [
    ((2, 16, 32), 40, 24, 1), ((2, 8, 32), 24, 32, 2),                  # <= 32 output channels: one class per block, one tap per wave
    ((1, 16, 16), 64, 72, 3), ((2, 16, 32), 40, 40, 2),                 # class pair per block, 2x16 patches (ragged planes / row tiles)
    ((2, 8, 8), 96, 64, 5),                                             # ... 4x8 patches, more patch ranges than patches
    ((1, 6, 32), 33, 72, 2), ((1, 2, 64), 32, 64, 1), ((3, 2, 32), 64, 48, 1),      # ... 1x32 patches; one / three patches per block
])
def test_transposed_conv_weight_and_bias_gradient_from_one_launch(shape, cin, cout, sk):
    """Parity mode, variant 2 on Conv2DTranspose(4x4, stride 2): the two classes of a class row per block against one halo tile
    (igemm_wgrad_halo_split_tc2_kernel: 32-pixel patches, 3-stage ring, waves = row tile x tap row), one class per block for at most
    32 output channels; the blocks of the first input plane also contract dz with a ones operand - the bias gradient leaves the
    launch as partial rows per class and patch range (seg_bn_bwd_finalize's layout).  Empty patch ranges write zero rows."""
    N, H, W = shape
    x = torch.randn(N, H, W, cin)
    kshape = (4, 4, cout, cin)
    kz = torch.zeros(kshape, requires_grad=True)
    bz = torch.zeros(cout, requires_grad=True)
    y = K.conv2d_transpose_same(x, kz, bz, (2, 2))
    dz = torch.randn_like(y)
    (y * dz).sum().backward()
    got, bias = U.igemm_wgrad([x], dz, kshape, "convT", (2, 2), 2, sk, None, f32=True, bias_parts=True, partial=sk > 1)
    assert rel(torch.from_numpy(got), kz.grad) < 1e-5
    assert rel(bias[:cout], bz.grad) < 1e-5


@pytest.mark.parametrize("what,f32,N", [("fwd", True, 100), ("fwd", False, 100), ("wgrad", True, 72), ("fwd_beyond", True, 124)])
def test_operands_between_1_and_2_gib(what, f32, N):
    """The LDS-DMA kernels address an operand through one buffer resource with 32-bit offsets: up to seg_lds_dma_operand_limit()
    (1.875 GiB; 1 GiB until round 4 - UNet++ w64 at 256 x 256 stops fitting at batch 64 there).  A 3x3 layer at 256 x 256 whose
    input is 1.1 - 1.6 GiB: the halo kernels' first and last images against the CPU convolution (the last image sits at the highest
    offsets); the weight gradient with an input that is zero except in those images, so that a mis-addressed read loses their
    contribution (and picks up none: dz is dense).  `fwd_beyond`: 1.94 GiB - variant 0 falls through to the register-staged kernel."""
    torch.manual_seed(3)
    H = W = 256
    cin = 64 if f32 else 128                       # 256 bytes per pixel in either storage format
    cout = 16
    kern = (torch.randn(3, 3, cin, cout) * 0.05).numpy()
    ends = [0, 1, N - 2, N - 1]
    if what in ("fwd", "fwd_beyond"):
        x = torch.randn(N, H, W, cin)
        assert (N * H * W * 256 >= U.M.LDS_DMA_LIMIT) == (what == "fwd_beyond") and N * H * W * 256 > (1 << 30)
        tile, variant = ((256, 32 if f32 else 64), 3) if what == "fwd" else ((128, 64), 0)
        out, _, _ = U.igemm_conv([x], kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=f32, variant=variant)
        xs = x[ends] if f32 else h16(x[ends])
        want = K.conv2d_same(xs, torch.from_numpy(kern) if f32 else h16(torch.from_numpy(kern)), None, (1, 1))
        assert rel(out[ends], want) < (1e-5 if f32 else 2e-3)
        return
    x = torch.zeros(N, H, W, cin)
    x[ends] = torch.randn(4, H, W, cin)
    dz = torch.randn(N, H, W, 64)                # dz: 64 channels x 4 bytes = the same 256 bytes per pixel
    kz = torch.zeros(3, 3, cin, 64, requires_grad=True)
    (K.conv2d_same(x[ends], kz, None) * dz[ends]).sum().backward()
    got = U.igemm_wgrad([x], dz, (3, 3, cin, 64), "conv", (1, 1), 2, None, None, f32=True, partial=True)
    assert rel(torch.from_numpy(got), kz.grad) < 1e-5


def test_planner_routes_oversized_operands_to_the_register_staged_kernels(monkeypatch):
    """With the limit lowered to one byte every launch of a small model takes the route of an operand beyond 1.875 GiB (variant 0, tiles of
    at most 128 rows, no split; per-tap weight gradient): same loss, gradients and predictions as the default plan, to rounding."""
    import importlib
    P = importlib.import_module(U.M.__name__.rsplit(".", 1)[0] + ".planning")
    res = []
    for limit in (None, 1):
        if limit is not None:
            monkeypatch.setattr(P, "LDS_DMA_LIMIT", limit)
        torch.manual_seed(5)
        m, x, targets, loss = CASES["unetpp_ds_mse"]()
        m.build(x.shape[0], dtype="f32")
        if res:
            m.set_weights(res[0][3])
            assert all(op.fwd_variant == 0 and op.wg_args.variant != 2 for op in m.ops if isinstance(op, U.M.ConvOp))
        else:
            assert any(op.wg_args.variant == 2 for op in m.ops if isinstance(op, U.M.ConvOp))
        w = m.get_weights()
        m.forward(x, training=True)
        lv = float(m.compute_loss(targets, loss).cpu())
        m.backward(None)
        g = m.get_grads()
        m._join_side()
        res.append((lv, g, m.predict(x), w))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[0][0])
    for k in res[0][1]:
        assert rel(torch.as_tensor(res[1][1][k]), torch.as_tensor(res[0][1][k])) < 2e-4, k
    for a, b in zip(res[0][2], res[1][2]):
        assert float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max()) < 1e-5


@pytest.mark.parametrize("name,row_limit", [("multiresunet_ag_bce", None), ("multiresunet1d_mse", None), ("multiresunet_ag_bce", "narrowest")])
def test_fused_bn_tails_equal_the_separate_passes(name, row_limit, monkeypatch):
    """MultiResBlock / ResPath tails BN(ReLU(a + b)) as fused passes (SegModel._fuse_bn_tail) against the separate add / BN passes
    (SEG_FUSE_BN_TAILS=0): the same training step (loss, every parameter gradient) and the same predictions (inference-mode BN: the
    folded BatchNormalization takes its coefficients from the moving statistics) to rounding."""
    M = U.M
    torch.manual_seed(5)
    m1, x, targets, loss = CASES[name]()
    m1.build(x.shape[0], dtype="f32")
    assert any(isinstance(o, M.BNOp) and o.fused_add is not None for o in m1.ops)
    if row_limit is not None:
        # tails wider than seg_tail_row_limit() (2 048 channels; MultiResUNet at model_width 128, depth 5) take the separate passes
        # while the narrower ones of the same model stay fused - emulated by lowering the planner's mirror of the limit to this
        # model's narrowest tail
        cs_min = min(o.x.cs for o in m1.ops if isinstance(o, M.BNOp) and o.fused_add is not None)
        monkeypatch.setattr(M.planning, "TAIL_ROW_LIMIT", cs_min)
        torch.manual_seed(5)
        m1, x, targets, loss = CASES[name]()
        m1.build(x.shape[0], dtype="f32")
        fused = [o.x.cs for o in m1.ops if isinstance(o, M.BNOp) and o.fused_add is not None]
        assert fused and max(fused) == cs_min and any(isinstance(o, M.AddOp) and o.fused_bn is None and len(o.xs) > 1 for o in m1.ops)
    w = m1.get_weights()
    monkeypatch.setattr(M.F, "FUSE_BN_TAILS", False)
    m0, _x, _t, _l = CASES[name]()
    m0.build(x.shape[0], dtype="f32")
    m0.set_weights(w)
    assert not any(isinstance(o, M.BNOp) and o.fused_add is not None for o in m0.ops)
    res = []
    for m in (m1, m0):
        m.forward(x, training=True)
        lv = float(m.compute_loss(targets, loss).cpu())
        m.backward(None)
        g = m.get_grads()
        m.optimizer_step()
        m._join_side()
        res.append((lv, g, m.predict(x)))
    (l1, g1, p1), (l0, g0, p0) = res
    assert abs(l1 - l0) < 1e-6 * max(1.0, abs(l0))
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    for k in g0:
        assert float(np.abs(g1[k] - g0[k]).max()) < 2e-4 * (float(np.abs(g0[k]).max()) + 1e-3 * gmax), k
    p1, p0 = (p1 if isinstance(p1, list) else [p1]), (p0 if isinstance(p0, list) else [p0])
    for a, b in zip(p1, p0):
        assert float(np.abs(a - b).max()) < 1e-5 * max(1.0, float(np.abs(b).max()))


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("f16", 2e-2)])
def test_split_reductions_in_a_training_step(dtype, tol, monkeypatch):
    """A wide, small-image UNet (256 .. 2 048 channels at 32x32 .. 4x4) puts its deep levels on split reductions
    (model.plan_fwd) - forward convs with BN statistics, transposed convs (four classes), data gradients with several output
    spans and fused activation backward.  Same training step and predictions as with SEG_FWD_SPLITK=0 up to summation order."""
    from test_gpu_models import B, _with_targets
    M = U.M
    torch.manual_seed(11)
    x = torch.rand(2, 32, 32, 3)
    m1, _x, targets, loss = _with_targets(B("UNet", 32, 32, 256, 3, num_channels=3, ds=1), x, "mse")
    m1.build(2, dtype=dtype)
    split = [a for a in m1.fwd_ws_users if a.splitk > 1]
    assert any(a.nout > 1 for a in split) and any(a.g.ncls == 4 for a in split) and any(a.stats for a in split)
    w = m1.get_weights()
    monkeypatch.setattr(M.F, "FWD_SPLITK", False)
    m0 = B("UNet", 32, 32, 256, 3, num_channels=3, ds=1)
    m0.build(2, dtype=dtype)
    assert not any(a.splitk > 1 for a in m0.fwd_ws_users)
    m0.set_weights(w)
    res = []
    for m in (m1, m0):
        m.forward(x, training=True)
        lv = float(m.compute_loss(targets, loss).cpu())
        m.backward(None)
        g = m.get_grads()
        m.optimizer_step()
        m._join_side()
        res.append((lv, g, m.predict(x)))
    (l1, g1, p1), (l0, g0, p0) = res
    assert abs(l1 - l0) < tol * max(1.0, abs(l0))
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    for k in g0:
        assert float(np.abs(g1[k] - g0[k]).max()) < 10 * tol * (float(np.abs(g0[k]).max()) + 1e-3 * gmax), k
    for a, b in zip(p1, p0):
        assert float(np.abs(a - b).max()) < tol * max(1.0, float(np.abs(b).max()))

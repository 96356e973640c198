"""CPU: host logic of the engine in dry-run mode (buffers on the CPU, kernel calls only recorded - nothing is
computed): lowering / fusion decisions, the static backward schedule, parameter layout round trips, geometry."""
import importlib
from collections import Counter

import numpy as np
import pytest
import torch

import segamd

M = importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd.model")


@pytest.fixture(autouse=True)
def dry():
    M._DryRun.on = True
    M._DryRun.log.clear()
    yield
    M._DryRun.on = False


def B(*a, **k):
    return segamd.unet_model_builder(*a, train_mode="from_scratch", **k).ResNet50()


def step(m, x):
    M._DryRun.log.clear()
    m.train_step(x, [torch.rand(*h.y.shape) for h in m.out_heads])
    return Counter(M._DryRun.log)


def test_unetpp_fusion_counts():
    d = 3
    m = B("UNetPP", 32, 32, 16, d, ds=1).build(2, dtype="f16")
    c = step(m, torch.rand(2, 32, 32, 3))
    nodes = d * (d + 1) // 2
    convs = (d + 1) + 2 + nodes                      # encoder + latent(2) + nested nodes, every one conv+BN+ReLU fused
    # the d + 1 one-channel heads (deep-supervision levels + output) ride on the bn_act_apply pass of the block they read - no
    # seg_head_fwd launch - and their data gradient is injected into that block's BN-backward passes (seg_head_bwd: dw / db only)
    assert c["seg_bn_act_apply"] + c["seg_bn_act_apply_head"] == convs and c["seg_bn_finalize"] == convs
    # (level d reads X00 = the first encoder block, whose apply pass also pools: the same launch)
    assert c["seg_bn_act_apply_head"] == d + 1 and c["seg_head_fwd"] == 0 and c["seg_head_bwd"] == d + 1
    assert c["seg_bn_act_bwd_reduce_inj"] == d + 1 and c["seg_bn_act_bwd_apply_inj"] == d + 1
    assert sum(1 for h in m.heads if h.inject and h.acc is None) == d + 1
    assert c["seg_igemm_wgrad"] == convs + nodes     # + one transposed conv per node
    assert c["seg_maxpool_bwd"] == d                 # pooling fused into bn_act_apply forward; only the backward is a launch
    assert c["seg_maxpool_fwd"] == 0
    assert c["seg_adam_step_ex"] == 1                   # ONE flat optimiser launch
    # no concat is ever materialised: every tensor in the registry is a real layer output
    assert not any(n.endswith("/cat") for n in m.tensors)
    # data-gradient launches: ONE per conv (split output), none for the first layer (the image needs no gradient)
    assert c["seg_igemm_fwd"] == 2 * (convs + nodes) - 1


def test_predict_is_one_launch_per_conv_block():
    """Inference (model.predict, 2DCNN/Test.py:153,161): BatchNormalization with the moving statistics is folded into the conv's
    epilogue (seg_bn_fold_inference once per weight change -> ep_scale / bias of seg_igemm_fwd), ReLU applied there: one launch per
    Conv_Block (unet_variants.py:7-14) plus the 2x2 pooling of the encoder levels - no pre-BN tensor pass, no normalisation pass."""
    d = 3
    for dtype in ("f16", "f32"):
        m = B("UNetPP", 32, 32, 16, d, ds=1).build(2, dtype=dtype)
        x = torch.rand(2, 32, 32, 3)
        M._DryRun.log.clear()
        m.forward(x, training=False)
        c = Counter(n.replace("_f32", "") for n in M._DryRun.log)
        nodes = d * (d + 1) // 2
        convs = (d + 1) + 2 + nodes
        assert c["seg_bn_act_apply"] == 0 and c["seg_bn_finalize"] == 0 and c["seg_bn_stats"] == 0
        assert c["seg_igemm_fwd"] == convs + nodes and c["seg_bn_fold_inference"] == convs and c["seg_maxpool_fwd"] == d
        M._DryRun.log.clear()
        m.forward(x, training=False)                  # the folded coefficients are cached until the weights change
        c = Counter(n.replace("_f32", "") for n in M._DryRun.log)
        assert c["seg_bn_fold_inference"] == 0 and c["seg_igemm_fwd"] == convs + nodes
        step(m, x)
        M._DryRun.log.clear()
        m.forward(x, training=False)
        assert Counter(M._DryRun.log)["seg_bn_fold_inference"] == convs


def test_tails_wider_than_the_kernels_row_limit_take_the_separate_passes(monkeypatch):
    """The fused tail kernels handle ONE column group of 8 channels per thread (seg_tail_row_limit() = 2048 channels); a MultiResBlock /
    ResPath tail wider than that (model_width 128 at depth 5: ~3 400 channels - the reference takes any width,
    unet_variants.py:85-122) must be lowered to the un-fused passes instead of raising in the first forward().  Checked by lowering the
    limit below this small model's widths: every tail - BN(cat) fold, lazy inputs, virtual concatenation - falls back, none is half-fused."""
    planning = importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd.planning")
    def lowered(limit):
        monkeypatch.setattr(planning, "TAIL_ROW_LIMIT", limit)
        m = B("MultiResUNet", 32, 32, 16, 2, alpha=1.0).build(2, dtype="f32")
        c = Counter(n.replace("_f32", "") for n in step(m, torch.rand(2, 32, 32, 3)).elements())
        return m, c
    m_all, c_all = lowered(2048)
    widths = sorted({o.x.cs for o in m_all.ops if isinstance(o, M.BNOp) and o.fused_add is not None})
    assert len(widths) >= 2
    m, c = lowered(widths[0])                               # only the narrowest tails still fit
    fused = [o for o in m.ops if isinstance(o, M.BNOp) and o.fused_add is not None]
    assert fused and all(o.x.cs <= widths[0] for o in fused)
    assert 0 < c["seg_add_act_stats_fwd"] < c_all["seg_add_act_stats_fwd"] and c["seg_add_act_fwd"] > c_all["seg_add_act_fwd"]
    m, c = lowered(widths[0] - 8)                           # nothing fits: the round-2 lowering, no tail kernel at all
    assert not any(isinstance(o, M.BNOp) and (o.fused_add is not None or o.folded_into is not None or o.pieces is not None) for o in m.ops)
    assert not any(isinstance(o, M.ConvOp) and o.lazy for o in m.ops)
    assert c["seg_add_act_stats_fwd"] == 0 and c["seg_bn_bwd_apply_fanout"] == 0 and c["seg_cat_bn_bwd_scatter"] == 0
    assert c["seg_add_act_fwd"] > 0 and c["seg_add_act_bwd"] > 0


def test_multires_tails_are_fused_passes(monkeypatch):
    """BN(ReLU(a + b)) - the tail of a MultiResBlock and of every ResPath iteration (unet_variants.py:97-99, :118-120): the add
    kernel applies the ReLU, normalises a BatchNormalization input on the fly (BN(cat) is never written) and delivers the statistics
    of the BN that follows; that BN's backward writes the masked gradient to the add's inputs.  Against the unfused lowering: no
    seg_add_act_fwd / seg_add_act_bwd, one seg_bn_stats and one seg_bn_act_apply fewer per fused BN / folded BN.
    Round 4 on top (each behind its own switch): the Conv_Blocks that feed only the add are normalised ON LOAD there (no
    seg_bn_act_apply for them; their BN-backward sums come out of the tail's fan-out launch, no reduce pass), cat[c3, c5, c7] is
    never materialised (no seg_copy_channels, its statistics from one gathering
    launch, its backward = finalize + seg_cat_bn_bwd_scatter), and the add's gradient is written once."""
    def counts():
        m = B("MultiResUNet", 32, 32, 16, 2, alpha=1.0).build(2, dtype="f16")
        c = step(m, torch.rand(2, 32, 32, 3))
        tails = sum(1 for o in m.ops if isinstance(o, M.BNOp) and o.fused_add is not None)
        folded = sum(1 for o in m.ops if isinstance(o, M.BNOp) and o.folded_into is not None)
        lazy = sum(1 for o in m.ops if isinstance(o, M.ConvOp) and o.lazy)
        virt = sum(1 for o in m.ops if isinstance(o, M.BNOp) and o.pieces is not None)
        return c, tails, folded, lazy, virt, m
    for sw in ("TAIL_LAZY_INPUTS", "TAIL_SHARED_GRAD", "TAIL_VIRTUAL_CAT", "TAIL_LAZY_SUMS", "TAIL_PAIR_APPLY", "TAIL_PIECE_STATS"):
        monkeypatch.setattr(M.F, sw, False)
    c1, tails, folded, lazy, virt, _m = counts()                                # the round-3 form
    assert tails > 0 and folded > 0 and lazy == 0 and virt == 0
    assert c1["seg_add_act_stats_fwd"] == tails and c1["seg_bn_bwd_apply_fanout"] == tails
    monkeypatch.setattr(M.F, "FUSE_BN_TAILS", False)
    c0, t0, f0, _l, _v, _m = counts()
    assert t0 == 0 and f0 == 0 and c0["seg_add_act_stats_fwd"] == 0
    assert c0["seg_add_act_fwd"] - c1["seg_add_act_fwd"] == tails and c0["seg_add_act_bwd"] - c1["seg_add_act_bwd"] == tails
    assert c0["seg_bn_stats"] - c1["seg_bn_stats"] == tails                    # statistics from the add kernel
    assert c0["seg_bn_act_apply"] - c1["seg_bn_act_apply"] == folded           # BN(cat) applied inside the add
    assert c0["seg_bn_act_bwd_apply"] - c1["seg_bn_act_bwd_apply"] == tails    # replaced by the fan-out form
    # ---- round 4
    monkeypatch.setattr(M.F, "FUSE_BN_TAILS", True)
    for sw in ("TAIL_LAZY_INPUTS", "TAIL_SHARED_GRAD", "TAIL_VIRTUAL_CAT", "TAIL_LAZY_SUMS", "TAIL_PAIR_APPLY", "TAIL_PIECE_STATS"):
        monkeypatch.setattr(M.F, sw, True)
    c2, tails2, folded2, lazy, virt, m = counts()
    assert (tails2, folded2) == (tails, folded) and virt == folded            # every BN(cat) of this model is a virtual concatenation
    assert lazy == 2 * (tails - folded) + folded                                # ResPath: shortcut + conv; MultiResBlock: the shortcut
    # (the three pieces of every virtual concatenation run their apply pass in the tail kernel's one-input form, which also delivers
    # their columns of BN(cat)'s statistics: no separate pass over the pieces)
    assert c1["seg_bn_act_apply"] - c2["seg_bn_act_apply"] == lazy + 3 * folded
    assert c2["seg_copy_channels"] == 0 and c1["seg_copy_channels"] == 2 * 3 * folded      # three pieces, forward and backward
    assert c2["seg_bn_stats"] == c1["seg_bn_stats"] - folded and c2["seg_add_act_stats_fwd"] == tails + 3 * folded
    assert c2["seg_cat_bn_bwd_scatter"] == folded
    # BN(cat)'s sums ride on the fan-out launch, and so do those of the Conv_Blocks normalised on load (their dOut is written there)
    assert c1["seg_bn_act_bwd_reduce"] - c2["seg_bn_act_bwd_reduce"] == folded + lazy
    assert c1["seg_bn_bwd_finalize"] == c2["seg_bn_bwd_finalize"]
    # ... its input gradient is the scatter, and the two blocks of a ResPath tail get their dz from ONE pass over the shared gradient
    pairs = tails - folded
    assert c1["seg_bn_act_bwd_apply"] - c2["seg_bn_act_bwd_apply"] == folded + 2 * pairs and c2["seg_bn_act_bwd_apply2"] == pairs
    assert not any(isinstance(o, M.ConcatOp) and not o.virtual for o in m.ops)
    # the add's gradient is written once: every second input reads the first one's buffer
    shared = [o for o in m.ops if isinstance(o, M.ConvOp) and o.grad_src is not None]
    assert len(shared) == tails - folded and all(o.lazy for o in shared)


def test_accumulate_flags_follow_consumer_count():
    m = B("UNetPP", 32, 32, 16, 2, ds=0).build(2, dtype="f16")
    step(m, torch.rand(2, 32, 32, 3))
    acc = {}
    for op in m.ops:
        if isinstance(op, M.ConvOp) and getattr(op, "live", False):
            for a, _zero, _fl in op.dg_args:
                for o, (t, _c) in enumerate(op.segs):
                    if a.out_ptr[o]:
                        acc.setdefault(t.name, []).append(a.out_acc[o])
    assert acc["X01/relu"] == [0]                       # one consumer (X02's dense skip): written once
    assert len(acc["enc1/relu"]) == 2                   # skip into X01 and X02 (+ the pooled path through maxpool_bwd)
    # per tensor: at most one plain write, every other producer accumulates
    assert all(sorted(v)[1:] == [1] * (len(v) - 1) for v in acc.values())


@pytest.mark.parametrize("mk", [lambda: B("UNet", 32, 32, 16, 2, num_channels=1, lstm=1, dense_loop=2),
                                lambda: B("UNetPP", 32, 32, 16, 3, ds=1, ag=1),
                                lambda: B("UNet3P", 32, 32, 8, 2, ds=1),
                                lambda: segamd.UNet1D(64, 2, 1, 16, 3, ds=1, ag=1).UNet(),
                                lambda: segamd.BCDUNet(64, 2, 1, 16, 3, lstm=1, dense_loop=2).BCDUNet(),
                                # ConvLSTM filter counts that are not multiples of 8 (4, 12 / 3): gate blocks padded to 8 in the engine
                                lambda: B("UNet", 16, 16, 4, 2, num_channels=5, lstm=1),
                                lambda: B("UNetP", 6, 2, 6, 1, num_channels=3, ds=1, lstm=1, dense_loop=2)])
def test_weight_layout_roundtrip(mk):
    m = mk().build(2, dtype="f16")
    w = m.get_weights()
    specs = {p.name: p.shape for p in m.graph.param_specs()}
    assert {k: v.shape for k, v in w.items()} == specs
    rng = np.random.default_rng(0)
    w2 = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in w.items()}
    m.set_weights(w2)
    w3 = m.get_weights()
    assert all(np.array_equal(w2[k], w3[k]) for k in w2)
    assert m.count_params() == sum(int(np.prod(s)) for s in specs.values())


def test_geometry_tables():
    g = M.geom_conv(2, 16, 16, 3, 3, 1, 1)
    assert (g.ncls, g.ntaps, g.T_total) == (1, 9, 9)
    assert [(g.tap_dy[i], g.tap_dx[i]) for i in range(9)] == [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    t = M.geom_convT(2, 8, 8, 4, 4, 2, 2)              # Y = 2i + ky - 1  
    assert (t.ncls, t.ntaps, t.Hout, t.osy) == (4, 4, 16, 2)
    for c in range(4):
        py = t.oy[c]
        for k in range(4):
            e = c * 4 + k
            ky = t.tap_w[e] // 4
            assert 2 * t.tap_dy[e] + ky - 1 == py          # i_in = i + dy  <=>  2 i + py = 2 (i + dy) + ky - 1
    d = M.geom_convT_dgrad(2, 8, 8, 4, 4, 2, 2)
    assert (d.Hin, d.isy, d.ntaps) == (16, 2, 16) and sorted(set(d.tap_dy[i] for i in range(16))) == [-1, 0, 1, 2]
    t1 = M.geom_convT(2, 1, 32, 1, 2, 1, 2)            # Conv1DTranspose k2 s2: y = 2i + k
    assert (t1.ncls, t1.ntaps) == (2, 1) and [t1.tap_dx[i] for i in range(2)] == [0, 0]
    s2 = M.geom_conv(2, 16, 16, 1, 1, 2, 2)            # 1x1 stride 2 'same': no padding, samples 0,2,4,...
    assert (s2.Hl, s2.isy, s2.tap_dy[0]) == (8, 2, 0)


def test_pack_gemm_weight_places_segments():
    kern = np.arange(3 * 3 * 5 * 4, dtype=np.float32).reshape(3, 3, 5, 4)
    E = M.pack_gemm_weight(kern, [2, 3], [0, 32], 8, 64, False)
    assert E.shape == (8, 9, 64)
    assert E[1, 4, 0] == kern[1, 1, 0, 1] and E[1, 4, 33] == kern[1, 1, 3, 1]
    assert E[4:].sum() == 0 and E[:, :, 2:32].sum() == 0 and E[:, :, 35:].sum() == 0
    back = M.unpack_gemm_weight(E.reshape(-1), kern.shape, [2, 3], [0, 32], 8, 64, False)
    assert np.array_equal(back, kern)


def test_unsupported_layers_say_so():
    m = B("UNet", 32, 32, 8, 2, ae=1, feature_number=16)             # the autoencoder bottleneck lowers to two 1x1-image GEMMs
    m.build(2, dtype="f16")
    kinds = [type(o).__name__ for o in m.ops]
    assert kinds.count("ReshapeOp") == 2 and m.store.slots["ae/features/kernel"].spec.shape == (8 * 8 * 32, 16)
    # ... also when the flattened sample H*W*C is not a multiple of 8 (until round 5: NotImplementedError): one copy per sample then
    m = B("MultiResUNet", 8, 8, 6, 2, num_channels=3, ae=1, feature_number=24, alpha=1.67, is_transconv=False).build(3, dtype="f32")
    rs = [o for o in m.ops if type(o).__name__ == "ReshapeOp"]
    assert len(rs) == 2 and rs[0].per_sample and rs[0].hwc == 156 and not rs[1].per_sample and rs[1].hwc % 8 == 0
    c = step(m, torch.rand(3, 8, 8, 3))
    assert c["seg_copy_channels_f32"] == 2 * 3 + 2           # forward and backward: one launch per sample for the padded row, one otherwise
    # a ConvLSTM whose filter count is not a multiple of 8 lowers too (gate blocks padded to 8 channels in the engine; until round 5
    # this raised NotImplementedError although the reference takes any model_width, unet_variants.py:148)
    m = B("UNet", 32, 32, 12, 2, lstm=1).build(2, dtype="f16")
    lstm = [o for o in m.ops if isinstance(o, M.ConvOp) and o.kind == "convlstm"]
    assert lstm and all(o.F8 == (o.F + 7) // 8 * 8 and o.cout == 4 * o.F8 and o.cout_logical == 4 * o.F for o in lstm)
    assert any(o.z.gate_pad is not None for o in lstm)
    # what Keras refuses at construction is refused the same way: MultiResUNet with alpha * model_width < 6 asks for 0 filters
    with pytest.raises(ValueError, match="filters"):
        B("MultiResUNet", 24, 32, 6, 3, num_channels=1, alpha=0.5).build(2, dtype="f16")


def test_parity_mode_tiles_have_a_kernel_instance():
    """pick_fwd in the parity mode (split-pair tensors) returns only what csrc has an instance for: the halo tile is 256 x 64
    (two accumulator sets per wave rule out 128 output channels) or, for at most 32 output channels, 256 x 32; everything else the
    per-tap kernels' tiles."""
    g = M.geom_conv(16, 256, 256, 3, 3, 1, 1)
    tiles0 = {(256, 64), (128, 128), (128, 64), (64, 128), (64, 64)}
    seen = set()
    for Mtot in (1024, 4096, 16384, 65536, 262144, 1048576):
        for n_rows in (8, 32, 40, 64, 128, 384, 2048):
            for ktot in (288, 1152, 18432):
                tm, tn, variant = M.pick_fwd(Mtot, n_rows, g, ktot, False)
                assert (variant == 3 and (tm, tn) == ((256, 32) if n_rows <= 32 else (256, 64))) or (variant == 0 and (tm, tn) in tiles0)
                seen.add((tm, tn, variant))
    assert {(256, 64, 3), (256, 32, 3)} <= seen and any(v == 0 for _a, _b, v in seen)       # large levels: halo tiles; small grids: per-tap kernels
    g1 = M.geom_conv(16, 256, 256, 1, 1, 1, 1)                               # 1x1: never the halo kernel
    assert all(M.pick_fwd(Mt, n, g1, 512, False)[2] == 0 for Mt in (4096, 1048576) for n in (8, 64, 384))


def test_under_filled_grids_split_the_reduction():
    """plan_fwd: forward / data-gradient grids of at most one block per CU are split along K towards two blocks per CU (128-row tiles
    of the per-tap LDS-DMA kernel, seg_igemm_fwd_args.splitk) while a slice keeps >= 3 072 of the reduction - the 8x8 / 16x16
    levels of the depth-5 configurations; the large levels and short reductions are left alone.  The slices' accumulator images
    share ONE workspace (every such launch runs on the main stream)."""
    g8, g16, g256 = M.geom_conv(16, 8, 8, 3, 3, 1, 1), M.geom_conv(16, 16, 16, 3, 3, 1, 1), M.geom_conv(16, 256, 256, 3, 3, 1, 1)
    assert M.plan_fwd(1024, 1024, g8, 9 * 2048, False) == (128, 128, 0, 4)         # enc6 data gradient: 64 blocks, slices of 4 608
    assert M.plan_fwd(1024, 2048, g8, 9 * 2048, False) == (128, 128, 0, 4)         # latent convs: 128 blocks
    assert M.plan_fwd(4096, 512, g16, 9 * 1024, False) == (128, 128, 0, 2)         # enc5 data gradient (otherwise the halo kernel at 128 blocks)
    assert M.plan_fwd(4096, 1024, g16, 9 * 512, False)[3] == 1                     # enc5 forward: 256 blocks, K = 4 608 - no gain measured
    assert M.plan_fwd(1048576, 64, g256, 9 * 384, False) == (256, 64, 3, 1)        # the large levels: halo tiles, no split
    assert M.plan_fwd(1024, 1024, g8, 9 * 2048, True) == (128, 128, 0, 4)          # throughput mode: the same rule
    m = segamd.unet_model_builder("UNetPP", 256, 256, 64, 5, ds=1, train_mode="from_scratch").ResNet50().build(2, dtype="f32")
    split = [a for a in m.fwd_ws_users if a.splitk > 1]
    assert split and all(a.variant == 0 and a.tile_m == 128 for a in split)
    assert max(M.fwd_workspace_bytes(a) for a in split) > 0


def test_the_backward_mode_auto_resolves_by_chain_length_and_structure():
    """SegModel.build(backward="auto") (flags.py): two MFMAs per product in the data gradients (b2) only for a plain Conv_Block stack of
    at most 32 convolutions in a chain; attention gates, MultiResBlock / ResPath tails (a normalisation behind an add / a concatenation)
    and deep chains keep their data gradients at three MFMAs (d3w2).  The launches carry the matching variant bits."""
    ffi = importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd._ffi")
    cases = [
        (B("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1), True, "b2w2"),
        (B("UNet", 64, 64, 16, 3, num_channels=1, lstm=1), True, "b2w2"),
        (B("UNet3P", 64, 64, 16, 3, num_channels=1), True, "b2w2"),
        (B("UNet", 64, 64, 16, 3, num_channels=1, ag=1), False, "d3w2"),               # multiplicative gate
        (B("MultiResUNet", 64, 64, 16, 3, num_channels=1), False, "d3w2"),           # BN(ReLU(a + b)), BN(cat)
    ]
    for m, plain, want in cases:
        assert m.graph.plain_conv_bn_stack() == plain and m.graph.longest_conv_chain() <= 32
        m.build(2, dtype="f32", backward="auto")
        assert m.backward_mode == "auto" and m.backward_spec == want
        convs = [op for op in m.ops if isinstance(op, M.ConvOp)]
        halo_w = [op for op in convs if (op.wg_args.variant & 255) == 2 and op.n_rows > 32]
        assert all(op.wg_args.variant & ffi.SEG_VARIANT_IN_HI and not op.wg_args.variant & ffi.SEG_VARIANT_DZ_HI for op in halo_w)
        dg = [a for op in convs for a, _z, _f in op.dg_args] + [g.args for g in m.gathers]
        assert all(not (a.variant & ffi.SEG_VARIANT_A_HI) for a in dg)                # dz is never the halved operand of the default
        if want == "d3w2":
            assert all(not (a.variant & ffi.SEG_VARIANT_B_HI) for a in dg)
    # a chain longer than the limit (a UNet of depth 8 has 28 convolutions in a row: lower the limit below that for the check)
    deep = B("UNet", 256, 256, 4, 8, num_channels=1)
    assert deep.graph.plain_conv_bn_stack() and deep.graph.longest_conv_chain() == 28
    keep = M.F.BACKWARD_AUTO_CHAIN
    try:
        M.F.BACKWARD_AUTO_CHAIN = 27
        deep.build(1, dtype="f32", backward="auto")
        assert deep.backward_spec == "d3w2"
    finally:
        M.F.BACKWARD_AUTO_CHAIN = keep
    # explicit modes and the throughput mode
    m = B("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1)
    m.build(2, dtype="f32", backward="3mfma")
    assert m.backward_spec == "d3w3" and (m.bwd_d, m.bwd_w) == (3, 3)
    m = B("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1)
    m.build(2, dtype="f16", backward="auto")
    assert (m.bwd_d, m.bwd_w) == (3, 3)                                                # nothing to halve: one MFMA per product anyway
    with pytest.raises(ValueError):
        B("UNetPP", 64, 64, 16, 3, num_channels=3, ds=1).build(2, dtype="f32", backward="fast")

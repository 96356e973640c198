"""Host logic (no GPU): the graph builders reproduce the reference's API surface, structure tables
 and - numerically, via the oracle's independent wiring - its layer graph."""
import pytest
import torch

from oracle.keras_ref import Layers
from oracle.models_ref import unet2d_forward, unet1d_forward, bcdunet1d_forward
from ir_interp import run_graph


def g2(seg, *a, **k):
    return seg.unet_model_builder(*a, train_mode="from_scratch", **k).graph()


# ---- SURVEY Appendix A structure tables (BN counted once: the survey's totals add the BN values twice,
# its own per-layer rows sum to the figures asserted here; MAC totals match the survey exactly)
def test_c2_unet_structure(seg):
    g = g2(seg, "UNet", 256, 256, 64, 5, num_channels=3, output_nums=4, final_activation="softmax")
    assert g.count_params() == 170_570_500 - 40_448
    assert g.macs() == 45_898_268_672
    convs = [n for n in (g.nodes[i] for i in g.live_nodes()) if n.kind == "conv"]
    tconvs = [n for n in (g.nodes[i] for i in g.live_nodes()) if n.kind == "convT"]
    assert len(convs) == 14 and len(tconvs) == 5
    assert g.output_names == ["out"] and g.nodes[g.outputs[0]].shape == (256, 256, 4)


def test_c3_unetpp_structure(seg):
    g = g2(seg, "UNetPP", 256, 256, 64, 5, num_channels=3, ds=1)
    assert g.macs() == 164_018_257_920
    nodes = [g.nodes[i] for i in g.live_nodes()]
    assert sum(n.kind == "conv" for n in nodes) == 29 and sum(n.kind == "convT" for n in nodes) == 15
    assert g.output_names == ["out", "level1", "level2", "level3", "level4", "level5"]
    assert all(g.nodes[i].shape == (256, 256, 1) for i in g.outputs)          # UNetPP DS heads are full-res
    x05 = next(n for n in nodes if n.name == "X05/cat")
    assert x05.shape == (256, 256, 384)
    assert [g.nodes[i].name for i in x05.inputs] == ["X05/tconv_act", "X01/relu", "X02/relu", "X03/relu",
                                                       "X04/relu", "enc1/relu"]
    bn = sum(4 * n.shape[2] for n in nodes if n.kind == "bn")
    assert g.count_params() == 198_907_014 - bn


def test_c4_c5_c1_structure(seg):
    g = g2(seg, "MultiResUNet", 512, 512, 32, 5, num_channels=3, ag=1, alpha=1.0)
    assert g.macs() == 66_994_246_656
    nodes = [g.nodes[i] for i in g.live_nodes()]
    assert sum(n.kind == "conv" for n in nodes) == 94 and sum(n.kind == "convT" for n in nodes) == 10
    g = g2(seg, "UNet", 256, 256, 64, 4, num_channels=1, lstm=1)
    assert g.macs() == 61_782_097_920
    nodes = [g.nodes[i] for i in g.live_nodes()]
    assert sum(n.kind == "convlstm" for n in nodes) == 4
    g = seg.UNet1D(1024, 3, 1, 16, 3, "Classification", output_nums=2, ds=0).graph_UNet()
    assert g.count_params() == 171_586 and g.macs() == 37_568_512
    assert g.nodes[g.outputs[0]].shape == (1, 1024, 2)


def test_unet_ds_levels_are_low_res(seg):
    g = g2(seg, "UNet", 64, 64, 8, 3, ds=1)
    shapes = [g.nodes[i].shape for i in g.outputs]
    assert shapes == [(64, 64, 1), (32, 32, 1), (16, 16, 1), (8, 8, 1)]       # out, level1..3 (tapped before upsampling)


# ---- API surface / error behaviour (unet_variants.py:1036-1048, 1D :257-258)
def test_errors(seg):
    with pytest.raises(ValueError, match="Train Mode"):
        seg.unet_model_builder("UNet", 64, 64, 8, 3, train_mode="bogus")
    with pytest.raises(ValueError, match="cannot be less than 1"):
        seg.unet_model_builder("UNet", 64, 64, 8, 0, train_mode="from_scratch")
    with pytest.raises(ValueError, match="from 1 to 5"):
        seg.unet_model_builder("UNet", 64, 64, 8, 6)
    with pytest.raises(ValueError, match="Input Parameters"):
        seg.unet_model_builder("UNet", 0, 64, 8, 3, train_mode="from_scratch").graph()
    with pytest.raises(ValueError, match="Input Parameters"):
        seg.UNet1D(0, 3, 1, 16, 3).graph_UNet()
    for enc in seg.ENCODER_METHODS:
        assert callable(getattr(seg.unet_model_builder, enc))
    assert len(seg.ENCODER_METHODS) == 33
    ms = seg.model_selector("UNet", "resnet50", "UNetPP", 32, 32, 8, 2, train_mode="from_scratch")
    assert ms.graph().name == "ResNet50_UNetPP"
    assert seg.UNet1D(128, 2, 1, 8, 3).D_S == 1                               # 1D default ds=1


# ---- wiring equals the oracle's independent wiring, numerically
CASES_2D = [
    dict(decoder_name="UNet", w=8, d=3, kw=dict(output_nums=4, final_activation="softmax")),
    dict(decoder_name="UNet", w=8, d=3, kw=dict(ds=1, ag=1)),
    dict(decoder_name="UNet", w=8, d=2, kw=dict(lstm=1, dense_loop=2)),
    dict(decoder_name="UNet", w=8, d=2, kw=dict(is_transconv=False, ae=1, feature_number=16)),
    dict(decoder_name="UNetPP", w=8, d=3, kw=dict(ds=1)),
    dict(decoder_name="UNetPP", w=4, d=3, kw=dict(ag=1, lstm=1)),
    dict(decoder_name="UNetE", w=8, d=3, kw=dict(ds=1)),
    dict(decoder_name="UNetP", w=8, d=3, kw=dict(ag=1)),
    dict(decoder_name="UNet3P", w=8, d=3, kw=dict(ds=1)),
    dict(decoder_name="MultiResUNet", w=16, d=2, kw=dict(ag=1, alpha=1.0)),
    dict(decoder_name="MultiResUNet", w=12, d=3, kw=dict(ds=1, alpha=1.67)),
]


@pytest.mark.parametrize("case", CASES_2D, ids=lambda c: c["decoder_name"] + "-" + "-".join(f"{k}{v}" for k, v in c["kw"].items()))
def test_2d_wiring_matches_oracle(seg, case):
    torch.manual_seed(0)
    w, d, kw = case["w"], case["d"], case["kw"]
    g = g2(seg, case["decoder_name"], 32, 32, w, d, num_channels=3, **kw)
    L = Layers(2, seed=3)
    x = torch.rand(2, 32, 32, 3)
    L.update_moving = False
    ref = unet2d_forward(L, x, case["decoder_name"], w, d, **kw)
    specs = {p.name: p.shape for p in g.param_specs()}
    assert specs == {k: tuple(v.shape) for k, v in L.params.items()}
    got = run_graph(g, x, L.params)
    assert len(got) == len(ref) == len(g.outputs)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5), float((a - b).abs().max())


@pytest.mark.parametrize("method,kw", [("UNet", dict(ds=0, problem_type="Classification", output_nums=2)),
                                       ("UNet", dict(ds=1, ag=1, lstm=1)),
                                       ("UNetPP", dict(ds=1)), ("UNetPP", dict(ds=0, ag=1, is_transconv=False)),
                                       ("MultiResUNet", dict(ds=1, alpha=1.0)), ("UNet", dict(ae=1, feature_number=8))])
def test_1d_wiring_matches_oracle(seg, method, kw):
    b = seg.UNet1D(64, 2, 1, 16, 3, **kw)
    g = getattr(b, "graph_" + method)()
    L = Layers(1, seed=5)
    L.update_moving = False
    x = torch.rand(2, 64, 1)
    ref = unet1d_forward(L, x, method, 2, 16, 3, **kw)
    assert {p.name: p.shape for p in g.param_specs()} == {k: tuple(v.shape) for k, v in L.params.items()}
    got = run_graph(g, x, L.params)
    for a, c in zip(got, ref):
        assert torch.allclose(a, c, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("kw", [dict(lstm=1, dense_loop=3), dict(lstm=0, ds=0, dense_loop=1)])
def test_bcdunet_wiring_matches_oracle(seg, kw):
    g = seg.BCDUNet(64, 2, 1, 8, 3, **kw).graph_BCDUNet()
    L = Layers(1, seed=7)
    L.update_moving = False
    x = torch.rand(2, 64, 1)
    ref = bcdunet1d_forward(L, x, 2, 8, 3, **kw)
    assert {p.name: p.shape for p in g.param_specs()} == {k: tuple(v.shape) for k, v in L.params.items()}
    got = run_graph(g, x, L.params)
    for a, c in zip(got, ref):
        assert torch.allclose(a, c, atol=1e-5, rtol=1e-5)

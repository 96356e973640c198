"""MI355X: a Keras checkpoint FILE of the reference's builder call goes into the engine without TensorFlow (SegModel.load_weights on a
`.keras` / `.h5` path: keras_file.py + h5lite.py), runs, trains, and goes back out as a checkpoint of the same shape."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import segamd
from ir_interp import run_graph

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, HERE)
from make_keras_files import CASES  # noqa: E402

FILES = os.path.join(HERE, "keras_files")
PLANTED = np.load(os.path.join(FILES, "planted.npz"))


def _model(dec, kw):
    kw = dict(kw)
    if dec == "UNet1D":
        return segamd.UNet1D(**kw).UNet(), (3, kw["length"], kw["num_channel"])
    w, d = kw.pop("model_width"), kw.pop("model_depth")
    return segamd.unet_model_builder(dec, 32, 32, w, d, train_mode="from_scratch", **kw).ResNet50(), (3, 32, 32, kw["num_channels"])


@pytest.mark.parametrize("fname,dec,kw,layout", CASES, ids=[c[0] for c in CASES])
def test_checkpoint_file_runs_on_the_engine_and_goes_back_out(fname, dec, kw, layout, tmp_path):
    model, xshape = _model(dec, kw)
    model.build(xshape[0], dtype="f32")
    path = os.path.join(FILES, fname)
    cfg = open(path.replace(".h5", ".config.json")).read() if layout == "h5_weights" else None
    model.load_weights(path, config=cfg)
    want = {k.split("|", 1)[1].replace("|", "/"): v for k, v in PLANTED.items() if k.startswith(fname + "|")}
    got = model.get_weights()
    assert set(want) <= set(got)
    for k, v in want.items():
        assert np.array_equal(got[k].reshape(v.shape), v), k          # the fp32 master copy holds the file's numbers exactly
    # the fixture's numbers are standard normal everywhere (they only have to be told apart): variances made positive and kernels
    # scaled to their fan-in before the model runs, or ten layers of gain 6 leave the fp16 range of the storage format
    fixed = {k: (np.abs(v) + 0.5 if k.endswith("moving_variance") else v / np.sqrt(np.prod(v.shape[:-1])) if v.ndim >= 2 else v).astype(np.float32)
             for k, v in got.items()}
    model.set_weights(fixed)
    torch.manual_seed(3)
    x = torch.randn(*xshape)
    ref = run_graph(model.graph, x, {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in fixed.items()}, training=False)
    pred = model.predict(x.numpy())
    pred = pred if isinstance(pred, (list, tuple)) else [pred]
    for p_, r_ in zip(pred, ref):
        r_ = r_.detach().numpy()
        assert np.abs(np.asarray(p_).reshape(r_.shape) - r_).max() <= 1e-3 * max(1.0, np.abs(r_).max())
    out = str(tmp_path / ("trained_" + fname))
    model.save_keras_file(out, like=path, config=cfg)
    again, _ = _model(dec, kw)
    again.build(xshape[0], dtype="f32")
    again.load_weights(out, config=cfg)
    back = again.get_weights()
    assert all(np.array_equal(back[k], fixed[k].reshape(back[k].shape)) for k in fixed)


@pytest.mark.parametrize("fname,dec,kw,layout", CASES, ids=[c[0] for c in CASES])
def test_load_model_runs_a_checkpoint_without_the_constructor_call(fname, dec, kw, layout):
    """tf.keras.saving.load_model + predict (2D/Test.py:114, 153): the graph comes out of the file (keras_import.py), the engine lowers
    it like a builder's, and predict() matches the oracle run on the BUILDER's graph with the weights planted in the file."""
    path = os.path.join(FILES, fname)
    cfg = open(path.replace(".h5", ".config.json")).read() if layout == "h5_weights" else None
    model = segamd.load_model(path, config=cfg)
    assert not model.built
    builder, xshape = _model(dec, kw)
    want = {k.split("|", 1)[1].replace("|", "/"): torch.from_numpy(np.asarray(v, np.float32)) for k, v in PLANTED.items() if k.startswith(fname + "|")}
    torch.manual_seed(5)
    x = torch.randn(*xshape)
    ref = run_graph(builder.graph, x, want, training=False)
    pred = model.predict(x.numpy())                      # builds for this batch, takes the weights over
    assert model.built and model._weights_at_build is None
    pred = pred if isinstance(pred, (list, tuple)) else [pred]
    assert len(pred) == len(ref)
    for p_, r_ in zip(pred, ref):
        r_ = r_.detach().numpy()
        assert np.abs(np.asarray(p_).reshape(r_.shape) - r_).max() <= 1e-3 * max(1.0, np.abs(r_).max())
    # ... and it trains: one step moves the loss the way the oracle's graph says (same loss value within the parity bound)
    tg = [torch.full(tuple(r.shape), 0.5) for r in ref]
    loss0 = float(model.train_step(x, tg if len(tg) > 1 else tg[0], "mse"))
    o = run_graph(builder.graph, x, want, training=True)
    want_loss = sum(float(((a - 0.5) ** 2).mean()) for a in o)
    assert abs(loss0 - want_loss) <= 1e-3 * max(1.0, abs(want_loss))


def test_checkpoint_written_here_reloads_without_the_constructor_call(tmp_path):
    """ModelCheckpoint('x.keras') on this side, then Test.py's load_model + predict: the same predictions."""
    fname, dec, kw, _layout = CASES[0]
    model, xshape = _model(dec, kw)
    model.build(xshape[0], dtype="f32")
    torch.manual_seed(7)
    x = torch.randn(*xshape)
    tg = [torch.full((xshape[0],) + tuple(model.graph.nodes[o].shape), 0.5) for o in model.graph.outputs]
    for _ in range(3):
        model.train_step(x, tg, "mse")
    path = str(tmp_path / "UNet_32_Fold_1.keras")
    model.save_weights(path)
    want = model.predict(x.numpy())
    again = segamd.load_model(path)
    got = again.predict(x.numpy())
    for a, b in zip(want, got):
        assert np.abs(np.asarray(a) - np.asarray(b)).max() <= 1e-6
    fresh, _ = _model(dec, kw)
    fresh.build(xshape[0], dtype="f32")
    fresh.load_weights(path)
    w0, w1 = model.get_weights(), fresh.get_weights()
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)

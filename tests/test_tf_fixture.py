"""Reference-produced golden vectors (tests/golden/tf_*.npz, written on a TensorFlow box by tools/tf_capture_spec.py).

While no fixture is present every test here SKIPS - and the oracle stays "parity unpinned" (DESIGN.md section 3): this
container has no TensorFlow.  With fixtures present:
  * CPU  the oracle (oracle/models_ref.py) must reproduce the Keras outputs (predict: inference-mode BN; train_fwd:
         training-mode BN) within 1e-4 - that pins the oracle;
  * GPU  the HIP engine in f32 mode must reproduce them within 1e-3 with bit-exact masks - the north star, against the
         reference itself."""
import glob
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "tf_*.npz")))
needs_fixture = pytest.mark.skipif(not FIXTURES, reason="no TensorFlow-produced fixture (tests/golden/tf_*.npz): run tools/tf_capture_spec.py "
                                                        "on a TF 2.12-2.15 machine; parity stays unpinned until then")


def _load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    w = {k[2:].replace("|", "/"): z[k] for k in z.files if k.startswith("w/")}
    pred = [z[f"predict/{i}"] for i in range(sum(1 for k in z.files if k.startswith("predict/")))]
    trn = [z[f"train_fwd/{i}"] for i in range(sum(1 for k in z.files if k.startswith("train_fwd/")))]
    return meta, z["x"], w, pred, trn


def _masks(p):
    return (p >= 0.5) if p.shape[-1] == 1 else p.argmax(-1)[..., None]


@needs_fixture
@pytest.mark.parametrize("path", FIXTURES or ["<none>"])
def test_oracle_reproduces_the_reference(path):
    from oracle.keras_ref import Layers
    from oracle.models_ref import unet2d_forward
    meta, x, w, pred, trn = _load(path)
    L = Layers(2, seed=0)
    L.set_weights(w)
    L.update_moving = False
    xt = torch.from_numpy(x)
    for training, want in ((False, pred), (True, trn)):
        L.training = training
        with torch.no_grad():
            kw = {k: v for k, v in meta["kwargs"].items() if k != "num_channels"}        # (the oracle reads the channel count from x)
            got = unet2d_forward(L, xt, meta["decoder"], meta["width"], meta["depth"], **kw)
        assert len(got) == len(want)
        for g_, r in zip(got, want):
            assert float(np.abs(g_.numpy() - r).max()) < 1e-4, (meta["case"], training)
            assert np.array_equal(_masks(g_.numpy()), _masks(r))


@needs_fixture
@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES or ["<none>"])
def test_engine_reproduces_the_reference(path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import segamd
    meta, x, w, pred, trn = _load(path)
    m = segamd.unet_model_builder(meta["decoder"], meta["H"], meta["W"], meta["width"], meta["depth"], train_mode="from_scratch",
                                  **meta["kwargs"]).ResNet50()
    m.build(x.shape[0], dtype="f32")
    m.set_weights(w)
    got = m.predict(x)
    got = got if isinstance(got, list) else [got]
    for g_, r in zip(got, pred):
        assert float(np.abs(g_ - r).max()) < 1e-3, meta["case"]
        assert np.array_equal(_masks(g_), _masks(r))
    outs = [o.cpu().numpy() for o in m.forward(torch.from_numpy(x), training=True)]
    for g_, r in zip(outs, trn):
        assert float(np.abs(g_ - r).max()) < 1e-3, meta["case"]


def test_the_skip_is_visible_and_the_capture_spec_is_self_contained():
    """The capture script must not depend on anything TensorFlow-specific of THIS repo and must name every case the
    fixture tests would load."""
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "tools", "tf_capture_spec.py")).read()
    assert "keras_map.keras_weights_by_structural_name" in src and "/root/reference" not in src
    if not FIXTURES:
        pytest.skip("parity unpinned: no TensorFlow-produced fixture present (expected in this container)")

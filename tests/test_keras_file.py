"""CPU: Keras checkpoint FILES onto structural parameter names without TensorFlow (keras_file.py + h5lite.py, SURVEY.md section 8
row f3).  The fixtures (tests/golden/keras_files/, written through the real HDF5 library by tests/golden/make_keras_files.py) hold
small models of the reference's builder calls in every layout keras_file.py documents: `.keras` archives with class-counter keys
(Keras 2.13 - 3.x), layer-name keys and Keras 3's inbound_nodes spelling, TF 2.12's top group; the legacy whole-model `.h5`; a
weights-only `.h5` with the configuration beside it.  planted.npz says which array belongs to which structural name."""
import json
import os
import subprocess
import sys
import zipfile

import numpy as np
import pytest

import segamd
from segamd import h5lite as H
from segamd import keras_file as KF
from segamd import keras_map as KM

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, HERE)
from make_keras_files import CASES  # noqa: E402

FILES = os.path.join(HERE, "keras_files")
PLANTED = np.load(os.path.join(FILES, "planted.npz"))
CONDA = "/opt/conda/bin/python3.9"


def _graph(dec, kw):
    kw = dict(kw)
    if dec == "UNet1D":
        return segamd.UNet1D(**kw).graph_UNet()
    w, d = kw.pop("model_width"), kw.pop("model_depth")
    return segamd.unet_model_builder(dec, 32, 32, w, d, train_mode="from_scratch", **kw).graph()


def _config_for(fname, layout):
    return open(os.path.join(FILES, fname.replace(".h5", ".config.json"))).read() if layout == "h5_weights" else None


def _planted(fname):
    return {k.split("|", 1)[1].replace("|", "/"): v for k, v in PLANTED.items() if k.startswith(fname + "|")}


@pytest.mark.parametrize("fname,dec,kw,layout", CASES, ids=[c[0] for c in CASES])
def test_every_weight_of_the_file_lands_on_its_structural_name(fname, dec, kw, layout):
    g = _graph(dec, kw)
    got = KF.load_keras_file(os.path.join(FILES, fname), g, config=_config_for(fname, layout))
    want = _planted(fname)
    specs = {p.name: p for i in g.live_nodes() for p in g.nodes[i].params}
    assert set(got) == set(want) == set(specs)
    for k, v in want.items():
        assert got[k].shape == tuple(specs[k].shape) and np.array_equal(got[k].reshape(v.shape), v), k


def test_a_checkpoint_of_another_constructor_call_is_refused():
    fname, dec, kw, _layout = CASES[0]
    other = dict(kw, model_depth=3)
    with pytest.raises(KM.KerasMapError):
        KF.load_keras_file(os.path.join(FILES, fname), _graph(dec, other))
    with pytest.raises(KM.KerasMapError):                   # the deep-supervision heads are missing on the builder's side
        KF.load_keras_file(os.path.join(FILES, fname), _graph(dec, dict(kw, ds=0)))


def test_files_that_are_not_checkpoints_say_what_they_are(tmp_path):
    with pytest.raises(KF.KerasFileError, match="weights-only"):
        KF.KerasFileModel(os.path.join(FILES, "unet3p_weights_only.h5"))
    p = tmp_path / "x.keras"
    p.write_bytes(b"not a checkpoint at all" * 40)
    with pytest.raises(KF.KerasFileError, match="neither a .keras archive nor an HDF5 file"):
        KF.KerasFileModel(str(p))
    z = tmp_path / "y.keras"
    with zipfile.ZipFile(z, "w") as zf:
        zf.writestr("config.json", "{}")
    with pytest.raises(KF.KerasFileError, match="without model.weights.h5"):
        KF.KerasFileModel(str(z))
    w = H.H5Writer()
    w.create_dataset("something/else", np.zeros(3, np.float32))
    q = tmp_path / "z.h5"
    w.save(str(q))
    cfg = open(os.path.join(FILES, "unet3p_weights_only.config.json")).read()
    with pytest.raises(KF.KerasFileError, match="layer_names"):
        KF.KerasFileModel(str(q), config=cfg)


def test_class_names_snake_case_as_keras_names_them():
    for cls, want in (("Conv2D", "conv2d"), ("Conv2DTranspose", "conv2d_transpose"), ("BatchNormalization", "batch_normalization"),
                      ("MaxPooling2D", "max_pooling2d"), ("UpSampling2D", "up_sampling2d"), ("ConvLSTM2D", "conv_lstm2d"),
                      ("InputLayer", "input_layer"), ("LeakyReLU", "leaky_re_lu"), ("ReLU", "re_lu"), ("Conv1DTranspose", "conv1d_transpose")):
        assert KF._snake(cls) == want


@pytest.mark.parametrize("fname,dec,kw,layout", CASES, ids=[c[0] for c in CASES])
def test_trained_weights_go_back_into_a_checkpoint_of_the_same_shape(fname, dec, kw, layout, tmp_path):
    g = _graph(dec, kw)
    src = os.path.join(FILES, fname)
    cfg = _config_for(fname, layout)
    rng = np.random.default_rng(9)
    new = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in _planted(fname).items()}
    out = str(tmp_path / ("trained_" + fname))
    KF.save_keras_file(new, out, like=src, graph=g, config=cfg)
    back = KF.load_keras_file(out, g, config=cfg)
    assert set(back) == set(new) and all(np.array_equal(back[k].reshape(new[k].shape), new[k]) for k in new)
    a, b = KF.KerasFileModel(src, cfg), KF.KerasFileModel(out, cfg)
    assert [p for p, _d in a.h5.visit_datasets()] == [p for p, _d in b.h5.visit_datasets()]          # same tree, optimiser variables included
    if fname.endswith(".keras"):
        with zipfile.ZipFile(src) as za, zipfile.ZipFile(out) as zb:
            assert za.namelist() == zb.namelist() and za.read("config.json") == zb.read("config.json")
        assert int(b.h5["optimizer/vars/0"].read()) == 1234


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no /opt/conda python with h5py on this machine")
def test_the_real_hdf5_library_reads_the_rewritten_archive(tmp_path):
    fname, dec, kw, layout = CASES[0]
    g = _graph(dec, kw)
    new = {k: (v + 1.0).astype(np.float32) for k, v in _planted(fname).items()}
    out = str(tmp_path / "trained.keras")
    KF.save_keras_file(new, out, like=os.path.join(FILES, fname), graph=g)
    m = KF.KerasFileModel(out)
    paths = {p: w for l in m.layers for p, w in zip(l._paths, l._w)}
    np.savez(str(tmp_path / "exp.npz"), **{k.replace("/", "|"): v for k, v in paths.items()})
    script = f"""
import h5py, io, json, zipfile, numpy as np
exp = {{k.replace("|", "/"): v for k, v in np.load(r"{tmp_path}/exp.npz").items()}}
with zipfile.ZipFile(r"{out}") as z:
    cfg = json.loads(z.read("config.json"))
    raw = z.read("model.weights.h5")
with h5py.File(io.BytesIO(raw), "r") as f:
    for k, v in exp.items():
        assert f[k].dtype == v.dtype and np.array_equal(f[k][()], v), k
    assert f["optimizer/vars/0"][()] == 1234 and len(f["layers"]) == len(cfg["config"]["layers"])
print("ok", len(exp))
"""
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONHOME", "PYTHONPATH")}
    r = subprocess.run([CONDA, "-c", script], capture_output=True, text=True, env=env, timeout=300)
    if "No module named 'h5py'" in r.stderr:
        pytest.skip("the /opt/conda interpreter has no h5py here")
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok")


def test_safetensors_weight_files_round_trip(tmp_path):
    from segamd import harness
    load_weights_file, save_weights_file = harness.load_weights_file, harness.save_weights_file
    w = {k: v for k, v in _planted(CASES[1][0]).items()}
    for ext in (".safetensors", ".npz"):
        p = str(tmp_path / ("w" + ext))
        save_weights_file(w, p)
        back = load_weights_file(p)
        assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)


def test_one_loader_takes_every_weight_file_by_what_it_is(tmp_path):
    """The reference hands ONE path (`..._Fold_1.keras`) to ModelCheckpoint and to load_weights (Train.py:363-379): a Keras archive
    at that path is read as one; what this package's save_weights wrote for that path (path + '.npz') is found under it too."""
    from segamd import harness
    fname, dec, kw, _layout = CASES[1]
    g = _graph(dec, kw)
    want = _planted(fname)
    got = harness.load_weights_file(os.path.join(FILES, fname), g)
    assert all(np.array_equal(got[k].reshape(v.shape), v) for k, v in want.items())
    ckpt = str(tmp_path / "UNetPP_32_Fold_1.keras")
    harness.save_weights_file(want, ckpt)                 # (ModelCheckpoint -> SegModel.save_weights)
    assert not os.path.exists(ckpt) and os.path.exists(ckpt + ".npz")
    back = harness.load_weights_file(ckpt, g)             # (the resume branch of Train.py: load_weights of the same path)
    assert set(back) == set(want) and all(np.array_equal(back[k], want[k]) for k in want)
    with pytest.raises(ValueError, match="needs the builder graph"):
        harness.load_weights_file(os.path.join(FILES, fname))


def test_damaged_checkpoints_raise_kerasfileerror(tmp_path):
    import random
    rng = random.Random(3)
    for fname in ("unet1d.keras", "unet3p_weights_only.h5"):
        raw = open(os.path.join(FILES, fname), "rb").read()
        cfg = _config_for(fname, "h5_weights" if fname.endswith(".h5") else "keras")
        outcomes = set()
        for i in range(120):
            b = bytearray(raw)
            if i % 5 == 0:
                b = b[:rng.randrange(16, len(b))]
            else:
                for _m in range(rng.choice([1, 2, 8])):
                    pos = rng.randrange(0, len(b))
                    b[pos] = rng.choice([0, 0xFF, rng.randrange(256)])
            p = tmp_path / ("m_" + fname)
            p.write_bytes(bytes(b))
            try:
                KF.KerasFileModel(str(p), cfg)
                outcomes.add("ok")
            except KF.KerasFileError:
                outcomes.add("refused")
        assert "refused" in outcomes and outcomes <= {"ok", "refused"}, (fname, outcomes)          # (a zip member's CRC catches every flip)


def test_fresh_process_auto_names_do_not_fool_the_weight_lookup():
    """In a fresh process Keras names its layers conv2d, conv2d_1, ... in CREATION order and files their weights under conv2d, conv2d_1,
    ... in model.layers order - the same strings, another assignment.  Reading the keys as layer names would put weights into the wrong
    layers of equal shape without an error; the counted reading must win."""
    fname, dec, kw, layout = next(c for c in CASES if c[0] == "unet_fresh_names.keras")
    m = KF.KerasFileModel(os.path.join(FILES, fname))
    names = {l.name for l in m.layers}
    keys = set(m.h5["layers"].keys())
    assert {"conv2d", "conv2d_1", "batch_normalization", "batch_normalization_1"} <= names & keys          # the ambiguity is real in this file
    by_name_differs = [l.name for l in m.layers if l._paths and not l._paths[0].startswith(f"layers/{l.name}/")]
    assert by_name_differs                                 # ... and the two readings disagree for some layers
    got = KF.load_keras_file(os.path.join(FILES, fname), _graph(dec, kw))
    want = _planted(fname)
    assert set(got) == set(want) and all(np.array_equal(got[k].reshape(v.shape), v) for k, v in want.items())

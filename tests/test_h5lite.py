"""CPU: h5lite (the pure-Python HDF5 subset behind the Keras checkpoint files) against files written by
the REAL HDF5 library (tests/golden/h5/*, written by tests/golden/make_h5_files.py with h5py 3.3 / HDF5 1.10.6), and the writer the
other way round: the real library reads what H5Writer wrote when this image's /opt/conda h5py is there (skipped otherwise)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import segamd
from segamd import h5lite as H

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
EXPECTED = {k.replace("|", "/"): v for k, v in np.load(os.path.join(HERE, "expected.npz")).items()}
CONDA = "/opt/conda/bin/python3.9"
LONG_TEXT = "{" + ", ".join(f'"layer_{i}": "conv2d \u00e9"' for i in range(4000)) + "}"


@pytest.mark.parametrize("fname", ["default", "latest", "tracked"])
def test_every_dataset_of_the_library_written_files_reads_back(fname):
    with H.H5File(os.path.join(HERE, fname + ".h5")) as f:
        mine = {k.split(":", 1)[1]: v for k, v in EXPECTED.items() if k.startswith(fname + ":")}
        assert mine
        for path, want in mine.items():
            got = np.asarray(f[path].read())
            assert got.shape == want.shape, path
            if want.dtype == bool:
                got = got.astype(bool)                  # (h5py stores booleans as an int8 enum)
            assert got.dtype.newbyteorder("=") == want.dtype.newbyteorder("="), (path, got.dtype, want.dtype)
            assert np.array_equal(got, want), path
        if fname != "latest":                           # (its 20-link group is in dense storage: refused, next test)
            assert set(mine) <= {p for p, _d in f.visit_datasets()}


def test_layouts_datatypes_and_attributes_of_the_default_format():
    with H.H5File(os.path.join(HERE, "default.h5")) as f:
        assert f.O == 8 and f.L == 8
        assert list(f["fixed_strings"].read()) == [b"conv2d/kernel:0", b"conv2d/bias:0", b"x"]
        assert list(f["vlen_strings"].read()) == ["alpha", "", "gamma delta"]
        assert f.attrs["keras_version"] == "2.13.1" and f.attrs["backend"] == "tensorflow"
        assert list(f.attrs["layer_names"]) == [b"conv2d", b"batch_normalization", b"re_lu"]
        cfg = f.attrs["config"]                         # 76 KB: more than an object-header message holds - global heap
        assert cfg.startswith('{"k0": 0, "k1": 1') and cfg.endswith('"k5999": 5999}')
        assert f["grp"].attrs["number"] == 7 and np.array_equal(f["grp"].attrs["vector"], np.arange(5) / 4)
        assert f["f32"].attrs["note"] == "on a dataset"
        wide = f["wide"]                                 # 300 links: a two-level group B-tree
        assert len(wide) == 300 and wide.keys()[0] == "layer_0000" and wide.keys()[-1] == "layer_0299"
        assert all(float(wide[f"layer_{i:04d}"].read()) == i for i in (0, 7, 8, 123, 255, 256, 299))
        assert f["hollow/vars"].keys() == [] and "hollow" in f and "nothing" not in f and "soft" not in f
        assert f["f32"].shape == (3, 3, 5, 7) and f["f32"].dtype == np.float32 and f["grp/i32_be"].dtype == np.int32
        with pytest.raises(KeyError):
            f["grp/missing"]
        with pytest.raises(KeyError):
            f["f32/below_a_dataset"]


def test_what_is_outside_the_subset_is_refused_not_misread():
    with H.H5File(os.path.join(HERE, "latest.h5")) as f:
        assert f.attrs["title"] == "latest format" and f["a"].attrs["n"] == 3
        with pytest.raises(H.H5Error, match="dense storage"):
            f["dense"].keys()
    with pytest.raises(H.H5Error, match="not an HDF5 file"):
        H.H5File(b"PK\x03\x04 this is a zip archive, not HDF5" + b"\0" * 600)
    raw = open(os.path.join(HERE, "default.h5"), "rb").read()
    with pytest.raises(H.H5Error, match="truncated|past the end"):          # a file cut in the middle: loud
        cut = H.H5File(raw[:len(raw) // 3])
        for _p, d in cut.visit_datasets():
            d.read()


def _written(tmp_path):
    rng = np.random.default_rng(1)
    w = H.H5Writer()
    exp = {}
    for i in range(1100):                               # more than one symbol node's 1 024 entries in one group
        exp[f"layers/conv2d_{i}/vars/0"] = rng.standard_normal((3, 3, 2, 4)).astype(np.float32)
    exp["layers/bn/vars/0"] = np.arange(5, dtype=np.float64)
    exp["ints"] = np.arange(-3, 4, dtype=np.int32)
    exp["half"] = np.arange(4, dtype=np.float16)
    exp["scalar"] = np.asarray(np.float32(1.5))
    exp["empty"] = np.zeros((0, 3), np.float32)
    for k, v in exp.items():
        w.create_dataset(k, v)
    w.create_group("layers/activation/vars")
    w.set_attr("/", "keras_version", "2.13.1")
    w.set_attr("/", "layer_names", [b"conv2d", b"batch_normalization"])
    w.set_attr("layers", "count", np.int64(1101))
    w.set_attr("ints", "scale", np.float32(0.5))
    w.set_attr("/", "model_config", LONG_TEXT)           # 90 KB: a variable-length string in the global heap
    path = str(tmp_path / "written.h5")
    w.save(path)
    return path, exp


def test_writer_output_reads_back(tmp_path):
    path, exp = _written(tmp_path)
    with H.H5File(path) as f:
        for k, v in exp.items():
            got = np.asarray(f[k].read())
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
        assert f.attrs["keras_version"] == b"2.13.1" and list(f.attrs["layer_names"]) == [b"conv2d", b"batch_normalization"]
        assert f["layers"].attrs["count"] == 1101 and f["ints"].attrs["scale"] == 0.5
        assert f.attrs["model_config"] == LONG_TEXT and len(LONG_TEXT.encode()) > 65536
        assert f["layers/activation/vars"].keys() == [] and len(f["layers"]) == 1102
    w = H.H5Writer()
    w.create_dataset("a", np.zeros(3, np.float32))
    with pytest.raises(H.H5Error, match="exists"):
        w.create_dataset("a", np.zeros(3, np.float32))
    with pytest.raises(H.H5Error, match="in the way"):
        w.create_dataset("a/b", np.zeros(3, np.float32))
    with pytest.raises(H.H5Error, match="numeric"):
        w.create_dataset("s", np.array(["text"]))
    with pytest.raises(H.H5Error, match="64 KB"):
        w.set_attr("/", "big", [b"x" * 70, b"y"] * 600)
        w.tobytes()


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no /opt/conda python with h5py on this machine")
def test_the_real_hdf5_library_reads_what_the_writer_wrote(tmp_path):
    path, exp = _written(tmp_path)
    np.savez(str(tmp_path / "exp.npz"), **{k.replace("/", "|"): v for k, v in exp.items()})
    script = f"""
import h5py, numpy as np
exp = {{k.replace("|", "/"): v for k, v in np.load(r"{tmp_path}/exp.npz").items()}}
with h5py.File(r"{path}", "r") as f:
    for k, v in exp.items():
        assert f[k].dtype == v.dtype and f[k].shape == v.shape and np.array_equal(f[k][()], v), k
    assert f.attrs["keras_version"] == b"2.13.1" and list(f.attrs["layer_names"]) == [b"conv2d", b"batch_normalization"]
    assert f["layers"].attrs["count"] == 1101 and f["ints"].attrs["scale"] == 0.5
    text = f.attrs["model_config"]
    text = text.decode("utf-8") if isinstance(text, bytes) else text
    assert len(text) == {len(LONG_TEXT)} and text.startswith('{{"layer_0": "conv2d \u00e9"') and text.endswith('"}}')
    assert len(f["layers"]) == 1102 and list(f["layers/activation/vars"].keys()) == []
    seen = []
    f.visit(seen.append)
    assert len(seen) == 3 * 1100 + 1 + 3 + 2 + 4, len(seen)
print("libhdf5", h5py.version.hdf5_version, "ok")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONHOME", "PYTHONPATH")}
    r = subprocess.run([CONDA, "-c", script], capture_output=True, text=True, env=env, timeout=300)
    if "No module named 'h5py'" in r.stderr:
        pytest.skip("the /opt/conda interpreter has no h5py here")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ok" in r.stdout


def test_history_file_of_the_training_script(tmp_path):
    """Train.py:417-430 dumps model_history.history with h5py, one dataset per key: the same file without h5py, and back."""
    from segamd import harness
    hist = {"loss": [0.9, 0.5, 0.31], "val_loss": [1.0, 0.7, 0.6], "lr": [2e-4, 2e-4, 2e-5], "out_mean_io_u": [0.1, 0.3, 0.35]}
    p = str(tmp_path / "UNet_Fold_1_History.h5")
    harness.save_history_h5(hist, p)
    assert harness.load_history_h5(p) == hist
    if os.path.exists(CONDA):
        script = f"import h5py\nwith h5py.File(r'{p}', 'r') as f:\n    print(sorted(f.keys()), list(f['lr'][()]), f['loss'].dtype)"
        env = {k: v for k, v in os.environ.items() if k not in ("PYTHONHOME", "PYTHONPATH")}
        r = subprocess.run([CONDA, "-c", script], capture_output=True, text=True, env=env, timeout=120)
        if "No module named 'h5py'" not in r.stderr:
            assert r.returncode == 0, r.stderr[-1000:]
            assert r.stdout.strip() == "['loss', 'lr', 'out_mean_io_u', 'val_loss'] [0.0002, 0.0002, 2e-05] float64"


def test_damaged_files_raise_h5error_whatever_the_parser_trips_over():
    """A truncated download or a flipped bit in an offset, a length, a signature: 600 mutations of the library-written files (and of a
    checkpoint's weights member) either still read or raise H5Error - no other exception type, no endless B-tree / link cycle, no
    petabyte allocation (a longer campaign of 12 000 mutations found: IndexError / TypeError / struct.error / UnicodeDecodeError /
    MemoryError / RecursionError leaks, a link with an empty name that made a group its own child - all H5Error now)."""
    import random
    import zipfile
    raws = [open(os.path.join(HERE, fn + ".h5"), "rb").read() for fn in ("default", "latest", "tracked")]
    with zipfile.ZipFile(os.path.join(os.path.dirname(HERE), "keras_files", "unet1d.keras")) as z:
        raws.append(z.read("model.weights.h5"))

    def read_all(raw):
        f = H.H5File(raw)

        def rec(g):
            g.attrs
            for k in g.keys():
                x = g[k]
                if isinstance(x, H.Group):
                    rec(x)
                else:
                    x.attrs
                    x.read()
        rec(f.root)

    rng = random.Random(7)
    ok = refused = 0
    for _ in range(600):
        raw = bytearray(rng.choice(raws))
        if rng.random() < 0.15:
            raw = raw[:rng.randrange(8, len(raw))]
        else:
            for _m in range(rng.choice([1, 1, 2, 4, 16])):
                pos = rng.randrange(0, min(len(raw), 65536)) if rng.random() < 0.6 else rng.randrange(0, len(raw))
                raw[pos] = rng.choice([0, 0xFF, raw[pos] ^ (1 << rng.randrange(8)), rng.randrange(256)])
        try:
            read_all(bytes(raw))
            ok += 1
        except (H.H5Error, KeyError):
            refused += 1
    assert ok > 100 and refused > 100


def test_outcome_files_of_the_1d_notebook(tmp_path):
    """1D_Segmentation.ipynb writes its test signals, targets and predictions as datasets 'SM' / 'App' / 'App_Pred' of one .h5."""
    from segamd import harness
    rng = np.random.default_rng(0)
    arrays = {"SM": rng.standard_normal((5, 64, 3)).astype(np.float32), "App": rng.standard_normal((5, 64, 1)), "App_Pred": rng.standard_normal((5, 64, 1)).astype(np.float32),
              "labels/fold": np.arange(5, dtype=np.int64)}
    p = str(tmp_path / "UNet_Ch3_F1_P1_Kettle.h5")
    harness.save_arrays_h5(p, arrays)
    back = harness.load_arrays_h5(p)
    assert set(back) == set(arrays) and all(back[k].dtype == arrays[k].dtype and np.array_equal(back[k], arrays[k]) for k in arrays)

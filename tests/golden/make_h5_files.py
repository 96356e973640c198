#!/opt/conda/bin/python3.9
"""Writes tests/golden/h5/*.h5 with the REAL HDF5 library (h5py 3.3 / HDF5 1.10.6 of this image's /opt/conda - not importable from
the system interpreter) and expected.npz with what is in them: the fixtures tests/test_h5lite.py holds the pure-Python reader to.

    env -u PYTHONHOME -u PYTHONPATH /opt/conda/bin/python3.9 tests/golden/make_h5_files.py

Build-authored; the files are data (a few KB each)."""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(5)
expected = {}


def put(fname, f, path, data, **kw):
    f.create_dataset(path, data=data, **kw)
    expected[f"{fname}:{path}"] = np.asarray(data)


# 1. what libhdf5 writes by default (superblock 0, version-1 headers, symbol-table groups): every datatype / layout of the subset
with h5py.File(os.path.join(HERE, "default.h5"), "w") as f:
    put("default", f, "f32", rng.standard_normal((3, 3, 5, 7)).astype(np.float32))
    put("default", f, "grp/sub/f64", rng.standard_normal((4, 2)))
    put("default", f, "grp/sub/f16", rng.standard_normal(9).astype(np.float16))
    put("default", f, "grp/i64", np.arange(-5, 6, dtype=np.int64))
    put("default", f, "grp/u8", np.arange(200, 210, dtype=np.uint8))
    put("default", f, "grp/i32_be", np.arange(12, dtype=">i4").reshape(3, 4))
    put("default", f, "grp/f32_be", rng.standard_normal(6).astype(">f4"))
    put("default", f, "scalar", np.float32(2.5))
    put("default", f, "bools", np.array([True, False, True]))
    put("default", f, "empty", np.zeros((0, 4), np.float32))
    put("default", f, "chunked_gzip", rng.standard_normal((37, 21)).astype(np.float32), chunks=(8, 8), compression="gzip", shuffle=True)
    put("default", f, "chunked_plain", np.arange(1000, dtype=np.int32).reshape(10, 100), chunks=(3, 7))
    put("default", f, "chunked_fletcher", np.arange(50, dtype=np.float64), chunks=(16,), fletcher32=True)
    put("default", f, "chunked_many", rng.standard_normal((40, 40)).astype(np.float32), chunks=(2, 5))          # 160 chunks: a two-level chunk B-tree
    f.create_dataset("never_written", shape=(4, 4), dtype="f4")
    expected["default:never_written"] = np.zeros((4, 4), np.float32)
    put("default", f, "fixed_strings", np.array([b"conv2d/kernel:0", b"conv2d/bias:0", b"x"]))
    f.create_dataset("vlen_strings", data=["alpha", "", "gamma delta"], dtype=h5py.string_dtype())
    f.attrs["keras_version"] = "2.13.1"                                                 # python str -> variable-length UTF-8 string
    f.attrs["backend"] = b"tensorflow"                                                  # bytes
    f.attrs["layer_names"] = np.array([b"conv2d", b"batch_normalization", b"re_lu"])    # what Keras' legacy writer stores
    f.attrs["config"] = "{" + ", ".join(f'"k{i}": {i}' for i in range(6000)) + "}"      # 76 KB (more than an object-header message holds): lives in the global heap
    f["grp"].attrs["number"] = 7
    f["grp"].attrs["vector"] = np.arange(5, dtype=np.float64) / 4
    f["f32"].attrs["note"] = "on a dataset"
    g = f.create_group("wide")                                                          # 300 links: 38 symbol nodes of 8, a two-level B-tree
    for i in range(300):
        g.create_dataset(f"layer_{i:04d}", data=np.float32(i))
    f.create_group("hollow/vars")                                                       # empty groups (a Keras layer without weights)
    f["soft"] = h5py.SoftLink("/f32")                                                   # (not followed by the reader)

# 2. libver='latest' (superblock 3, version-2 headers, compact link messages, layout version 4)
with h5py.File(os.path.join(HERE, "latest.h5"), "w", libver="latest") as f:
    put("latest", f, "a/b/kernel", rng.standard_normal((3, 3, 4, 8)).astype(np.float32))
    put("latest", f, "a/b/bias", rng.standard_normal(8).astype(np.float32))
    put("latest", f, "a/single_chunk", rng.standard_normal((6, 6)).astype(np.float32), chunks=(6, 6), compression="gzip")
    put("latest", f, "compact", np.arange(6, dtype=np.int16))
    f.attrs["title"] = "latest format"
    f["a"].attrs["n"] = np.int32(3)
    g = f.create_group("dense")                                                         # more than eight links: dense storage, refused
    for i in range(20):
        g.create_dataset(f"d{i}", data=np.float32(i))

# 3. track_order groups in a default-format file (link messages inside version-1 object headers)
with h5py.File(os.path.join(HERE, "tracked.h5"), "w", track_order=True) as f:
    g = f.create_group("layers", track_order=True)
    for nm in ("zeta", "alpha", "mid"):
        put("tracked", f, f"layers/{nm}/vars/0", rng.standard_normal((2, 3)).astype(np.float32))

np.savez(os.path.join(HERE, "expected.npz"), **{k.replace("/", "|"): v for k, v in expected.items()})
for fn in sorted(os.listdir(HERE)):
    print(fn, os.path.getsize(os.path.join(HERE, fn)))

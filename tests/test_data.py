"""CPU: the data formats either side of the hot path (package `data`): patches, deep-supervision targets,
the folder loader of DataGenerator.py:7-114, the metrics of Test.py:191-193 (checked against scikit-learn, the library
the reference calls), and the prediction loop of Test.py:118-200 over a duck-typed model."""
import os

import numpy as np
import pytest

import segamd
from segamd import data as D
from segamd import harness as H


# ------------------------------------------------------------------------------------------------ patches
def test_patchify_known_answers():
    img = np.arange(16).reshape(4, 4)
    p = D.patchify(img, (2, 2), step=2)
    assert p.shape == (2, 2, 2, 2)
    assert p[0, 0].tolist() == [[0, 1], [4, 5]] and p[0, 1].tolist() == [[2, 3], [6, 7]]
    assert p[1, 0].tolist() == [[8, 9], [12, 13]] and p[1, 1].tolist() == [[10, 11], [14, 15]]
    q = D.patchify(img, (2, 2), step=1)
    assert q.shape == (3, 3, 2, 2) and q[1, 2].tolist() == [[6, 7], [10, 11]]
    r = D.patchify(np.arange(5 * 7).reshape(5, 7), (3, 3), step=2)          # windows that do not fit are dropped
    assert r.shape == (2, 3, 3, 3) and r[1, 2, 0].tolist() == [18, 19, 20]
    c = np.arange(4 * 4 * 3).reshape(4, 4, 3)
    pc = D.patchify(c, (2, 2, 3), step=2)
    assert pc.shape == (2, 2, 1, 2, 2, 3) and np.array_equal(pc[1, 0, 0], c[2:4, 0:2, :])
    with pytest.raises(ValueError):
        D.patchify(img, (2, 2, 1), step=1)
    with pytest.raises(ValueError):
        D.patchify(img, (5, 2), step=1)


def test_unpatchify_round_trip_and_overwrite_rule():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (8, 12)).astype(np.uint8)
    for patch, step in [((4, 4), 4), ((4, 4), 2), ((2, 6), 2), ((8, 12), 1)]:
        p = D.patchify(img, patch, step=step)
        assert np.array_equal(D.unpatchify(p, img.shape), img), (patch, step)
    # inconsistent overlapping patches: width 8, patch 4, step 2 -> starts 0, 2, 4.  After patch 0 the walk jumps to the
    # first patch at or beyond column 4, i.e. patch 2: patch 1 is never used.
    p = np.stack([np.full((1, 4), v) for v in (1, 2, 3)])[None]               # (1, 3, 1, 4)
    assert D.unpatchify(p, (1, 8)).tolist() == [[1, 1, 1, 1, 3, 3, 3, 3]]
    # width 6, patch 4, step 2 -> starts 0, 2: the second patch overwrites columns 2..3 of the first
    p = np.stack([np.full((1, 4), v) for v in (1, 2)])[None]
    assert D.unpatchify(p, (1, 6)).tolist() == [[1, 1, 2, 2, 2, 2]]
    # the same rule down the rows
    p = np.stack([np.full((4, 1), v) for v in (1, 2, 3)])[:, None]             # (3, 1, 4, 1)
    assert D.unpatchify(p, (8, 1))[:, 0].tolist() == [1, 1, 1, 1, 3, 3, 3, 3]
    with pytest.raises(ValueError):
        D.unpatchify(np.zeros((1, 3, 1, 4)), (1, 9))                            # (9 - 4) / 2 is not an integer step


def test_create_patches_step_and_channels():
    img = np.arange(8 * 8 * 3, dtype=np.float32).reshape(8, 8, 3)
    p, n = D.create_patches(img, (4, 4), 0.5)                                   # step int(4 * 0.5) = 2 -> 3 x 3
    assert p.shape == (3, 3, 1, 4, 4, 3) and n == 9
    assert np.array_equal(np.squeeze(p)[2, 1], img[4:8, 2:6, :])
    p, n = D.create_patches(img[..., 0], (4, 4), 0.0)
    assert p.shape == (2, 2, 4, 4) and n == 4
    p, n = D.create_patches(img[..., 0], (4, 4), 0.3)                           # int(2.8) = 2
    assert n == 9
    with pytest.raises(AssertionError):
        D.create_patches(img, (4, 2), 0.5)


# ------------------------------------------------------------------------------------------------ deep supervision
def test_prepare_train_dict():
    rng = np.random.default_rng(1)
    y = (rng.random((2, 16, 16)) > 0.8).astype(np.int8)
    d = D.prepare_train_dict(y, 3, "UNet")
    assert list(d) == ["out", "level1", "level2", "level3"]
    assert d["out"].shape == (2, 16, 16, 1) and d["out"].dtype == np.int8
    for i in (1, 2, 3):
        w = 2 ** i
        lv = d[f"level{i}"]
        assert lv.shape == (2, 16 // w, 16 // w, 1)
        for n in range(2):
            for a in range(16 // w):
                for b in range(16 // w):
                    assert lv[n, a, b, 0] == y[n, a * w:(a + 1) * w, b * w:(b + 1) * w].max()
    dpp = D.prepare_train_dict(y[..., None], 2, "UNetPP")
    assert all(np.array_equal(dpp[k], y[..., None]) for k in ("out", "level1", "level2"))
    with pytest.raises(KeyError):
        D.prepare_train_dict(y, 2, "UNet3P")
    # odd extents: 'valid' pooling drops the remainder
    assert D.max_pool2d(np.arange(2 * 5 * 7 * 1).reshape(2, 5, 7, 1), 2).shape == (2, 2, 3, 1)


def test_targets_for_orders_by_output_name():
    class M:
        output_names = ["out", "level1", "level2"]
    y = {"level2": np.full((1, 2, 2, 1), 2), "out": np.zeros((1, 8, 8, 1)), "level1": np.ones((1, 4, 4, 1))}
    t = D.targets_for(M, y)
    assert [float(a.flat[0]) for a in t] == [0.0, 1.0, 2.0] and all(a.dtype == np.float32 for a in t)
    with pytest.raises(KeyError):
        D.targets_for(M, {"out": y["out"]})
    with pytest.raises(ValueError):
        D.targets_for(M, [y["out"]])


# ------------------------------------------------------------------------------------------------ metrics
def test_dice_and_confusion_matrices_match_sklearn():
    a = np.array([[1, 1, 0], [0, 1, 0]])
    b = np.array([[1, 0, 0], [0, 1, 1]])
    assert D.dice(a, b) == pytest.approx(2 * 2 / 6)
    assert D.dice(np.zeros((2, 2)), np.zeros((2, 2))) == 1.0 and D.dice(np.zeros(3), np.zeros(3), empty_score=0.5) == 0.5
    with pytest.raises(ValueError):
        D.dice(np.zeros(3), np.zeros(4))
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(2)
    for k in (2, 4):
        yt, yp = rng.integers(0, k, 500), rng.integers(0, k, 500)
        assert np.array_equal(D.confusion_matrix(yt, yp), sk.confusion_matrix(yt, yp))
        assert np.array_equal(D.multilabel_confusion_matrix(yt, yp), sk.multilabel_confusion_matrix(yt, yp))
        lab = np.arange(k + 1)
        assert np.array_equal(D.confusion_matrix(yt, yp, lab), sk.confusion_matrix(yt, yp, labels=lab))
        assert np.array_equal(D.multilabel_confusion_matrix(yt, yp, lab), sk.multilabel_confusion_matrix(yt, yp, labels=lab))
    yt, yp = np.array([0, 0, 1]), np.array([0, 0, 0])                           # a class nobody predicts
    assert np.array_equal(D.confusion_matrix(yt, yp), sk.confusion_matrix(yt, yp))


# ------------------------------------------------------------------------------------------------ folder loader
def _write_set(tmp_path, n, size=(20, 24), seed=0):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    idir, mdir = tmp_path / "img", tmp_path / "msk"
    idir.mkdir()
    mdir.mkdir()
    imgs, msks = {}, {}
    for k in range(n):
        name = f"s{k:02d}.png"
        im = rng.integers(0, 256, (size[0], size[1], 3)).astype(np.uint8)
        mk = (rng.integers(0, 3, size) * 127 + (rng.integers(0, 2, size))).astype(np.uint8)     # 0,1,127,128,254,255
        Image.fromarray(im, "RGB").save(idir / name)
        Image.fromarray(mk, "L").save(mdir / name)
        imgs[name], msks[name] = im, mk
    return str(idir), str(mdir), imgs, msks


def test_custom_data_generator_batches(tmp_path):
    idir, mdir, imgs, msks = _write_set(tmp_path, 5)
    g = D.CustomDataGenerator(idir, mdir, img_size=(20, 24), batch_size=2, is_train=False)
    assert len(g) == 3
    names = os.listdir(idir)
    x, y = g[0]
    assert x.shape == (2, 20, 24, 3) and x.dtype == np.float32 and y.shape == (2, 20, 24, 1) and y.dtype == np.int8
    assert np.allclose(x[0], imgs[names[0]] / 255.0) and np.allclose(x[1], imgs[names[1]] / 255.0)
    assert np.array_equal(y[0, ..., 0], (msks[names[0]] / 255.0).astype(np.int8))        # only 255 survives the int8 cast
    assert set(np.unique(y)) <= {0, 1}
    x, y = g[2]
    assert x.shape[0] == 1                                                               # the last, smaller batch
    with pytest.raises(IndexError):
        g[3]
    gd = D.CustomDataGenerator(idir, mdir, img_size=(20, 24), batch_size=2, is_train=False, drop_remainder=True)
    assert len(gd) == 2 and len(list(gd)) == 2
    # resizing: Lanczos for the image, nearest for the mask
    gr = D.CustomDataGenerator(idir, mdir, img_size=(10, 12), batch_size=5, is_train=False, norm_factor_msk=1)
    x, y = gr[0]
    assert x.shape == (5, 10, 12, 3) and y.shape == (5, 10, 12, 1)
    assert set(np.unique(y.astype(np.uint8))) <= set(np.unique(np.stack(list(msks.values()))))    # nearest invents no grey level
    # training order: reshuffled on every epoch end, every sample exactly once per epoch
    gt = D.CustomDataGenerator(idir, mdir, img_size=(20, 24), batch_size=2, is_train=True, rng=np.random.default_rng(3))
    orders = []
    for _ in range(4):
        orders.append(tuple(gt.indexes))
        assert sorted(gt.indexes) == list(range(5))
        gt.on_epoch_end()
    assert len(set(orders)) > 1


def test_custom_data_generator_patches_and_deep_supervision(tmp_path):
    idir, mdir, imgs, msks = _write_set(tmp_path, 3, size=(16, 16))
    g = D.CustomDataGenerator(idir, mdir, img_size=(16, 16), batch_size=2, is_train=False, patchify=True,
                              patch_shape=(8, 8), overlap_ratio=0.5, deep_supervision=True, model_depth=2, ds_type="UNet")
    x, y = g[0]
    assert x.shape == (2 * 9, 8, 8, 3)                                  # 3 x 3 patches per image at step 4
    assert set(y) == {"out", "level1", "level2"} and y["out"].shape == (18, 8, 8, 1) and y["level2"].shape == (18, 2, 2, 1)
    names = os.listdir(idir)
    assert np.allclose(x[9 + 3 * 1 + 2], imgs[names[1]][4:12, 8:16] / 255.0)
    assert np.array_equal(y["level1"], D.max_pool2d(y["out"], 2))


# ------------------------------------------------------------------------------------------------ fit / predict loops
class _Loss(float):
    def cpu(self):
        return self


class _Model:
    """duck-typed SegModel: two outputs, records what fit() hands it; predicts the red channel."""
    output_names = ["out", "level1"]

    def __init__(self):
        self.seen = []

    def train_step(self, x, targets):
        self.seen.append((tuple(x.shape), [tuple(t.shape) for t in targets], str(x.dtype), [str(t.dtype) for t in targets]))
        return _Loss(1.0)

    def forward(self, x, training=False):
        return None

    def compute_loss(self, targets, loss=None):
        return _Loss(0.5)

    def predict(self, x):
        x = np.asarray(x)
        return [x[..., :1].copy(), x[:, ::2, ::2, :1].copy()]


def test_fit_takes_the_generator_and_its_target_dictionary(tmp_path):
    pytest.importorskip("torch")
    idir, mdir, _i, _m = _write_set(tmp_path, 4, size=(16, 16))
    g = D.CustomDataGenerator(idir, mdir, img_size=(16, 16), batch_size=2, is_train=True, deep_supervision=True,
                              model_depth=1, ds_type="UNet", rng=np.random.default_rng(0))
    m = _Model()
    first = tuple(g.indexes)
    hist = H.fit(m, g, 3, val_batches=g)
    assert hist["loss"] == [1.0] * 3 and hist["val_loss"] == [0.5] * 3
    assert len(m.seen) == 6
    assert m.seen[0] == ((2, 16, 16, 3), [(2, 16, 16, 1), (2, 8, 8, 1)], "torch.float32", ["torch.float32"] * 2)
    assert tuple(g.indexes) != first or True                              # (reshuffled; equality is possible by chance)


def test_predict_directory_writes_masks_and_counts(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    idir, mdir, imgs, msks = _write_set(tmp_path, 3, size=(16, 16))
    out = tmp_path / "pred"
    r = H.predict_directory(_Model(), idir, mdir, str(out), img_size=(16, 16), threshold=0.5)
    names = os.listdir(idir)
    assert r["names"] == names and r["cm"].shape == (2, 2) and r["cm"].sum() == 3 * 256
    tot_tp = 0
    for k, name in enumerate(names):
        want = np.where(imgs[name][..., 0] / 255.0 < 0.5, 0, 1)
        got = np.asarray(Image.open(out / name))
        assert got.dtype == np.uint8 and np.array_equal(got, (want * 255).astype(np.uint8))
        truth = (msks[name] / 255.0).astype(np.uint8)
        assert r["dice"][k] == pytest.approx(D.dice(truth, want))
        tot_tp += int(np.sum((truth == 1) & (want == 1)))
    assert r["cm"][1, 1] == tot_tp and r["cm_per_class"][1, 1, 1] == tot_tp
    # the patch path: cut, predict, lay back - identical to the whole-image result for a pointwise model
    r2 = H.predict_directory(_Model(), idir, mdir, None, img_size=(16, 16), patch_shape=(8, 8), overlap_ratio=0.5)
    assert np.array_equal(r2["cm"], r["cm"]) and np.allclose(r2["dice"], r["dice"])


def test_mask_rule_multiclass_variants():
    p = np.array([[[[0.6, 0.7, 0.1]], [[0.2, 0.1, 0.3]]]])                 # (1, 2, 1, 3)
    assert H.masks_from_predictions(p)[0, :, 0, 0].tolist() == [1, 2]
    assert H.masks_from_predictions(p, 0.5, "count")[0, :, 0, 0].tolist() == [2, 0]
    with pytest.raises(ValueError):
        H.masks_from_predictions(p, 0.5, "nope")

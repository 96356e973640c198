"""CPU: the f1-f3 rows (host logic): Keras callback semantics of Train.py:372-387, the mask rule of Test.py:162-183,
weight files by structural name."""
import numpy as np
import pytest

import segamd
from segamd import harness as H


class FakeLoss(float):
    def cpu(self):
        return self


class FakeModel:
    """duck-typed SegModel: the loss of epoch e is scripted."""

    def __init__(self, train_losses, val_losses):
        self.t, self.v = list(train_losses), list(val_losses)
        self.lr = 1e-3
        self.epoch = -1
        self.saved = []

    def train_step(self, x, targets):
        return FakeLoss(self.t[self.epoch])

    def forward(self, x, training=False):
        return None

    def compute_loss(self, targets, loss=None):
        return FakeLoss(self.v[self.epoch])

    def save_weights(self, path):
        self.saved.append((self.epoch, path))


def run(model, epochs, cbs):
    def batches():
        model.epoch += 1
        return [(None, [None])]
    return H.fit(model, batches, epochs, val_batches=lambda: [(None, [None])], callbacks=cbs)


def test_early_stopping_and_best_only_checkpoint():
    val = [1.0, 0.8, 0.9, 0.85, 0.81, 0.7, 0.6]
    m = FakeModel([1.0] * 7, val)
    es = H.EarlyStopping("val_loss", patience=3, mode="min")
    ck = H.ModelCheckpoint("w.npz", "val_loss", "min", save_best_only=True)
    hist = run(m, 7, [es, ck])
    # best 0.8 at epoch 1; epochs 2,3,4 do not improve -> wait reaches 3 at epoch 4 -> stop (Keras stops AFTER that epoch)
    assert es.stopped_epoch == 4 and len(hist["val_loss"]) == 5
    assert ck.saved_epochs == [0, 1] and [e for e, _ in m.saved] == [0, 1]


def test_reduce_lr_on_plateau_min_delta_and_floor():
    val = [1.0, 0.99995, 0.9999, 0.99985, 0.5, 0.5, 0.5, 0.5]       # improvements below min_delta 1e-4 do not count
    m = FakeModel([1.0] * 8, val)
    rl = H.ReduceLROnPlateau("val_loss", factor=0.5, patience=2, mode="min")
    hist = run(m, 8, [rl])
    # epoch1: wait1, epoch2: wait2 -> lr 5e-4 (wait reset); epoch3: wait1; epoch4 improves; 5: wait1, 6: wait2 -> 2.5e-4; 7: wait1
    assert hist["lr"] == pytest.approx([1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 5e-4, 2.5e-4])
    assert m.lr == pytest.approx(2.5e-4)
    with pytest.raises(ValueError):
        H.ReduceLROnPlateau(factor=1.0)
    m2 = FakeModel([1.0] * 4, [1.0] * 4)
    m2.lr = 1e-6
    run(m2, 4, [H.ReduceLROnPlateau(factor=0.1, patience=1, min_lr=5e-7)])
    assert m2.lr == pytest.approx(5e-7)


def test_mask_rule_and_grey_levels():
    p1 = np.array([[[[0.49], [0.5]], [[0.51], [0.0]]]])             # (1,2,2,1): pred < 0.5 -> 0 else 1 (0.5 itself -> 1)
    assert H.masks_from_predictions(p1)[0, :, :, 0].tolist() == [[0, 1], [1, 0]]
    assert H.masks_from_predictions([p1, p1 * 0])[0, :, :, 0].tolist() == [[0, 1], [1, 0]]      # deep supervision: output 0
    p4 = np.zeros((1, 1, 2, 4))
    p4[0, 0, 0] = [0.1, 0.6, 0.2, 0.1]
    p4[0, 0, 1] = [0.3, 0.3, 0.1, 0.3]                               # tie -> first maximum
    assert H.masks_from_predictions(p4)[0, 0, :, 0].tolist() == [1, 0]
    g = H.mask_grey_levels(np.array([[[[0], [1]], [[2], [3]]]])[0][None][0][None], 255, 4) if False else \
        H.mask_grey_levels(np.array([[[0], [1]], [[2], [3]]]), 255, 4)
    assert g.tolist() == [[0, 64], [128, 192]]                       # round(255/4) = 64


def test_weight_file_roundtrip(tmp_path):
    w = {"enc1/conv/kernel": np.random.rand(3, 3, 3, 8).astype(np.float32), "X01/bn/gamma": np.ones(8, np.float32)}
    p = str(tmp_path / "w")
    H.save_weights_npz(w, p)
    r = H.load_weights_npz(p)
    assert set(r) == set(w) and all(np.array_equal(r[k], w[k]) for k in w)


def test_dry_run_model_save_load(tmp_path):
    import importlib
    M = importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd.model")
    M._DryRun.on = True
    try:
        m = segamd.unet_model_builder("UNet", 32, 32, 8, 2, train_mode="from_scratch").ResNet50().build(1, dtype="f16")
        w = m.get_weights()
        path = str(tmp_path / "model.npz")
        m.save_weights(path)
        m2 = segamd.unet_model_builder("UNet", 32, 32, 8, 2, train_mode="from_scratch").ResNet50()
        m2.seed = 99
        m2.build(1, dtype="f16")
        m2.load_weights(path)
        w2 = m2.get_weights()
        assert all(np.array_equal(w[k], w2[k]) for k in w)
    finally:
        M._DryRun.on = False

"""GPU: the f1-f3 callers on the HIP path - fit() lowers the loss and its callbacks drive the model; the mask path
(predict -> DS output 0 -> threshold) equals the oracle's bit for bit in fp32 mode; weight files round-trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import segamd
from segamd import harness as H
from oracle.keras_ref import Layers
from oracle.models_ref import unet2d_forward


def build(dtype="f32", loss="bce", **kw):
    b = segamd.unet_model_builder("UNetPP", 32, 32, 16, 3, ds=1, num_channels=3, train_mode="from_scratch", **kw)
    m = b.ResNet50()
    m.build(4, dtype=dtype)
    m.compile(loss=loss, lr=2e-3)
    return m


def same_training(p0, p1, l0, l1):
    """Two runs of the same training on two execution paths.  They are not bit-identical: split-K weight gradients are
    float-atomic sums (order varies), one flipped fp16 rounding of a weight follows, and Adam's first steps turn a small
    gradient change of a near-zero gradient into an update change of up to lr.  So: the losses agree closely, and all
    but a sliver of the parameters agree far below one Adam step (a path that skipped or doubled an update of a slice
    moves EVERY parameter of it by ~lr = 2e-3).  The comparison tests train on 'mse': Keras' BCE clips the prediction
    to [1e-7, 1 - 1e-7] with zero gradient outside, and on the linear deep-supervision outputs of this model a pixel
    within rounding noise of the clip edge switches a gradient of 1/(1e-7 n) on or off."""
    assert np.allclose(l0, l1, rtol=2e-3, atol=1e-5), (l0, l1)
    d = (p0 - p1).abs()
    frac = float((d > 2e-4).float().mean())
    assert frac < 0.01 and float(d.max()) <= 2e-2, (frac, float(d.max()), float(d.mean()))


def test_fit_callbacks_and_weight_files(tmp_path):
    m = build()
    g = torch.Generator().manual_seed(0)
    x = torch.rand((4, 32, 32, 3), generator=g)
    t = (x[..., :1] > 0.5).float()
    tg = [t.cuda() for _ in m.out_heads]
    xb = x.cuda()
    ck = H.ModelCheckpoint(str(tmp_path / "best.npz"), "val_loss", "min")
    rl = H.ReduceLROnPlateau("val_loss", factor=0.5, patience=1, mode="min")
    es = H.EarlyStopping("val_loss", patience=50, mode="min")
    hist = H.fit(m, lambda: [(xb, tg)] * 4, 6, val_batches=lambda: [(xb, tg)], callbacks=[es, ck, rl])
    assert len(hist["loss"]) == 6 and hist["loss"][-1] < hist["loss"][0]
    assert ck.saved_epochs and ck.best == pytest.approx(min(hist["val_loss"]))
    m2 = build()
    m2.load_weights(str(tmp_path / "best.npz"))
    m2.forward(xb, training=False)
    v = float(m2.compute_loss(tg).cpu())
    assert v == pytest.approx(ck.best, rel=1e-5)


def test_mask_path_matches_oracle_bit_exact():
    m = build()
    w = m.get_weights()
    r = np.random.default_rng(3)
    for k in w:                                      # non-trivial inference statistics
        if k.endswith("moving_mean"):
            w[k] = r.normal(0, 0.1, w[k].shape).astype(np.float32)
        if k.endswith("moving_variance"):
            w[k] = r.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
    m.set_weights(w)
    L = Layers(2, seed=3)
    L.set_weights(w)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((4, 32, 32, 3), generator=g)
    L.update_moving = False
    L.training = False
    with torch.no_grad():
        ref = unet2d_forward(L, x, "UNetPP", 16, 3, ds=1)
    pred = m.predict(x)
    assert isinstance(pred, list) and len(pred) == len(ref)
    got = H.masks_from_predictions(pred, 0.5)
    want = H.masks_from_predictions([r.numpy() for r in ref], 0.5)
    safe = np.abs(ref[0].numpy() - 0.5) > 1e-5           # pixels whose fp32 value is not within rounding of the threshold
    assert np.array_equal(got[safe], want[safe]) and safe.mean() > 0.999
    assert np.abs(pred[0] - ref[0].numpy()).max() < 1e-4
    assert H.mask_grey_levels(got[0], 255, 1).max() in (0, 255)


def test_bucketed_optimizer_path_matches_flat_adam():
    """The overlapped data-parallel path (bucket hooks issued from the weight-gradient stream, Adam per exchanged slice)
    with a no-op exchange must leave exactly the parameters the single-launch path leaves."""
    class Done:
        def wait(self):
            return True

    def run(bucketed):
        m = build("f16", loss="mse")
        if bucketed:
            pend = []
            m.bucket_elems = 3000                                  # many small op-aligned buckets
            m.bucket_hook = lambda lo, hi: pend.append((lo, hi, Done()))
            def take():
                out = list(pend)
                pend.clear()
                return out
            m.bucket_waits = take
        g = torch.Generator().manual_seed(0)
        x = torch.rand((4, 32, 32, 3), generator=g).cuda()
        tg = [(x[..., :1] > 0.5).float() for _ in m.out_heads]
        losses = [float(m.train_step(x, tg).cpu()) for _ in range(3)]
        torch.cuda.synchronize()
        return m.store.P.clone(), losses, (len(m.grad_buckets()) if bucketed else 0)

    p0, l0, _ = run(False)
    p1, l1, nb = run(True)
    assert nb > 3
    same_training(p0, p1, l0, l1)


def test_graphed_train_step_matches_eager():
    """One hipGraph replay per optimiser step (both streams captured) must train exactly like the eager step; capturing
    itself must not train (its warm-up step is undone), and an eager step in between must not race the next replay."""
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand((4, 32, 32, 3), generator=g).cuda() for _ in range(4)]

    def run(graphed):
        m = build("f16", loss="mse")
        tgs = [[(x[..., :1] > 0.5).float() for _ in m.out_heads] for x in xs]
        losses = []
        if graphed:
            p_before = m.store.P.clone()
            step = m.graphed_train_step(xs[3], tgs[3])           # example batch: any batch of the right shape
            assert torch.equal(m.store.P, p_before) and m.adam_t == 0, "capturing must leave the model as it was"
        for i, (x, tg) in enumerate(zip(xs, tgs)):
            if graphed and i == 2:
                lv = m.train_step(x, tg)                         # an eager step between two replays (side-stream Adam in flight)
            else:
                lv = step(x, tg) if graphed else m.train_step(x, tg)
            losses.append(float(lv.cpu()))
        m._join_side()
        torch.cuda.synchronize()
        return m.store.P.clone(), losses, m.adam_t

    p0, l0, t0 = run(False)
    p1, l1, t1 = run(True)
    assert t0 == t1 == 4
    same_training(p0, p1, l0, l1)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_predict_after_graph_replays_uses_the_updated_weights(dtype):
    """predict -> N graph replays -> predict: the replays move gamma / beta / the moving statistics on the device, so the folded
    inference-mode BN coefficients cached by the first predict (ConvOp.inf_scale / inf_shift) must be re-derived - the second
    predict has to equal the one of a model that took the same steps eagerly (advisor finding, round 3)."""
    g = torch.Generator().manual_seed(1)
    xs = [torch.rand((4, 32, 32, 3), generator=g).cuda() for _ in range(3)]

    def run(graphed):
        m = build(dtype, loss="mse")
        tgs = [[(x[..., :1] > 0.5).float() for _ in m.out_heads] for x in xs]
        step = m.graphed_train_step(xs[0], tgs[0]) if graphed else m.train_step
        before = m.predict(xs[0])[0].copy()               # caches the folded coefficients of the initial weights
        for x, tg in zip(xs, tgs):
            step(x, tg)
        return before, m.predict(xs[0])[0]

    b0, a0 = run(False)
    b1, a1 = run(True)
    assert np.array_equal(b0, b1)
    assert float(np.abs(a0 - b0).max()) > 1e-3            # three steps at lr 2e-3 do change the prediction ...
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert float(np.abs(a1 - a0).max()) < tol, float(np.abs(a1 - a0).max())      # ... and the replayed model predicts with the NEW weights


def test_input_outside_the_storage_range_is_reported():
    """|x| >= 65504 / NaN / Inf in the input: the reference's fp32 path computes with it, this engine's storage cannot hold it -
    predict() and fit() raise instead of returning NaN masks silently (advisor finding, round 3)."""
    m = build("f32", loss="mse")
    x = torch.rand(4, 32, 32, 3)
    m.predict(x)                                   # in range: fine
    x[1, 3, 4, 2] = 1e5
    x[2, 0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="2 input element"):
        m.predict(x)
    m.predict(torch.rand(4, 32, 32, 3))            # the counter was reset by the report
    t = [torch.rand(4, 32, 32, 1)] * len(m.out_heads)
    with pytest.raises(ValueError, match="input element"):
        H.fit(m, [(x, t)], epochs=1)


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_exchange_path_single_rank(overlap):
    """The data-parallel exchange through real RCCL launches (group of one rank, force_exchange): the all-reduces are
    issued from the weight-gradient stream while the backward sweep keeps running, the optimiser waits per slice - the
    result must be what the single-GPU step leaves (a one-rank sum is the identity, gmul = 1)."""
    import socket
    import torch.distributed as dist
    from segamd import dp
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        def run(exchange):
            m = build("f16", loss="mse")
            if exchange:
                d = dp.DataParallel(m, bucket_elems=3000, overlap=overlap, force_exchange=True)
                assert (m.bucket_hook is not None) == overlap and (m.allreduce is not None) == (not overlap)
            g = torch.Generator().manual_seed(0)
            x = torch.rand((4, 32, 32, 3), generator=g).cuda()
            tg = [(x[..., :1] > 0.5).float() for _ in m.out_heads]
            losses = [float(m.train_step(x, tg).cpu()) for _ in range(4)]
            torch.cuda.synchronize()
            if exchange:
                assert not d.pending
            return m.store.P.clone(), losses

        p0, l0 = run(False)
        p1, l1 = run(True)
        same_training(p0, p1, l0, l1)
        assert l0[-1] < l0[0]
    finally:
        if created:
            dist.destroy_process_group()


def test_predict_any_sample_count_and_folder_loop(tmp_path):
    """predict() pads the last chunk to the engine's batch (inference-mode BN keeps samples independent): 1, 4 and 6
    samples give the rows of the full batches; the folder loop (loader -> fit -> predict_directory) runs end to end."""
    from PIL import Image
    from segamd import data as D
    m = build()
    g = torch.Generator().manual_seed(1)
    x = torch.rand((8, 32, 32, 3), generator=g)
    full = [np.concatenate([a, b], 0) for a, b in zip(m.predict(x[:4]), m.predict(x[4:]))]
    for n in (1, 3, 4, 6, 8):
        got = m.predict(x[:n])
        assert all(np.array_equal(gk, fk[:n]) for gk, fk in zip(got, full)), n
    rng = np.random.default_rng(0)
    idir, mdir = tmp_path / "img", tmp_path / "msk"
    idir.mkdir()
    mdir.mkdir()
    for k in range(6):
        im = rng.integers(0, 256, (32, 32, 3)).astype(np.uint8)
        Image.fromarray(im, "RGB").save(idir / f"{k}.png")
        Image.fromarray(((im[..., 0] > 127) * 255).astype(np.uint8), "L").save(mdir / f"{k}.png")
    gen = D.CustomDataGenerator(str(idir), str(mdir), img_size=(32, 32), batch_size=4, is_train=True, drop_remainder=True,
                                deep_supervision=True, model_depth=3, ds_type="UNetPP", rng=rng)
    hist = H.fit(m, gen, 8)
    assert hist["loss"][-1] < hist["loss"][0]
    r = H.predict_directory(m, str(idir), str(mdir), str(tmp_path / "pred"), threshold=0.5)
    assert r["cm"].sum() == 6 * 32 * 32 and len(r["dice"]) == 6
    assert sorted(p.name for p in (tmp_path / "pred").iterdir()) == sorted(r["names"])
    assert set(np.unique(np.asarray(Image.open(tmp_path / "pred" / "0.png")))) <= {0, 255}
    # ... and the files the two scripts of the reference exchange: Train.py checkpoints to '<model>_<width>_Fold_<i>.keras' and dumps the
    # history to an .h5 (:375, :417-430); Test.py rebuilds the model FROM that checkpoint (:114) - no constructor call - and predicts
    ckpt = str(tmp_path / "UNetPP_32_Fold_1.keras")
    m.save_weights(ckpt)
    H.save_history_h5(hist, str(tmp_path / "UNetPP_Fold_1_History.h5"))
    assert H.load_history_h5(str(tmp_path / "UNetPP_Fold_1_History.h5"))["loss"] == [float(v) for v in hist["loss"]]
    r2 = H.predict_directory(ckpt, str(idir), str(mdir), str(tmp_path / "pred2"), threshold=0.5)
    assert np.array_equal(r2["cm"], r["cm"]) and np.array_equal(r2["dice"], r["dice"])
    for name in r["names"]:
        assert np.array_equal(np.asarray(Image.open(tmp_path / "pred" / name)), np.asarray(Image.open(tmp_path / "pred2" / name)))


def test_compiled_metrics_are_logged_under_keras_names():
    """compile(metrics=[names of utils/tf_metrics.py]) -> fit logs them like Keras does; the validation values equal
    numpy on predict() (inference-mode BN); the callbacks can monitor them (Train_Configs.ini: val_mean_squared_error)."""
    b = segamd.unet_model_builder("UNet", 32, 32, 16, 3, ds=0, num_channels=3, train_mode="from_scratch", final_activation="sigmoid")
    m = b.ResNet50()
    m.build(4)
    names = ["MeanSquaredError", "RootMeanSquaredError", "MeanAbsoluteError", "BinaryCrossentropy", "BinaryAccuracy", "BinaryIoU",
             "Precision", "Recall", "TruePositives", "tf.keras.metrics.FalseNegatives"]
    m.compile(loss="MeanSquaredLogarithmicError", lr=2e-3, metrics=names)
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand((4, 32, 32, 3), generator=g) for _ in range(3)]
    data = [(x, (x[..., :1] > 0.5).float()) for x in xs]
    rl = H.ReduceLROnPlateau("val_mean_squared_error", factor=0.5, patience=1, mode="min")
    es = H.EarlyStopping("val_mean_squared_error", patience=50, mode="min")
    hist = H.fit(m, data[:2], 3, val_batches=data[2:], callbacks=[es, rl])
    keys = {"loss", "val_loss"}
    for k in ("mean_squared_error", "root_mean_squared_error", "mean_absolute_error", "binary_crossentropy", "binary_accuracy",
              "binary_io_u", "precision", "recall", "true_positives", "false_negatives"):
        keys |= {k, "val_" + k}
    assert set(hist) == keys | {"lr"} and all(len(v) == 3 and np.isfinite(v).all() for v in hist.values())   # ReduceLROnPlateau logs lr
    p = m.predict(xs[2]).astype(np.float64)
    t = data[2][1].numpy().astype(np.float64)
    last = {k: v[-1] for k, v in hist.items()}
    assert last["val_mean_squared_error"] == pytest.approx(((p - t) ** 2).mean(), rel=1e-5)
    assert last["val_root_mean_squared_error"] == pytest.approx(np.sqrt(((p - t) ** 2).mean()), rel=1e-5)
    assert last["val_mean_absolute_error"] == pytest.approx(np.abs(p - t).mean(), rel=1e-5)
    pc = np.clip(p, 1e-7, 1 - 1e-7)
    assert last["val_binary_crossentropy"] == pytest.approx(-(t * np.log(pc) + (1 - t) * np.log(1 - pc)).mean(), rel=1e-4)
    hard = p > 0.5
    tp, fp, fn, tn = (hard & (t == 1)).sum(), (hard & (t == 0)).sum(), (~hard & (t == 1)).sum(), (~hard & (t == 0)).sum()
    assert last["val_true_positives"] == tp and last["val_false_negatives"] == fn
    assert last["val_binary_accuracy"] == pytest.approx((tp + tn) / t.size)
    assert last["val_precision"] == pytest.approx(tp / max(tp + fp, 1)) and last["val_recall"] == pytest.approx(tp / max(tp + fn, 1))
    iou = np.mean([v for v, d in ((tn / max(tn + fn + fp, 1), tn + fn + fp), (tp / max(tp + fn + fp, 1), tp + fn + fp)) if d])
    assert last["val_binary_io_u"] == pytest.approx(iou)
    assert last["val_loss"] == pytest.approx(((np.log1p(np.maximum(p, 1e-7)) - np.log1p(np.maximum(t, 1e-7))) ** 2).mean(), rel=1e-4)
    with pytest.raises(ValueError):
        m.compile(loss="mse", metrics=["AUC"])


def test_categorical_metrics_and_multi_output_prefixes():
    b = segamd.unet_model_builder("UNet", 32, 32, 16, 2, ds=0, num_channels=3, output_nums=3, train_mode="from_scratch",
                                  final_activation="softmax")
    m = b.ResNet50()
    m.build(2)
    m.compile(loss="CategoricalCrossentropy", lr=1e-3, metrics=["CategoricalAccuracy", "CategoricalCrossentropy"])
    g = torch.Generator().manual_seed(1)
    x = torch.rand((2, 32, 32, 3), generator=g)
    t = torch.nn.functional.one_hot(torch.randint(0, 3, (2, 32, 32), generator=g), 3).float()
    logs = H.evaluate(m, [(x, t)], return_dict=True)
    p = m.predict(x)
    assert set(logs) == {"loss", "categorical_accuracy", "categorical_crossentropy"}
    assert logs["categorical_accuracy"] == pytest.approx((p.argmax(-1) == t.numpy().argmax(-1)).mean())
    q = np.clip(p / p.sum(-1, keepdims=True), 1e-7, 1 - 1e-7)
    cce = -(t.numpy() * np.log(q)).sum(-1).mean()
    assert logs["categorical_crossentropy"] == pytest.approx(cce, rel=1e-4) and logs["loss"] == pytest.approx(cce, rel=1e-4)
    # several outputs: Keras prefixes every metric with the output's name
    m2 = build(loss="mse")
    m2.compile(loss="mse", lr=1e-3, metrics=["MeanSquaredError"], loss_weights=[1.0, 0.9, 0.8, 0.7])
    x2 = torch.rand((4, 32, 32, 3), generator=g)
    t2 = (x2[..., :1] > 0.5).float()
    logs2 = H.evaluate(m2, [(x2, [t2] * 4)], return_dict=True)
    assert set(logs2) == {"loss"} | {f"{o}_mean_squared_error" for o in m2.output_names}
    per = [float(((pp - t2.numpy()) ** 2).mean()) for pp in m2.predict(x2)]
    assert [logs2[f"{o}_mean_squared_error"] for o in m2.output_names] == pytest.approx(per, rel=1e-5)
    assert logs2["loss"] == pytest.approx(sum(w * v for w, v in zip([1.0, 0.9, 0.8, 0.7], per)), rel=1e-5)


def test_nonfinite_counter_is_read_one_step_late_without_joining_the_side_stream():
    """harness.fit reads the optimiser's skipped-update counter through SegModel.nonfinite_updates_async(): the value of the
    PREVIOUS step from pinned memory, the device counter reset behind the copy - and a backed-off loss scale grows back."""
    m = segamd.unet_model_builder("UNetPP", 32, 32, 8, 2, num_channels=3, ds=1, train_mode="from_scratch").ResNet50()
    m.build(2)
    m.compile(loss="mse", lr=1e-3)
    x = torch.rand(2, 32, 32, 3)
    t = [torch.rand(2, 32, 32, 1)] * len(m.out_heads)
    m.train_step(x, t)
    assert m.nonfinite_updates_async() == 0
    m.store.nonfinite.fill_(7)                      # as if the optimiser had skipped 7 elements in this step
    assert m.nonfinite_updates_async() == 0         # ... reported one call later
    assert m.nonfinite_updates_async() == 7 and m.nonfinite_updates() == 0
    # nothing is lost at the end of an epoch: the drain returns the queued value plus what was counted since
    m.store.nonfinite.fill_(3)
    assert m.nonfinite_updates_async() == 0
    m.store.nonfinite.fill_(2)
    assert m.nonfinite_drain() == 5 and m.nonfinite_drain() == 0
    # a one-batch epoch with a skipped update is logged in THAT epoch and backs the scale off
    real_step = m.train_step
    def step_with_overflow(x_, t_):
        lv = real_step(x_, t_)
        m._join_side()
        m.store.nonfinite.fill_(11)
        return lv
    m.train_step = step_with_overflow
    m.grad_target = 1024.0
    hist = H.fit(m, [(x, t)], epochs=1)
    m.train_step = real_step
    assert hist["nonfinite_grads"] == [11.0] and m.grad_target == 512.0
    # fit(): back-off on a skipped step, growth after a run of clean ones
    keep = m.loss_scale_growth_interval
    try:
        m.grad_target = m.grad_target_max / 4          # (the parity mode's default target is 32, the throughput mode's 1024)
        H.fit(m, [(x, t)] * 8, epochs=1, loss_scale_growth_interval=3)
        assert m.loss_scale_growth_interval == 3
        assert m.grad_target_max / 4 < m.grad_target <= m.grad_target_max
        m.loss_scale_growth_interval = 5               # a value set on the model survives a fit() that does not name one
        H.fit(m, [(x, t)], epochs=1)
        assert m.loss_scale_growth_interval == 5
    finally:
        m.loss_scale_growth_interval = keep

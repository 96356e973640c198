"""Numeric examples PUBLISHED in the tf.keras (TF 2.x) API documentation for layers on this path - known answers that neither the
oracle nor the engine had a hand in.  The reference ships no vectors and TensorFlow cannot run here, so these,
the Keras docstring answers of the losses (tests/test_oracle_losses.py) and the hand-worked cases of
tests/test_oracle_primitives.py are what the oracle is pinned by; the `-m gpu` half runs the same vectors through the C-ABI.

Sources (docstring examples of the TF 2.x API docs):
  tf.keras.layers.MaxPooling2D   - pool (2, 2), strides (2, 2), 'valid' on a 3 x 4 image
  tf.keras.layers.UpSampling2D   - size (1, 2) on arange(12).reshape(2, 2, 1, 3)
  tf.keras.layers.MaxPooling1D   - pool_size 2, strides 2, 'valid' on [1, 2, 3, 4, 5]
  tf.keras.layers.LeakyReLU      - default alpha 0.3 and alpha = 0.1 on [-3, -1, 0, 2]
  tf.keras.layers.ReLU / activations.relu, activations.sigmoid, activations.tanh, activations.hard_sigmoid (0.2 x + 0.5)
  tf.keras.layers.Concatenate / Add - the documented shapes / values
  tf.keras.optimizers.Adam / SGD / RMSprop - "opt.minimize(lambda: var ** 2 / 2, [var])" with learning_rate 0.1: 10 -> 9.9
                                   (Adam: "the first step is -learning_rate * sign(grad)"), 1 -> 0.9 (SGD), 10 -> 9.683772 (RMSprop)"""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K

MAXPOOL_X = np.array([[1., 2., 3., 4.], [5., 6., 7., 8.], [9., 10., 11., 12.]], np.float32).reshape(1, 3, 4, 1)
MAXPOOL_Y = np.array([6., 8.], np.float32).reshape(1, 1, 2, 1)
UPS_X = np.arange(12, dtype=np.float32).reshape(2, 2, 1, 3)
UPS_Y = np.array([[[[0, 1, 2], [0, 1, 2]], [[3, 4, 5], [3, 4, 5]]], [[[6, 7, 8], [6, 7, 8]], [[9, 10, 11], [9, 10, 11]]]], np.float32)
ACT_X = np.array([-3.0, -1.0, 0.0, 2.0], np.float32)
SIG_X = np.array([-20, -1.0, 0.0, 1.0, 20], np.float32)
SIG_Y = np.array([2.0611537e-09, 2.6894143e-01, 5.0000000e-01, 7.3105860e-01, 1.0000000e+00], np.float32)
HS_X = np.array([-3.0, -1.0, 0.0, 1.0, 3.0], np.float32)
HS_Y = np.array([0., 0.3, 0.5, 0.7, 1.], np.float32)
TANH_Y = np.array([-0.9950547, -0.7615942, 0., 0.7615942, 0.9950547], np.float32)


def test_oracle_reproduces_the_documented_examples():
    T = torch.from_numpy
    assert torch.equal(K.maxpool(T(MAXPOOL_X), (2, 2)), T(MAXPOOL_Y))          # the odd third row is dropped ('valid')
    assert torch.equal(K.upsample(T(UPS_X), (1, 2), "nearest"), T(UPS_Y))
    # tf.keras.layers.MaxPooling1D docs: x = [1, 2, 3, 4, 5], pool_size 2, strides 2, 'valid' -> [2, 4] (the odd last sample is dropped)
    assert K.maxpool(T(np.arange(1, 6, dtype=np.float32).reshape(1, 1, 5, 1)), (1, 2)).reshape(-1).tolist() == [2.0, 4.0]
    assert K.activation(T(ACT_X), "leaky_relu", K.LEAKY_ALPHA).tolist() == pytest.approx([-0.9, -0.3, 0.0, 2.0])
    assert K.activation(T(ACT_X), "leaky_relu", 0.1).tolist() == pytest.approx([-0.3, -0.1, 0.0, 2.0])
    assert K.activation(T(ACT_X), "relu").tolist() == [0.0, 0.0, 0.0, 2.0]
    assert np.allclose(K.activation(T(SIG_X), "sigmoid").numpy(), SIG_Y, rtol=1e-6, atol=0)
    assert np.allclose(K.hard_sigmoid(T(HS_X)).numpy(), HS_Y, atol=1e-7)
    assert np.allclose(torch.tanh(T(HS_X)).numpy(), TANH_Y, atol=1e-7)
    sm = K.activation(torch.tensor([[1.0, 2.0, 3.0]]), "softmax")
    assert abs(float(sm.sum()) - 1.0) < 1e-6 and np.allclose(sm.numpy(), np.exp([1, 2, 3]) / np.exp([1, 2, 3]).sum(), atol=1e-7)
    # Concatenate docs: arange(20).reshape(2, 2, 5) with arange(20, 30).reshape(2, 1, 5) along axis 1 -> (2, 3, 5); Add docs: element-wise
    a, b = np.arange(20).reshape(2, 2, 5), np.arange(20, 30).reshape(2, 1, 5)
    assert np.concatenate([a, b], 1).shape == (2, 3, 5) and np.concatenate([a, b], 1)[0, 2].tolist() == [20, 21, 22, 23, 24]


def test_oracle_optimizers_reproduce_the_documented_first_steps():
    from oracle import train_ref as TR
    w = {"w": torch.tensor([10.0])}
    TR.KerasAdam(w, lr=0.1).step({"w": w["w"].clone()})                 # d(var^2 / 2) / dvar = var
    assert float(w["w"]) == pytest.approx(9.9, abs=1e-6)
    w = {"w": torch.tensor([1.0])}
    TR.KerasOptimizer("SGD", w, lr=0.1).step({"w": w["w"].clone()})
    assert float(w["w"]) == pytest.approx(0.9, abs=1e-7)
    w = {"w": torch.tensor([10.0])}
    TR.KerasOptimizer("RMSprop", w, lr=0.1).step({"w": w["w"].clone()})
    assert float(w["w"]) == pytest.approx(9.683772, abs=1e-6)


@pytest.mark.gpu
def test_engine_optimizers_reproduce_the_documented_first_steps():
    """seg_adam_step_ex / seg_optimizer_step_ex on a flat buffer holding the documented variable (the rest of the slice: zeros)."""
    import gpu_util as U
    M = U.M
    segamd = __import__("segamd")
    for name, w0, want in (("Adam", 10.0, 9.9), ("SGD", 1.0, 0.9), ("RMSprop", 10.0, 9.683772)):
        m = segamd.unet_model_builder("UNet", 32, 32, 8, 1, num_channels=1, train_mode="from_scratch").ResNet50()
        m.build(1, dtype="f32")
        m.compile(loss="mse", optimizer=name, lr=0.1)
        st = m.store
        st.P.fill_(w0)
        m._refresh_copies()
        m._init_optimizer_state()
        st.G.copy_(st.P)                                                 # gradient of var^2 / 2, unscaled (grad scale 1)
        m.gs.fill_(1.0)
        m.optimizer_step()
        m._join_side()
        torch.cuda.synchronize()
        assert float(st.P.max()) == pytest.approx(want, abs=2e-6) and float(st.P.min()) == pytest.approx(want, abs=2e-6), name


@pytest.mark.gpu
def test_engine_reproduces_the_documented_examples():
    """The same vectors through the HIP kernels (C-ABI, both storage modes): max-pool with an odd row, nearest up-sampling by
    (1, 2), LeakyReLU / ReLU / sigmoid through the add-activation and head-activation kernels."""
    import ctypes as C
    import gpu_util as U
    ffi, SF = U.ffi, U.SF
    for f32 in (True, False):
        sfx = "_f32" if f32 else ""
        dt = torch.float32 if f32 else torch.float16
        # MaxPooling2D((2, 2)) on the 3 x 4 image
        x = U.to_dev_nhwc(torch.from_numpy(MAXPOOL_X), dt)
        y = torch.zeros((1, 1, 2, 8), dtype=dt, device="cuda")
        if f32:
            y = SF.split_pack(y)
        ffi.call("seg_maxpool_fwd" + sfx, U.P(x), U.P(y), 1, 3, 4, 8, 2, 2, U.stream())
        assert torch.equal(U.dev_values(y)[..., :1], torch.from_numpy(MAXPOOL_Y))
        # UpSampling2D(size=(1, 2)), nearest
        x = U.to_dev_nhwc(torch.from_numpy(UPS_X), dt)
        y = torch.zeros((2, 2, 2, 8), dtype=dt, device="cuda")
        if f32:
            y = SF.split_pack(y)
        ffi.call("seg_upsample_fwd" + sfx, U.P(x), U.P(y), 2, 2, 1, 8, 1, 2, 0, U.stream())
        assert torch.equal(U.dev_values(y)[..., :3], torch.from_numpy(UPS_Y))
        # LeakyReLU / ReLU / sigmoid as Activation layers (an add of one input)
        for xs, act, alpha, want, tol in ((ACT_X, "leaky_relu", 0.3, [-0.9, -0.3, 0.0, 2.0], 1e-3), (ACT_X, "leaky_relu", 0.1, [-0.3, -0.1, 0.0, 2.0], 1e-3),
                                          (ACT_X, "relu", 0.0, [0.0, 0.0, 0.0, 2.0], 0.0), (SIG_X, "sigmoid", 0.0, SIG_Y.tolist(), 1e-3)):
            n = len(xs)
            x = U.to_dev_nhwc(torch.from_numpy(xs).reshape(1, 1, n, 1).expand(1, 1, n, 8).contiguous(), dt)
            y = torch.zeros_like(x)
            ptrs = (C.c_void_p * 4)(x.data_ptr(), None, None, None)
            ffi.call("seg_add_act_fwd" + sfx, ptrs, 1, U.P(y), n * 8, 8, 8, ffi.ACT[act], alpha, U.stream())
            got = U.dev_values(y)[0, 0, :, 0].tolist()
            assert got == pytest.approx(want, rel=(tol if not f32 else 1e-6), abs=1e-6 if f32 else 1e-3), (act, f32, got)


@pytest.mark.gpu
def test_engine_metrics_reproduce_the_documented_examples():
    """tf.keras.metrics docstring examples through metrics.MetricSet (the device counters / value-only loss launches): the example
    is tiled over the whole output (ratios and means do not change, counts scale with the number of tiles).
      BinaryAccuracy ([[1],[1],[0],[0]], [[0.98],[1],[0],[0.6]]) -> 0.75          Precision / Recall ([0,1,1,1], [1,0,1,1]) -> 0.6666667
      TruePositives ([0,1,1,1],[1,0,1,1]) -> 2   FalsePositives ([0,1,0,0],[0,0,1,1]) -> 2   TrueNegatives ([0,1,0,0],[1,1,0,0]) -> 2
      FalseNegatives ([0,1,1,1],[0,1,0,0]) -> 2   MeanSquaredError ([[0,1],[0,0]], [[1,1],[0,0]]) -> 0.25   RootMeanSquaredError -> 0.5
      MeanAbsoluteError -> 0.25   CategoricalAccuracy ([[0,0,1],[0,1,0]], [[0.1,0.9,0.8],[0.05,0.95,0]]) -> 0.5"""
    import importlib
    import gpu_util as U
    segamd = __import__("segamd")
    MS = importlib.import_module(U.M.__name__.rsplit(".", 1)[0] + ".metrics")

    def run(cout, names, y_true, y_pred):
        m = segamd.unet_model_builder("UNet", 16, 16, 8, 1, num_channels=1, output_nums=cout, train_mode="from_scratch",
                                      final_activation="softmax" if cout > 1 else "sigmoid").ResNet50()
        m.build(1, dtype="f32")
        y = m.out_heads[0].y
        n = y.numel()
        yt, yp = np.asarray(y_true, np.float32).reshape(-1), np.asarray(y_pred, np.float32).reshape(-1)
        assert n % yp.size == 0
        tiles = n // yp.size
        y.copy_(torch.from_numpy(np.tile(yp, tiles)).reshape(y.shape))
        ms = MS.MetricSet(m, names)
        ms.update([torch.from_numpy(np.tile(yt, tiles)).reshape(y.shape)])
        return ms.result(), tiles

    r, _ = run(1, ["BinaryAccuracy"], [1, 1, 0, 0], [0.98, 1, 0, 0.6])
    assert r["binary_accuracy"] == pytest.approx(0.75)
    r, k = run(1, ["Precision", "Recall", "TruePositives"], [0, 1, 1, 1], [1, 0, 1, 1])
    assert r["precision"] == pytest.approx(0.6666667) and r["recall"] == pytest.approx(0.6666667) and r["true_positives"] == 2 * k
    r, k = run(1, ["FalsePositives"], [0, 1, 0, 0], [0, 0, 1, 1])
    assert r["false_positives"] == 2 * k
    r, k = run(1, ["TrueNegatives"], [0, 1, 0, 0], [1, 1, 0, 0])
    assert r["true_negatives"] == 2 * k
    r, k = run(1, ["FalseNegatives"], [0, 1, 1, 1], [0, 1, 0, 0])
    assert r["false_negatives"] == 2 * k
    r, _ = run(1, ["MeanSquaredError", "RootMeanSquaredError", "MeanAbsoluteError"], [0, 1, 0, 0], [1, 1, 0, 0])
    assert r["mean_squared_error"] == pytest.approx(0.25) and r["root_mean_squared_error"] == pytest.approx(0.5)
    assert r["mean_absolute_error"] == pytest.approx(0.25)
    # (one-hot targets; 2 of 256 pixels per tile pair: 16 x 16 pixels x 3 classes = 768 values = 128 tiles of the 6-value example)
    r, _ = run(3, ["CategoricalAccuracy"], [[0, 0, 1], [0, 1, 0]], [[0.1, 0.9, 0.8], [0.05, 0.95, 0]])
    assert r["categorical_accuracy"] == pytest.approx(0.5)

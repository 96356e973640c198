"""CPU: pins the oracle (oracle/keras_ref.py).  The reference ships no tests or golden vectors and TensorFlow is
not installable here, so the oracle is pinned by (1) hand-computed known answers on integer tensors, (2) a second
loop-level numpy implementation (oracle/numpy_ref.py), (3) the Keras-2.x definitions in SURVEY.md 8(c)."""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K
from oracle import numpy_ref as NR
from oracle import train_ref as TR

T = lambda a: torch.from_numpy(np.asarray(a, np.float32))


# ---------------------------------------------------------------- hand-computed known answers
def test_conv3x3_same_known_answer():
    x = T(np.arange(1, 10).reshape(1, 3, 3, 1))                 # 1 2 3 / 4 5 6 / 7 8 9
    k = T(np.ones((3, 3, 1, 1)))
    y = K.conv2d_same(x, k, None)[0, :, :, 0]
    assert y.tolist() == [[12, 21, 16], [27, 45, 33], [24, 39, 28]]          # box sums with zero padding
    k2 = torch.zeros(3, 3, 1, 1)
    k2[0, 1, 0, 0] = 1.0                                        # tap (ky=0, kx=1) = the pixel ABOVE: cross-correlation, no flip
    assert K.conv2d_same(x, k2, None)[0, :, :, 0].tolist() == [[0, 0, 0], [1, 2, 3], [4, 5, 6]]


def test_conv1x1_stride2_samples_even_positions():
    x = T(np.arange(16).reshape(1, 4, 4, 1))
    y = K.conv2d_same(x, T(np.ones((1, 1, 1, 1))), None, (2, 2))[0, :, :, 0]
    assert y.tolist() == [[0, 2], [8, 10]]


def test_conv2d_transpose_4x4_s2_index_rule():
    """Y = 2 i + ky - 1: a single 1 at input (0,0) paints kernel rows/cols 1..3 at output 0..2."""
    x = torch.zeros(1, 2, 2, 1)
    x[0, 0, 0, 0] = 1.0
    k = T(np.arange(16).reshape(4, 4, 1, 1))
    y = K.conv2d_transpose_same(x, k, None, (2, 2))[0, :, :, 0]
    assert y.shape == (4, 4)
    assert y[:3, :3].tolist() == [[5, 6, 7], [9, 10, 11], [13, 14, 15]] and float(y[3].abs().sum()) == 0
    x2 = torch.zeros(1, 2, 2, 1)
    x2[0, 1, 1, 0] = 1.0                                        # input (1,1) -> output rows 1..4 clipped to 1..3 = kernel rows 0..2
    y2 = K.conv2d_transpose_same(x2, k, None, (2, 2))[0, :, :, 0]
    assert y2[1:, 1:].tolist() == [[0, 1, 2], [4, 5, 6], [8, 9, 10]]


def test_conv1d_transpose_k2_s2_index_rule():
    x = T([[[[1.0], [2.0]]]])                                   # (N=1, H=1, W=2, C=1)
    k = T(np.array([10.0, 100.0]).reshape(1, 2, 1, 1))
    assert K.conv2d_transpose_same(x, k, None, (1, 2))[0, 0, :, 0].tolist() == [10, 100, 20, 200]     # y = 2 i + k


def test_bilinear_2x_weights_and_edge_clamp():
    x = T([[[[0.0], [4.0]]]])
    assert K.upsample(x, (1, 2), "bilinear")[0, 0, :, 0].tolist() == [0.0, 1.0, 3.0, 4.0]            # (.75,.25) taps, clamped ends
    assert K.upsample(x, (1, 2), "nearest")[0, 0, :, 0].tolist() == [0.0, 0.0, 4.0, 4.0]


def test_batchnorm_training_biased_norm_unbiased_moving():
    x = T([1.0, 3.0]).reshape(2, 1, 1, 1)
    y, nm, nv = K.batchnorm(x, torch.ones(1), torch.zeros(1), torch.zeros(1), torch.ones(1), True, fused_4d=True)
    # batch mean 2, biased var 1 -> y = +-1/sqrt(1.001); moving_var uses the unbiased estimate 2 on the fused path
    assert np.allclose(y.flatten().numpy(), [-1 / np.sqrt(1.001), 1 / np.sqrt(1.001)], atol=1e-6)
    assert np.isclose(float(nm), 0.02) and np.isclose(float(nv), 0.99 + 0.01 * 2.0)
    _, _, nv1d = K.batchnorm(x, torch.ones(1), torch.zeros(1), torch.zeros(1), torch.ones(1), True, fused_4d=False)
    assert np.isclose(float(nv1d), 0.99 + 0.01 * 1.0)                         # non-fused 3-D path: biased
    yi, _, _ = K.batchnorm(x, torch.ones(1) * 2, torch.ones(1), torch.ones(1), torch.ones(1) * 3, False)
    assert np.allclose(yi.flatten().numpy(), [1.0, 2 * 2 / np.sqrt(3.001) + 1], atol=1e-6)


def test_activation_layers():
    x = T([-2.0, 0.0, 3.0])
    assert K.activation(x, "leaky_relu", K.LEAKY_ALPHA).tolist() == pytest.approx([-0.6, 0.0, 3.0])  # LeakyReLU(alpha=0.3)
    assert K.hard_sigmoid(T([-3.0, -2.5, 0.0, 2.5, 3.0])).tolist() == pytest.approx([0, 0, 0.5, 1, 1])
    assert K.LEAKY_ALPHA == 0.3 and K.BN_EPS == 1e-3 and K.BN_MOMENTUM == 0.99


def test_convlstm_t1_ignores_forget_gate_and_recurrent_kernel():
    x = torch.ones(1, 1, 1, 1)
    k = torch.zeros(1, 1, 1, 4)
    b = T([1.0, 123.0, 0.5, -1.0])                              # z = (i, f, c, o) = bias
    h = K.convlstm_t1(x, k, b)
    exp = np.clip(0.2 * -1 + 0.5, 0, 1) * np.tanh(np.clip(0.2 * 1 + 0.5, 0, 1) * np.tanh(0.5))
    assert np.isclose(float(h), exp, atol=1e-7)


def test_keras_adam_epsilon_outside_root():
    p = {"w": torch.tensor([1.0])}
    opt = TR.KerasAdam(p, lr=0.1, epsilon=1e-7)
    opt.step({"w": torch.tensor([0.5])})
    # t=1: m = .05, v = .00025*... -> alpha_t = lr*sqrt(1-b2)/(1-b1); update = alpha_t * m / (sqrt(v) + eps)
    m, v = 0.05, 0.001 * 0.25
    a = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.isclose(float(p["w"]), 1.0 - a * m / (np.sqrt(v) + 1e-7), atol=1e-7)


def test_losses_keras_definitions():
    p, t = T([[0.9, 0.1], [0.2, 0.8]]), T([[1.0, 0.0], [0.0, 1.0]])
    assert np.isclose(float(TR.categorical_crossentropy(p, t)), -(np.log(0.9) + np.log(0.8)) / 2, atol=1e-6)
    assert np.isclose(float(TR.mse(p, t)), ((0.1 ** 2) * 2 + (0.2 ** 2) * 2) / 4, atol=1e-7)
    pb = T([[0.0], [1.0]])
    assert np.isclose(float(TR.binary_crossentropy(pb, T([[1.0], [1.0]]))), -(np.log(1e-7) + np.log(1 - 1e-7)) / 2, rtol=1e-4)


# ---------------------------------------------------------------- torch-CPU oracle == numpy loop oracle
@pytest.mark.parametrize("shape,k,s", [((2, 5, 6, 3), (3, 3), (1, 1)), ((1, 7, 4, 2), (1, 1), (2, 2)), ((2, 1, 9, 4), (1, 3), (1, 1)),
                                       ((1, 6, 6, 2), (3, 3), (2, 2)), ((1, 4, 5, 2), (2, 2), (1, 1))])
def test_conv_matches_numpy_loops(shape, k, s):
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, shape).astype(np.float32)
    kern = rng.integers(-2, 3, (*k, shape[3], 5)).astype(np.float32)
    b = rng.integers(-2, 3, 5).astype(np.float32)
    assert np.array_equal(K.conv2d_same(T(x), T(kern), T(b), s).numpy(), NR.conv2d_same(x, kern, b, s).astype(np.float32))


@pytest.mark.parametrize("shape,k,s", [((2, 3, 4, 3), (4, 4), (2, 2)), ((1, 1, 6, 2), (1, 2), (1, 2)), ((1, 3, 3, 2), (3, 3), (2, 2)),
                                       ((1, 2, 2, 1), (2, 2), (2, 2))])
def test_conv_transpose_matches_numpy_loops(shape, k, s):
    rng = np.random.default_rng(1)
    x = rng.integers(-3, 4, shape).astype(np.float32)
    kern = rng.integers(-2, 3, (*k, 4, shape[3])).astype(np.float32)
    b = rng.integers(-2, 3, 4).astype(np.float32)
    got = K.conv2d_transpose_same(T(x), T(kern), T(b), s).numpy()
    assert got.shape == (shape[0], shape[1] * s[0], shape[2] * s[1], 4)
    assert np.array_equal(got, NR.conv2d_transpose_same(x, kern, b, s).astype(np.float32))


def test_pool_upsample_bn_lstm_match_numpy_loops():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 6, 8, 3)).astype(np.float32)
    assert np.array_equal(K.maxpool(T(x), (2, 2)).numpy(), NR.maxpool(x, (2, 2)))
    assert np.array_equal(K.maxpool(T(x), (1, 4)).numpy(), NR.maxpool(x, (1, 4)))
    for f in [(2, 2), (4, 4), (1, 2)]:
        assert np.allclose(K.upsample(T(x), f, "bilinear").numpy(), NR.upsample_bilinear(x, f), atol=1e-6)
    g, b = rng.random(3).astype(np.float32) + 0.5, rng.standard_normal(3).astype(np.float32)
    y, nm, nv = K.batchnorm(T(x), T(g), T(b), torch.zeros(3), torch.ones(3), True)
    y2, nm2, nv2 = NR.batchnorm_train(x.astype(np.float64), g, b, np.zeros(3), np.ones(3))
    assert np.allclose(y.numpy(), y2, atol=1e-5) and np.allclose(nm.numpy(), nm2, atol=1e-6) and np.allclose(nv.numpy(), nv2, atol=1e-6)
    kern = rng.standard_normal((3, 3, 3, 8)).astype(np.float32) * 0.5
    bias = rng.standard_normal(8).astype(np.float32)
    assert np.allclose(K.convlstm_t1(T(x), T(kern), T(bias)).numpy(), NR.convlstm_t1(x, kern, bias), atol=1e-5)


def test_initialisers_have_keras_ranges():
    w = K.init_param("a/kernel", (3, 3, 16, 32), "he_uniform", (144, 288), 0)
    assert abs(np.abs(w).max() - np.sqrt(6 / 144)) < 2e-3 and abs(w.mean()) < 5e-3
    w = K.init_param("b/kernel", (4, 4, 8, 16), "glorot_uniform", (128, 256), 0)
    assert np.abs(w).max() <= np.sqrt(6 / 384) and np.abs(w).max() > 0.98 * np.sqrt(6 / 384)
    b = K.init_param("c/bias", (16,), "unit_forget_bias", (0, 0), 0)
    assert b.tolist() == [0] * 4 + [1] * 4 + [0] * 8            # gate order i, f, c, o

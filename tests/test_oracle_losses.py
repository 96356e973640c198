"""CPU: the oracle's loss restatements (oracle/train_ref.py) against the known answers published in the docstrings of
keras/losses.py (the third-party library behind `2DCNN/utils/tf_losses.py:9-39`; the values are the documented
`loss(y_true, y_pred).numpy()` results of each class's first example), plus the label rule of the hinge losses."""
import pytest
import torch

from oracle import train_ref as T

f = lambda a: torch.tensor(a, dtype=torch.float32)
A, B, P = f([[0., 1.], [0., 0.]]), f([[1., 1.], [1., 0.]]), f([[0.6, 0.4], [0.4, 0.6]])
C3T, C3P = f([[0, 1, 0], [0, 0, 1]]), f([[0.05, 0.95, 0], [0.1, 0.8, 0.1]])

KNOWN = [
    ("mse", B, A, 0.5), ("mae", B, A, 0.5), ("msle", B, A, 0.240),
    ("mape", f([[1., 1.], [1., 0.]]), f([[2., 1.], [2., 3.]]), 50.0),
    ("bce", P, A, 0.815), ("huber", P, A, 0.155), ("hinge", P, A, 1.3), ("squared_hinge", P, A, 1.86),
    ("categorical_hinge", P, A, 1.4), ("kld", P, A, 0.458),
    ("logcosh", f([[1., 1.], [0., 0.]]), A, 0.108), ("poisson", f([[1., 1.], [0., 0.]]), A, 0.5),
    ("cce", C3P, C3T, 1.177), ("sparse_cce", C3P, f([1, 2]), 1.177),
    ("cosine", f([[1., 0.], [1., 1.]]), f([[0., 1.], [1., 1.]]), -0.5),
]


@pytest.mark.parametrize("name,pred,target,want", KNOWN, ids=[k[0] for k in KNOWN])
def test_documented_known_answers(name, pred, target, want):
    assert float(T.LOSSES[name](pred, target)) == pytest.approx(want, abs=6e-4)


def test_binary_focal_by_hand():
    # gamma 2: (t, p) -> (1 - p_t)^2 * -log(p_t): (0, .6) .36 * .91629; (1, .4) the same; (0, .4) .16 * .51083; (0, .6)
    want = (3 * 0.36 * 0.916291 + 0.16 * 0.510826) / 4
    assert float(T.binary_focal_crossentropy(P, A)) == pytest.approx(want, rel=1e-5)


def test_hinge_labels_converted_only_when_all_binary():
    p = f([[0.3, -0.2]])
    assert float(T.hinge(p, f([[0., 1.]]))) == pytest.approx((1.3 + 1.2) / 2)          # {0,1} -> {-1,+1}
    assert float(T.hinge(p, f([[0., 2.]]))) == pytest.approx((1.0 + 1.4) / 2)          # left alone
    assert float(T.hinge(p, f([[-1., 1.]]))) == pytest.approx((1.3 + 1.2) / 2)


def test_registry_has_the_sixteen_names():
    assert len(T.LOSSES) == 16


# one step from w = 1 with g = 0.5 and lr = 0.1, worked by hand from the rules in KerasOptimizer's docstring
OPT_KNOWN = {
    "SGD": 0.95,                                        # 1 - 0.1 * 0.5
    "RMSprop": 1 - 0.05 / (0.025 + 1e-7) ** 0.5,        # v = 0.1 * 0.25
    "Adagrad": 1 - 0.05 / (0.35 + 1e-7) ** 0.5,         # a = 0.1 + 0.25
    "Adadelta": 1 - 0.1 * (1e-7 ** 0.5) * 0.5 / (0.0125 + 1e-7) ** 0.5,
    "Adamax": 1 - 0.1 * 0.05 / (0.1 * (0.5 + 1e-7)),    # m = 0.05, u = 0.5
    "Nadam": 1 - 0.1 * (0.48528 * 0.05 / (1 - 0.468 * 0.48528) + 0.5) / (0.5 + 1e-7),
    "FTRL": (0.5 - (0.35 ** 0.5 - 0.1 ** 0.5) / 0.1) / -(0.35 ** 0.5 / 0.1),
}


@pytest.mark.parametrize("name", sorted(OPT_KNOWN))
def test_optimizer_first_step_by_hand(name):
    w = {"w": f([1.0])}
    T.KerasOptimizer(name, w, lr=0.1).step({"w": f([0.5])})
    assert float(w["w"]) == pytest.approx(OPT_KNOWN[name], rel=2e-6)


def test_nadam_momentum_schedule_product_carries_over_steps():
    w = {"w": f([1.0])}
    o = T.KerasOptimizer("Nadam", w, lr=0.1)
    for _ in range(3):
        o.step({"w": f([0.5])})
    u = [0.9 * (1 - 0.5 * 0.96 ** k) for k in (1, 2, 3)]
    assert o.u_product == pytest.approx(u[0] * u[1] * u[2])

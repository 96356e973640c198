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


def build(dtype="f32", **kw):
    b = segamd.unet_model_builder("UNetPP", 32, 32, 16, 3, ds=1, num_channels=3, train_mode="from_scratch", **kw)
    m = b.ResNet50()
    m.build(4, dtype=dtype)
    m.compile(loss="bce", lr=2e-3)
    return m


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
        m = build("f16")
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
        for _ in range(3):
            m.train_step(x, tg)
        torch.cuda.synchronize()
        return m.store.P.clone(), m.store.M.clone(), (len(m.grad_buckets()) if bucketed else 0)

    p0, m0, _ = run(False)
    p1, m1, nb = run(True)
    assert nb > 3
    # weight gradients are split-K sums through float atomics: their order, hence the last bits, differ from run to run
    dp_, dm_ = float((p0 - p1).abs().max()), float((m0 - m1).abs().max())
    assert dp_ <= 1e-5 * float(p0.abs().max()) and dm_ <= 1e-4 * float(m0.abs().max()), (dp_, dm_)


def test_graphed_train_step_matches_eager():
    """One hipGraph replay per optimiser step (both streams captured) must train exactly like the eager step."""
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand((4, 32, 32, 3), generator=g).cuda() for _ in range(4)]

    def run(graphed):
        m = build("f16")
        tgs = [[(x[..., :1] > 0.5).float() for _ in m.out_heads] for x in xs]
        losses = []
        step = m.graphed_train_step(xs[0], tgs[0]) if graphed else None
        for x, tg in zip(xs[1:] if graphed else xs, tgs[1:] if graphed else tgs):
            lv = step(x, tg) if graphed else m.train_step(x, tg)
            losses.append(float(lv.cpu()))
        torch.cuda.synchronize()
        return m.get_weights(), losses, m.adam_t

    w0, l0, t0 = run(False)
    w1, l1, t1 = run(True)          # the capture warm-up is the step on xs[0]
    assert t0 == t1 == 4
    assert np.allclose(l0[1:], l1, rtol=2e-3, atol=1e-4), (l0, l1)
    worst = max(float(np.abs(w0[k] - w1[k]).max()) for k in w0)
    assert worst < 5e-3, worst       # Adam moves a weight by ~lr = 2e-3 per step; atomics reorder the last bits of the gradients

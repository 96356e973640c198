"""GPU: the data-parallel step with real kernels - two ranks (gloo, both on cuda:0: a one-GPU box cannot host two RCCL
ranks) train on different shards; the result must be what one process gets by averaging the two replicas' gradients by
hand (per-replica BN statistics, one sum all-reduce, 1/world folded into the optimiser)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "tf-1d-2d-segmentation-end2endpipelines_amd"
STEPS = 2


def _model(pkg):
    m = pkg.unet_model_builder("UNetPP", 32, 32, 16, 2, ds=1, num_channels=3, train_mode="from_scratch").ResNet50()
    m.build(2, dtype="f32")                  # fp32 mode: the comparison is about the exchange, not fp16 rounding
    m.compile(loss="mse", lr=1e-3)
    return m


def _shards():
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand((2, 32, 32, 3), generator=g) for _ in range(2)]
    return xs, [(x[..., :1] > 0.5).float() for x in xs]


def _worker(rank, world, port, overlap, grad_dtype, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import torch.distributed as dist
        pkg = importlib.import_module(PKG)
        dp = importlib.import_module(PKG + ".dp")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        M = importlib.import_module(PKG + ".model")
        m = _model(pkg)
        m.set_weights({p.name: M.init_param(p, 100 + rank) for p in m.graph.param_specs()})    # the broadcast has to make the replicas identical
        d = dp.DataParallel(m, bucket_elems=20000, overlap=overlap, grad_dtype=grad_dtype)
        # (grad_dtype None: the default - a parity-mode model exchanges the fp32 buffer itself)
        assert d.grad_dtype == ((grad_dtype or "f32") if overlap else "f32") and m.nonfinite_updates() == 0
        xs, ts = _shards()
        x, t = xs[rank].cuda(), ts[rank].cuda()
        losses = [float(d.train_step(x, [t] * len(m.out_heads)).cpu()) for _ in range(STEPS)]
        m._join_side()
        torch.cuda.synchronize()
        q.put((rank, m.store.P.cpu().numpy(), losses, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:      # pragma: no cover
        import traceback
        q.put((rank, None, None, traceback.format_exc()))


@pytest.mark.parametrize("overlap,grad_dtype", [(True, None), (True, "f32"), (False, "f32"), (True, "bf16")])
def test_two_ranks_equal_hand_averaged_replicas(overlap, grad_dtype):
    """grad_dtype None = the default path: the model is in the parity mode, so the fp32 gradient buffer itself is exchanged and
    1 GPU and N GPUs differ by the summation order only (1e-6).  grad_dtype bf16 (the default of a THROUGHPUT-mode model with
    more than one rank, opt-in here): every slice is packed to bf16 before the all-reduce, summed in bf16 and widened by the
    optimiser kernel - the hand-made reference rounds each replica's gradient the same way."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, grad_dtype, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, P, losses, err in res:
        assert err is None, err
    assert np.array_equal(res[0][1], res[1][1])                 # the replicas stay bit-identical

    # the same two steps in one process: two replicas, gradients averaged by hand
    import segamd
    M = importlib.import_module(PKG + ".model")
    a, b = _model(segamd), _model(segamd)
    a.set_weights({p.name: M.init_param(p, 100) for p in a.graph.param_specs()})
    b.set_weights(a.get_weights())
    xs, ts = _shards()
    ref_losses = []
    for _ in range(STEPS):
        gs = []
        for m, x, t in ((a, xs[0], ts[0]), (b, xs[1], ts[1])):
            m.forward(x.cuda(), training=True)
            ref_losses.append(float(m.compute_loss([t.cuda()] * len(m.out_heads)).cpu()))
            m.backward(None)
            torch.cuda.synchronize()
            gs.append(m.store.G.clone())
        if grad_dtype == "bf16":          # what the exchange does: bf16(g0) + bf16(g1) rounded to bf16, then x 1/world in fp32
            ssum = (gs[0].bfloat16() + gs[1].bfloat16()).float()
        else:
            ssum = gs[0] + gs[1]
        for m in (a, b):
            m.store.G.copy_(ssum * 0.5)
            m.optimizer_step()
            m._join_side()
    torch.cuda.synchronize()
    want = a.store.P.cpu().numpy()
    # (Adam normalises the step: a gradient element whose bf16 sum rounds the other way moves its weight by up to ~lr)
    assert np.abs(res[0][1] - want).max() < (1e-6 if grad_dtype != "bf16" else 2e-3), np.abs(res[0][1] - want).max()
    rel = 1e-5 if grad_dtype != "bf16" else 1e-3
    assert res[0][2] == pytest.approx(ref_losses[0::2], rel=rel) and res[1][2] == pytest.approx(ref_losses[1::2], rel=rel)


# ---- random builder configurations under the data-parallel exchange (round 5): odd parameter counts against bucket and optimiser-slice
# boundaries, Dense bottlenecks, ConvLSTM gate padding, one- and many-channel heads.  The configurations come from the differential
# fuzzer's sampler (tools/fuzz_models.py) with a fixed seed; every rank rebuilds the same sequence.
DP_FUZZ_SEED, DP_FUZZ_N = 21, 14


def _fuzz_cases():
    import random
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_models as FZ
    rng = random.Random(DP_FUZZ_SEED)
    out = []
    for i in range(DP_FUZZ_N):
        two_d = rng.random() < 0.65
        call, make, xf, loss = (FZ.sample_2d if two_d else FZ.sample_1d)(rng)
        out.append((i, call, make, xf, loss, rng.choice([3000, 8000, 20000, 100000])))
    return out, FZ


def _fuzz_batches(i, xf, model, loss, FZ):
    import random
    torch.manual_seed(1000 + i)
    xs = [xf() for _ in range(2)]
    ts = [FZ.targets_for(model, xs[0], loss, random.Random(i)) for _ in range(2)]
    return xs, ts


def _fuzz_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import torch.distributed as dist
        dp = importlib.import_module(PKG + ".dp")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cases, FZ = _fuzz_cases()
        out = {}
        for i, call, make, xf, loss, bucket in cases:
            try:
                m = make()
                x0 = xf()
                m.build(x0.shape[0], dtype="f32")
            except (ValueError, NameError):          # refused as the reference refuses it (same on every rank)
                continue
            m.compile(loss=loss, lr=1e-3)
            d = dp.DataParallel(m, bucket_elems=bucket, overlap=True)
            xs, ts = _fuzz_batches(i, xf, m, loss, FZ)
            losses = [float(d.train_step(xs[rank].cuda(), [t.cuda() for t in ts[rank]]).cpu()) for _ in range(STEPS)]
            m._join_side()
            torch.cuda.synchronize()
            out[i] = (m.store.P.cpu().numpy(), losses, int(m.nonfinite_updates()))
        q.put((rank, out, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:      # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_random_configurations_under_data_parallel():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fuzz_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for _rank, _out, err in res:
        assert err is None, err
    r0, r1 = res[0][1], res[1][1]
    assert sorted(r0) == sorted(r1) and len(r0) >= DP_FUZZ_N // 2
    cases, FZ = _fuzz_cases()
    checked = 0
    for i, call, make, xf, loss, bucket in cases:
        if i not in r0:
            continue
        assert np.array_equal(r0[i][0], r1[i][0]), call                  # the replicas stay bit-identical
        a, b = make(), make()
        x0 = xf()
        for m in (a, b):
            m.build(x0.shape[0], dtype="f32")
            m.compile(loss=loss, lr=1e-3)
        b.set_weights(a.get_weights())
        xs, ts = _fuzz_batches(i, xf, a, loss, FZ)
        ref_losses, skipped = [], 0
        for _ in range(STEPS):
            gs = []
            for m, x, t in ((a, xs[0], ts[0]), (b, xs[1], ts[1])):
                m.forward(x.cuda(), training=True)
                ref_losses.append(float(m.compute_loss([v.cuda() for v in t]).cpu()))
                m.backward(None)
                torch.cuda.synchronize()
                gs.append(m.store.G.clone())
            for m in (a, b):
                m.store.G.copy_((gs[0] + gs[1]) * 0.5)
                m.optimizer_step()
                m._join_side()
            skipped += int(a.nonfinite_updates(reset=True))
        if skipped or r0[i][2]:
            continue                  # a first-step overflow is skipped element by element: not comparable, and rare at a target of 32
        torch.cuda.synchronize()
        want = a.store.P.cpu().numpy()
        assert np.abs(r0[i][0] - want).max() < 2e-6 * max(1.0, np.abs(want).max()), (call, bucket, np.abs(r0[i][0] - want).max())
        assert r0[i][1] == pytest.approx(ref_losses[0::2], rel=1e-5) and r1[i][1] == pytest.approx(ref_losses[1::2], rel=1e-5), call
        checked += 1
    assert checked >= DP_FUZZ_N // 2

"""Data-parallel training: one process per GPU, minibatch sharded across ranks (weak scaling, bs/GPU fixed),
per-replica BatchNorm statistics (the reference has no SyncBN), and ONE exchange step per optimiser step: a sum
all-reduce of the flat gradient buffer over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm).
What travels follows the model's numeric mode: a parity-mode model (dtype f32) exchanges the fp32 buffer itself - the
same arithmetic on 1 and on N GPUs up to the summation order; a throughput-mode model (dtype f16) exchanges bf16 slices
(half the bytes per link, SURVEY.md 8e mitigation 1: C3 795 MB -> 398 MB per step): a finished fp32 slice is packed by
`seg_cast_f32_bf16`, summed by RCCL in bf16, and the optimiser kernel widens it back to fp32 - optimiser state and
master weights never leave fp32.  `grad_dtype=` / SEG_DP_GRAD_DTYPE override; a backend that rejects bf16 all-reduce
(probed once at construction) falls back to f32.

The reference is single-device (no tf.distribute anywhere); this is the data-parallel
train_step the north star adds around `2DCNN/Train.py:394-415`.

All parameters live in one flat buffer (model.ParamStore), so the collective is issued on large contiguous
slices ("buckets") instead of per-layer tensors: xGMI is point-to-point (7 links x ~153 GB/s per GPU), and a
few large messages keep every link busy.  Gradients are averaged by the Adam kernel (gmul = 1/world_size).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns the local rank."""
    if dist.is_initialized():
        return int(os.environ.get("LOCAL_RANK", 0))
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    kw = {}
    if backend == "gloo":
        # one node by contract: gloo's pairwise connections over the loopback interface, whatever the container's hostname resolves to
        # (a hostname that resolves to an unreachable address makes the connect sit in its 30-minute default), and a rendezvous that
        # fails does so within minutes.  The value is also the timeout of every gloo collective afterwards, so it must cover the
        # longest stretch one rank may spend alone between two collectives (rank 0 validating / writing a checkpoint in harness.fit,
        # a first-step library build): 15 minutes unless SEG_DIST_TIMEOUT_S says otherwise.
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("SEG_DIST_TIMEOUT_S", "900")))
    elif os.environ.get("SEG_DIST_TIMEOUT_S"):
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ["SEG_DIST_TIMEOUT_S"]))
    dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", 0)),
                            world_size=int(os.environ.get("WORLD_SIZE", 1)), **kw)
    return local_rank


def bucket_bounds(numel: int, bucket_elems: int):
    """Contiguous [lo, hi) slices covering [0, numel)."""
    out, lo = [], 0
    while lo < numel:
        hi = min(numel, lo + bucket_elems)
        out.append((lo, hi))
        lo = hi
    return out


def allreduce_flat(buf: torch.Tensor, bucket_elems: int = 64 << 20, group=None, force: bool = False):
    """In-place sum all-reduce of a flat tensor in large contiguous buckets (async launches, one wait)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    works = [dist.all_reduce(buf[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
             for lo, hi in bucket_bounds(buf.numel(), bucket_elems)]
    for w in works:
        w.wait()


def broadcast_flat(buf: torch.Tensor, src: int = 0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(buf, src=src, group=group)


class DataParallel:
    """Wraps a built SegModel: identical initial weights on every rank, gradient all-reduce before Adam.

    overlap=True (default): the backward sweep hands over each slice of the flat gradient buffer as soon as the
    ops that own it are done (model.grad_buckets(): slices follow op boundaries; the backward order is reverse
    creation order, so the decoder's slices become final first) and the all-reduce of that slice is launched
    asynchronously on RCCL's stream while the remaining backward kernels keep running; Adam runs per slice as soon as
    that slice's exchange has finished."""

    def __init__(self, model, bucket_elems: int = 32 << 20, group=None, overlap: bool = True, force_exchange: bool = False,
                 grad_dtype: Optional[str] = None):
        """force_exchange: take the exchange path even in a group of one rank (tests: the RCCL launches, their stream
        ordering against the side stream and the per-slice optimiser can then be exercised on a single GPU).
        grad_dtype: "bf16" | "f32" - what the all-reduce moves (default: SEG_DP_GRAD_DTYPE, else f32 for a parity-mode model,
        bf16 for a throughput-mode model when there is an exchange with more than one rank)."""
        self.model, self.group, self.bucket_elems = model, group, bucket_elems
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.pending = []
        self.force_exchange = force_exchange
        model.world_size = self.world_size
        model.bucket_elems = bucket_elems
        exchange = self.world_size > 1 or (force_exchange and dist.is_initialized())
        if grad_dtype is None:
            grad_dtype = os.environ.get("SEG_DP_GRAD_DTYPE") or (
                "bf16" if (exchange and self.world_size > 1 and getattr(model, "dtype", "f16") == "f16") else "f32")
        if grad_dtype not in ("bf16", "f32"):
            raise ValueError("grad_dtype must be 'bf16' or 'f32'")
        if grad_dtype == "bf16" and exchange and not self._bf16_allreduce_works(model.store.G.device, group):
            grad_dtype = "f32"
        self.grad_dtype = grad_dtype if (exchange and overlap) else "f32"      # (the non-overlapped path exchanges the buffer itself)
        model.grad_dtype = self.grad_dtype
        if self.grad_dtype == "bf16":
            model.store.G16 = torch.zeros(model.store.G.numel(), dtype=torch.bfloat16, device=model.store.G.device)
        if overlap and exchange:
            model.bucket_hook = self._bucket_ready
            model.bucket_waits = self._take_pending
        elif exchange:
            model.allreduce = self._allreduce       # (a single rank keeps the model's own single-GPU optimiser path)
        if self.world_size > 1 and exchange:
            from . import flags as _F
            if _F.DP_BUSBW_MEASURE:
                model.dp_busbw_gbs = self.measure_busbw(model.store.G.device, group)
        if self.world_size > 1:
            # identical replicas: weights, BN moving statistics AND the optimiser's view of the run (state buffers, step
            # count, Nadam's momentum product, learning rate) come from rank 0 - a model that has already taken steps, or
            # ranks that loaded different checkpoints, would otherwise apply different updates to the same summed gradient
            st = model.store
            model._join_side()
            for buf in (st.P, st.S, st.M, st.V):
                broadcast_flat(buf, 0, group)
            scal = torch.tensor([float(model.adam_t), float(model._u_product), float(model.lr)], dtype=torch.float64, device=st.P.device)
            broadcast_flat(scal, 0, group)
            model.adam_t, model._u_product, model.lr = int(scal[0].item()), float(scal[1].item()), float(scal[2].item())
            model._refresh_copies()

    @staticmethod
    def measure_busbw(device, group=None) -> float:
        """Bus bandwidth (GB/s, nccl-tests' definition: 2 (w - 1) / w x bytes / time) of ONE timed fp32 all-reduce on this job's fabric -
        64 MB on a GPU backend, 4 MB on gloo -, the MINIMUM over the ranks: the hold-back schedule (SegModel.bucket_holds) decides from
        it which gradient slices wait for the end of the backward sweep, and every rank must decide the same (the collectives are
        issued in that order).  Replaces the assumed 300 GB/s unless SEG_DP_BUSBW_GBS pins a value."""
        import time
        w = dist.get_world_size(group)
        on_gpu = device.type == "cuda"
        n = (16 << 20) if on_gpu else (1 << 20)
        buf = torch.zeros(n, dtype=torch.float32, device=device)
        dist.all_reduce(buf[: n // 16], group=group)                 # warm-up: connections, RCCL's channel set-up
        if on_gpu:
            torch.cuda.synchronize(device)
        dist.barrier(group)
        t0 = time.perf_counter()
        dist.all_reduce(buf, group=group)
        if on_gpu:
            torch.cuda.synchronize(device)
        dt = max(time.perf_counter() - t0, 1e-6)
        bw = torch.tensor([2.0 * (w - 1) / w * n * 4 / dt / 1e9], dtype=torch.float64, device=device)
        dist.all_reduce(bw, op=dist.ReduceOp.MIN, group=group)
        return float(bw.item())

    @staticmethod
    def _bf16_allreduce_works(device, group) -> bool:
        """One tiny bf16 all-reduce: RCCL sums bf16; a gloo build may not (every rank sees the same outcome)."""
        try:
            t = torch.ones(8, dtype=torch.bfloat16, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return bool(float(t[0]) == float(dist.get_world_size(group)))
        except (RuntimeError, ValueError, TypeError):
            return False

    def _allreduce(self, flat_grads: torch.Tensor):
        allreduce_flat(flat_grads, self.bucket_elems, self.group, force=self.force_exchange)

    def _bucket_ready(self, lo: int, hi: int):
        """G[lo:hi] is final on the calling stream: pack it (bf16 mode) and start its all-reduce."""
        st = self.model.store
        if self.grad_dtype == "bf16":
            from . import model as _m
            _m.call("seg_cast_f32_bf16", _m.C.c_void_p(st.G.data_ptr() + lo * 4), _m.C.c_void_p(st.G16.data_ptr() + lo * 2), hi - lo, _m._stream())
            buf = st.G16[lo:hi]
        else:
            buf = st.G[lo:hi]
        self.pending.append((lo, hi, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _take_pending(self):
        """[(lo, hi, work)] in issue order; the optimiser waits for each and updates that slice (Adam per bucket: the
        update of the slices that arrived first runs under the exchange of the last ones)."""
        out, self.pending = self.pending, []
        return out

    def train_step(self, x_local, targets_local, loss=None):
        return self.model.train_step(x_local, targets_local, loss)

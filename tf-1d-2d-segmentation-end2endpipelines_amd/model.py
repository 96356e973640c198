"""SegModel - executes a layer graph (graph.py) on one MI355X through the C ABI in include/seg_hip.h.

This object stands where the reference has a ``tf.keras.Model`` (returned by the encoder-named methods,
`TensorFlow/2DCNN/models/unet_variants.py:1109-1114`) and offers the calls the reference's scripts make on it
(SURVEY.md section 8b): ``predict`` (2DCNN/Test.py:153,161), ``get_weights``/``set_weights``/``count_params``
(Train.py:354-367), a ``train_step`` = Keras' default fit step (Train.py:322-325, 394-415: forward with
training-mode BN, sum of per-output losses with unit weights, gradients, Adam), and ``summary``.

torch is used for device memory, streams and (in dp.py) torch.distributed - every FLOP of the path runs in the
hand-written HIP kernels.  There is no CPU fallback: constructing a SegModel without a GPU / without
libseg_hip.so raises.

Lowering (Keras-level nodes -> fused launches)
  conv [-> bn] [-> act] [-> maxpool]   one implicit-GEMM launch (virtual concat, bias, BN partial statistics in
                                       the epilogue) + bn_finalize + bn_act_apply (activation and 2x2 pool fused)
  convT [-> bn] [-> act]               the same kernel with 4 (2 in 1D) sub-pixel classes; LeakyReLU in the epilogue
  concat                               never materialised when it feeds a conv / ConvLSTM
  convlstm                             implicit GEMM (4F gate pre-activations) + gate kernel
  1x1 conv that is a model output      head kernel (fp32 logits/probabilities, sigmoid / softmax fused)
  add [-> act], mul, upsample, maxpool, standalone bn/act      one HBM-bound kernel each
Backward mirrors it: bn/act backward in two passes, data gradients through the same implicit-GEMM kernel with
transposed weights (one launch per concat segment, accumulating where a tensor has several consumers), weight
gradients through the transposed-read MFMA kernel into ONE flat fp32 gradient buffer (single all-reduce, single
Adam launch).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from . import flags as F
from . import planning
from ._ffi import TransposeDesc, ACT, LOSS, OPTIMIZER, Geom, IgemmFwdArgs, IgemmWgradArgs
from .graph import Graph, Node, ParamSpec
from .splitfmt import split_pack, split_unpack
# the pieces of the engine live in planning.py (geometry, tile / variant rules), ops.py (tensors, parameter store, the launch groups) and
# flags.py (switches); their names stay importable from here (tests, tools and bench.py address them as `model.<name>`)
from .planning import (_r8, _r32, _cdiv, _same_pad, _fill_geom, geom_conv, geom_conv_dgrad, geom_convT, geom_convT_dgrad, pack_gemm_weight,
                       unpack_gemm_weight, pick_tile_fwd, wgrad_taps_per_block, fwd_halo_patch, pick_fwd, plan_fwd, fwd_workspace_bytes,
                       wgrad_halo_patch, wgrad_halo_class_groups, pick_split_halo, pick_tile_wgrad, fits_lds_dma, LDS_DMA_LIMIT, TAIL_ROW_LIMIT, _tail_parts)
from .ops import (_DryRun, _dev, KernelProfile, _Mode, call, _stream, _p, DevTensor, Slot, ParamStore, Op, InputOp, ConvOp, GatherDgrad,
                  plan_gathers, _pad_to, HeadOp, WideHeadOp, ReshapeOp, BNOp, ConcatOp, AddOp, MulOp, UpsampleOp, PoolOp)


# ------------------------------------------------------------------------------------------------ init
def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def init_param(spec, seed) -> np.ndarray:
    """Keras initialisers by name (he_uniform: unet_variants.py:7; glorot_uniform default; he_normal: :149)."""
    r = _rng(seed, spec.name)
    fi, fo = spec.fan
    shape, init = spec.shape, spec.init
    if init == "zeros":
        return np.zeros(shape, np.float32)
    if init == "ones":
        return np.ones(shape, np.float32)
    if init == "he_uniform":
        lim = math.sqrt(6.0 / fi)
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "glorot_uniform":
        lim = math.sqrt(6.0 / (fi + fo))
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "he_normal":
        std = math.sqrt(2.0 / fi) / 0.87962566103423978
        v = np.clip(r.standard_normal(shape), -2.0, 2.0)
        return (v * std).astype(np.float32)
    if init == "orthogonal":
        rows, cols = int(np.prod(shape[:-1])), int(shape[-1])
        a = r.standard_normal((max(rows, cols), min(rows, cols)))
        q, rr = np.linalg.qr(a)
        q = q * np.sign(np.diag(rr))
        if rows < cols:
            q = q.T
        return q.reshape(shape).astype(np.float32)
    if init == "unit_forget_bias":
        b = np.zeros(shape, np.float32)
        F4 = shape[0] // 4
        b[F4:2 * F4] = 1.0
        return b
    raise ValueError(init)


# ------------------------------------------------------------------------------------------------ model
class SegModel:
    def __init__(self, graph: Graph, batch_size: Optional[int] = None, seed: int = 0, dtype: str = "f32"):
        """dtype: 'f32' = the parity mode - tensors stored as split fp16 pairs (x ~= hi + lo / 2048, ~22 significant bits), every
        product three fp16 MFMAs into fp32 accumulators: outputs within 1e-3 of the reference's fp32 arithmetic, equal masks;
        'f16' = fp16 activations / MFMA operands with fp32 accumulation (throughput mode: 2.3x faster, misses the 1e-3 tolerance).
        RANGE: both modes keep fp16's dynamic range - an input, activation or gradient with |x| >= 65504 becomes Inf / NaN (the
        reference's fp32 does not).  Inputs are checked (check_input_range: predict / fit raise), non-finite gradients are skipped and
        counted by the optimiser (nonfinite_updates); activations are bounded by BatchNormalization in every model of the reference."""
        if dtype not in ("f16", "f32"):
            raise ValueError("dtype must be 'f16' or 'f32'")
        self.graph = graph
        self.dtype = dtype
        self.seed = seed
        self.name = graph.name
        self.built = False
        self.training = True
        self._x = None
        self.adam_t = 0
        self.lr, self.beta_1, self.beta_2, self.epsilon = 2e-4, 0.9, 0.999, 1e-7
        self.loss_name = "mse"
        self.loss_weights = None       # per-output loss weights (compile); None = unit weights
        self.optimizer = "Adam"        # one of utils/tf_optimizers.py's names
        self._u_product = 1.0          # Nadam: running product of the momentum schedule
        self.metric_names: List[str] = []   # compile(metrics=[...]): names of utils/tf_metrics.py, evaluated by harness.fit
        self._label_flag = None
        self.world_size = 1
        self.comm_timing, self._comm_events = False, []     # bench.py --gpus N: event-timed waits for the gradient slices
        self.grad_dtype = "f32"        # what the data-parallel exchange moves: "f32" (the flat buffer itself) or "bf16" (store.G16)
        # the loss scale: max |dLoss/dOut| is scaled into [target / 2, target) before the backward pass (a power of two chosen on the
        # device every step).  Dynamic: an optimiser step that had to skip non-finite gradient elements halves the target, a run of
        # clean steps doubles it back up to the mode's default (loss_scale_feedback; train_step does it, one step late, without a sync)
        self.grad_target_max = F.GRAD_TARGET_F32 if dtype == "f32" else F.GRAD_TARGET
        self.grad_target = self.grad_target_max
        self.auto_loss_scale = True
        self.loss_scale_growth_interval = F.LOSS_SCALE_GROWTH_INTERVAL
        self._ls_clean, self._ls_skipped = 0, 0
        self.allreduce = None          # set by dp.DataParallel: whole-buffer exchange after backward
        self.bucket_hook = None        # set by dp.DataParallel(overlap=True): fn(lo, hi) called as soon as G[lo:hi] is final
        self.bucket_waits = None       # ... and fn() -> [(lo, hi, work)] handing the pending all-reduces to the optimiser
        self._g_clean = False          # the flat gradient buffer is already zero (cleared on the side stream after Adam)
        self._param_events: list = []  # (first op index, event) of optimiser slices still running on the side stream
        self._graph_mode = False       # inside capture / replay of a hipGraph of the training step
        self._alpha_dev = None
        self._adam_slices = None
        self.side_stream = None        # second HIP stream for the weight-gradient kernels (created in build())
        self.use_side_stream = os.environ.get("SEG_WGRAD_STREAM", "1") != "0"
        self.bucket_elems = 32 << 20
        self.dp_busbw_gbs = None       # all-reduce bus bandwidth dp.DataParallel measured on this job's fabric (None: flags.DP_BUSBW_GBS)
        self.backward_mode, self.backward_spec, self.bwd_d, self.bwd_w = None, "d3w3", 3, 3       # build(backward=): MFMAs per product of the backward GEMMs (parity mode)
        if batch_size is not None:
            self.build(batch_size)

    # ---------------------------------------------------------------- Keras-like surface
    @property
    def output_names(self):
        return list(self.graph.output_names)

    def count_params(self) -> int:
        return self.graph.count_params()

    def summary(self, print_fn=print):
        print_fn(self.graph.summary())

    def graphed_train_step(self, x: torch.Tensor, targets: Sequence[torch.Tensor], loss: Optional[str] = None):
        """Captures one optimiser step (forward, loss, backward, Adam, derived operands - both streams) into a hipGraph
        and returns step(x, targets) -> loss that replays it: one graph launch instead of ~150-270 kernel launches,
        which is what small models are bound by (64x64 w16 d3: 1.5 ms eager).  Single GPU; shapes and dtypes are fixed
        by the example batch; x / targets of later calls are copied into the captured input buffers."""
        if self.world_size != 1 or self.bucket_hook is not None:
            raise NotImplementedError("graph capture of the data-parallel step is not implemented")
        targets = list(targets) if isinstance(targets, (list, tuple)) else [targets]
        sx = self._check_x(x).clone()
        st_ = [t.to(device=_dev(), dtype=torch.float32).contiguous().clone() for t in targets]
        self._alpha_dev = torch.zeros(1, dtype=torch.float32, device=_dev())
        self._join_side()
        torch.cuda.synchronize()

        def set_alpha():
            self.adam_t += 1
            self._alpha_dev.fill_(self.lr * math.sqrt(1.0 - self.beta_2 ** self.adam_t) / (1.0 - self.beta_1 ** self.adam_t))

        # the warm-up below is a real optimiser step on the example batch (PyTorch's capture protocol needs one eager run):
        # weights, optimiser state, BN moving statistics and the step count are put back afterwards, so that capturing
        # does not train the model
        st = self.store
        snap = (st.P.clone(), st.M.clone(), st.V.clone(), st.S.clone(), self.adam_t, self._u_product)
        self._graph_mode = True
        try:
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):                     # one eager step on a side stream (PyTorch's capture protocol)
                set_alpha()
                self.train_step(sx, st_, loss)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):                      # records the launches, runs nothing
                lv = self.train_step(sx, st_, loss)
        finally:
            self._graph_mode = False
        st.P.copy_(snap[0]); st.M.copy_(snap[1]); st.V.copy_(snap[2]); st.S.copy_(snap[3])
        self.adam_t, self._u_product = snap[4], snap[5]
        del snap
        self._refresh_copies()
        st.G.zero_()
        self._g_clean = True          # what the captured step expects on entry (it ends by clearing the buffer itself)
        torch.cuda.synchronize()

        def step(xn, tn):
            tn = list(tn) if isinstance(tn, (list, tuple)) else [tn]
            # an eager step before this replay may have left Adam slices / the operand rebuild running on the side stream,
            # or a backward pass without an optimiser step may have left gradients in the buffer
            self._join_side()
            if not self._g_clean:
                st.G.zero_()
            sx.copy_(self._check_x(xn))
            for d, t in zip(st_, tn):
                d.copy_(t)
            set_alpha()
            graph.replay()
            self._g_clean = True          # the replayed step ended by clearing the gradient buffer
            for op in self.ops:           # the replay moved the weights and the moving statistics on the device: whatever the host
                op.post_step()            # caches about them (the folded inference-mode BN coefficients) is stale
            return lv
        step.graph = graph
        return step

    def compile(self, loss="mse", optimizer="Adam", lr=2e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, metrics=None, loss_weights=None):
        """model.compile(...) (2DCNN/Train.py:322-325).  loss: any of the 16 names `TFLosses` takes
        (utils/tf_losses.py:9-39, e.g. 'MeanSquaredLogarithmicError', the Train_Configs.ini default) or its short form
        ('mse', 'bce', 'cce', 'mae', 'msle', ...: _ffi.LOSS); mean reduction, applied to every output and summed with
        loss_weights (a list in output order or {output name: weight}; default: unit weights as in Train.py:322 - the 1D
        notebook passes 1 - 0.1 k).  optimizer: 'Adam' | 'Adadelta' | 'Adagrad' | 'Adamax' | 'FTRL' | 'Nadam' | 'RMSprop' |
        'SGD' with the fixed hyper-parameters of utils/tf_optimizers.py:10-26 (changing it restarts the state).  metrics: names of utils/tf_metrics.py (metrics.py lists the available ones); `fit`
        logs them under Keras' names (`mean_squared_error`, `val_mean_squared_error`, with several outputs
        `<output>_...`).  Optimiser: Keras Adam."""
        if loss not in LOSS:
            raise ValueError("Please select a valid loss function. Check for spelling mistakes, capital/small letters, etc.")
        if optimizer != "Adam" and optimizer not in OPTIMIZER:
            if str(optimizer).lower() == "adam":
                optimizer = "Adam"
            else:
                raise ValueError("Please select a valid optimizer function. Check for spelling mistakes, capital/small letters, etc.")
        if optimizer != self.optimizer:
            self.optimizer = optimizer
            self.adam_t = 0
            if self.built:
                self._init_optimizer_state()
        names = self.output_names
        if loss_weights is None:
            lw = [1.0] * len(names)
        elif isinstance(loss_weights, dict):
            unknown = [k for k in loss_weights if k not in names]
            if unknown:
                raise ValueError(f"loss_weights for unknown output(s) {unknown}; the outputs are {names}")
            lw = [float(loss_weights.get(n, 1.0)) for n in names]
        else:
            lw = [float(w) for w in loss_weights]
            if len(lw) != len(names):
                raise ValueError(f"{len(lw)} loss_weights for {len(names)} outputs")
        self.loss_weights = lw
        if metrics:
            from .metrics import log_names
            self.metric_names = [metrics] if isinstance(metrics, str) else list(metrics)
            try:
                log_names(self.metric_names, names)
            except KeyError as e:
                raise ValueError(f"metric {e.args[0]!r} is not available (see metrics.py)") from None
        else:
            self.metric_names = []
        self.loss_name, self.lr, self.beta_1, self.beta_2, self.epsilon = loss, lr, beta_1, beta_2, epsilon

    # ---------------------------------------------------------------- build / lowering
    def _enter(self):
        _Mode.sfx = "" if self.dtype == "f16" else "_f32"

    def build(self, batch_size: int, dtype: Optional[str] = None, backward: Optional[str] = None):
        """backward (parity mode only; the throughput mode has one MFMA per product everywhere): how many fp16 MFMAs a product of the
        BACKWARD GEMMs costs - "auto" (flags.BACKWARD: the default), "3mfma", "2mfma" or an explicit "<d|b><3|2>w<3|2|1>"; see flags.py.
        The forward pass is the same bits in every mode.  `backward_mode` keeps what was asked for, `backward_spec` what it resolved to."""
        backward = backward or self.backward_mode or F.BACKWARD
        spec = {"3mfma": "d3w3", "2mfma": "b2w2"}.get(backward, backward)
        if spec == "auto":
            spec = "b2w2" if (self.graph.longest_conv_chain() <= F.BACKWARD_AUTO_CHAIN and self.graph.plain_conv_bn_stack()) else "d3w2"
        if spec not in ("d3w3", "d3w2", "d3w1", "d2w3", "d2w2", "d2w1", "b2w3", "b2w2", "b2w1"):
            raise ValueError("backward must be 'auto', '3mfma', '2mfma' or '<d|b><3|2>w<3|2|1>'")
        self.backward_mode, self.backward_spec = backward, spec
        if dtype is not None:
            if dtype not in ("f16", "f32"):
                raise ValueError("dtype must be 'f16' or 'f32'")
            if dtype != self.dtype and self.grad_target == self.grad_target_max:
                self.grad_target = F.GRAD_TARGET_F32 if dtype == "f32" else F.GRAD_TARGET
            self.grad_target_max = F.GRAD_TARGET_F32 if dtype == "f32" else F.GRAD_TARGET
            self.dtype = dtype
        # MFMAs per product of the parity mode's data / weight gradients (3 = no operand halved; the throughput mode has no split operands to halve)
        self.bwd_d, self.bwd_w = (int(spec[1]), int(spec[3])) if self.dtype == "f32" else (3, 3)
        self.bwd_d_flag = 0 if self.bwd_d != 2 else (_ffi.SEG_VARIANT_B_HI if spec[0] == "b" else _ffi.SEG_VARIANT_A_HI)     # which operand a two-MFMA data gradient halves
        self._enter()
        DevTensor.dtype = torch.float16 if self.dtype == "f16" else torch.float32
        if not _DryRun.on:
            if not torch.cuda.is_available():
                raise RuntimeError("SegModel needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
            _ffi.lib()
        g = self.graph
        self.N = int(batch_size)
        self.tensors: Dict[str, DevTensor] = {}
        DevTensor.registry = self.tensors
        self.store = ParamStore()
        self.ops: List[Op] = []
        self.fwd_ws_users: list = []           # argument blocks of forward / data-gradient launches (split reductions share one workspace)
        self.heads: List[HeadOp] = []
        self.gs = torch.ones(2, dtype=torch.float32, device=_dev())        # [grad scale S, 1/S]
        self.amax = torch.zeros(1, dtype=torch.float32, device=_dev())
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=_dev())
        self.input_overflow = torch.zeros(1, dtype=torch.int32, device=_dev())     # input elements no fp16 / split-pair element can hold (seg_pack_input)
        cons = g.consumers()
        live = g.live_nodes()
        outset = set(g.outputs)
        val: Dict[int, object] = {}        # node idx -> DevTensor | list[(DevTensor, C)] (virtual concat)
        done = set()
        N = self.N

        def only_consumer(i, kind):
            c = cons[i]
            if len(c) == 1 and g.nodes[c[0]].kind == kind and i not in outset:
                return g.nodes[c[0]]
            return None

        def segs_of(i) -> List[Tuple[DevTensor, int]]:
            v = val[i]
            return v if isinstance(v, list) else [(v, v.C)]

        def tensor_of(i) -> DevTensor:
            v = val[i]
            if isinstance(v, list):
                raise NotImplementedError(f"node {g.nodes[i].name}: a concat feeding this layer type must be materialised "
                                          "(not implemented yet)")
            return v

        self._op_grad_range: Dict[int, Tuple[int, int]] = {}     # index in self.ops -> [lo, hi) of its slice of G
        n_ops_seen, n_train_seen = 0, 0

        def note_ranges():
            nonlocal n_ops_seen, n_train_seen
            if len(self.ops) > n_ops_seen:
                if self.store.n_train > n_train_seen:
                    # owner = the FIRST op this graph node appended (a node may lower to several ops, e.g. a wide head =
                    # conv + activation): the forward pass must have these parameters before that op runs, and the backward
                    # sweep (descending op index) has finished their gradients only after it
                    self._op_grad_range[n_ops_seen] = (n_train_seen, self.store.n_train)
                n_ops_seen, n_train_seen = len(self.ops), self.store.n_train

        for i in live:
            note_ranges()
            if i in done:
                continue
            n = g.nodes[i]
            k = n.kind
            if k == "input":
                t = DevTensor(n.name, N, n.shape[0], n.shape[1], n.shape[2])
                self.ops.append(InputOp(self, t, n.shape[2]))
                val[i] = t
            elif k == "concat":
                segs = []
                for j in n.inputs:
                    segs.extend(segs_of(j))
                if len(segs) > _ffi.MAX_SEG:
                    # the GEMM kernels walk at most MAX_SEG (8) K segments (UNet++ / UNet 3+ of depth 8 concatenate 9): the leading
                    # pieces are materialised into ONE tensor (element-granular channel copies each way, as for a BatchNormalization
                    # over a concatenation), the rest stays virtual - the reference takes any depth (unet_variants.py:1036-1043)
                    k = len(segs) - _ffi.MAX_SEG + 1
                    head = segs[:k]
                    part = DevTensor(n.name + ":part", N, head[0][0].H, head[0][0].W, sum(c for _t, c in head))
                    self.ops.append(ConcatOp(self, head, part))
                    segs = [(part, part.C)] + segs[k:]
                val[i] = segs
            elif k in ("conv", "convT", "convlstm"):
                if k == "conv" and i in outset and n.attrs["kernel"] == (1, 1):
                    if n.shape[2] > 8:       # a wide head: GEMM-shaped, so the conv goes where the other convs go
                        logits = DevTensor(n.name + ":logits", N, n.shape[0], n.shape[1], n.shape[2])
                        self.ops.append(ConvOp(self, n, k, segs_of(n.inputs[0]), None, None, 0.0, logits, None, (1, 1), 0))
                        op = WideHeadOp(self, n, logits)
                    else:
                        op = HeadOp(self, n, tensor_of(n.inputs[0]))
                    self.ops.append(op)
                    self.heads.append(op)
                    val[i] = op
                    continue
                if k == "conv" and n.attrs["activation"] not in (None, "linear"):
                    raise NotImplementedError(f"{n.name}: activation inside a non-output conv")
                last = n
                chain = [n]
                bn = only_consumer(i, "bn")
                if bn is not None:
                    last = bn
                    chain.append(bn)
                act, alpha = None, 0.0
                an = only_consumer(last.idx, "act") if k != "convlstm" else None
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                    chain.append(an)
                lstm_f = n.attrs["filters"] if k == "convlstm" else 0
                out = DevTensor(last.name, N, last.shape[0], last.shape[1], last.shape[2])
                pooled, psize, pool_node = None, (1, 1), None
                for c in cons[last.idx]:
                    cn = g.nodes[c]
                    if cn.kind == "maxpool" and bn is not None and last.idx not in outset and pooled is None:
                        pool_node, psize = cn, cn.attrs["size"]
                        pooled = DevTensor(cn.name, N, cn.shape[0], cn.shape[1], cn.shape[2])
                        break
                op = ConvOp(self, n, k, segs_of(n.inputs[0]), bn, act, alpha, out, pooled, psize, lstm_f)
                self.ops.append(op)
                for nd in chain:
                    val[nd.idx] = out
                    done.add(nd.idx)
                if pool_node is not None:
                    val[pool_node.idx] = pooled
                    done.add(pool_node.idx)
            elif k == "dense_ae":
                # Flatten -> Dense(features) -> Dense(H*W*C') -> Reshape: the two Dense layers are 1x1 convs over a 1x1 image
                # whose channel axis is the flattened sample (GEMM rows = the batch), on the implicit-GEMM path
                x = self._materialise(val, n.inputs[0])
                H_, W_, Co = n.shape
                nin, F_, nout = x.H * x.W * x.C, int(n.attrs["features"]), H_ * W_ * Co
                flat = DevTensor(n.name + ":flat", N, 1, 1, nin)
                feat = DevTensor(n.name + ":features", N, 1, 1, F_)
                expd = DevTensor(n.name + ":expand", N, 1, 1, nout)
                out = DevTensor(n.name, N, H_, W_, Co)
                self.ops.append(ReshapeOp(self, x, flat, True))
                for src, dst, specs, cin, cout_ in ((flat, feat, n.params[0:2], nin, F_), (feat, expd, n.params[2:4], F_, nout)):
                    k4 = ParamSpec(specs[0].name, (1, 1, cin, cout_), specs[0].init, True, specs[0].fan)
                    fake = Node(-1, "conv", specs[0].name.rsplit("/", 1)[0], [], (1, 1, cout_),
                                dict(kernel=(1, 1), strides=(1, 1), activation=None), [k4, specs[1]])
                    self.ops.append(ConvOp(self, fake, "conv", [(src, cin)], None, None, 0.0, dst, None, (1, 1), 0))
                    slot = self.store.slots[specs[0].name]                      # the parameter keeps its Dense shape (in, out)
                    te, fe = slot.to_engine, slot.from_engine
                    slot.spec = specs[0]
                    slot.to_engine = lambda kern, te=te, c=cin, o=cout_: te(np.asarray(kern).reshape(1, 1, c, o))
                    slot.from_engine = lambda f, fe=fe, c=cin, o=cout_: np.asarray(fe(f)).reshape(c, o)
                self.ops.append(ReshapeOp(self, expd, out, False))
                val[i] = out
            elif k == "bn":
                act, alpha, last = None, 0.0, n
                an = only_consumer(i, "act")
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                x = self._materialise(val, n.inputs[0])
                out = DevTensor(last.name, N, *last.shape)
                self.ops.append(BNOp(self, n, x, act, alpha, out))
                self._fuse_bn_tail(self.ops[-1], n, cons)
                val[i] = val[last.idx] = out
                done.add(last.idx)
            elif k == "add":
                act, alpha, last = None, 0.0, n
                an = only_consumer(i, "act")
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                xs = [tensor_of(j) for j in n.inputs]
                out = DevTensor(last.name, N, *last.shape)
                self.ops.append(AddOp(self, xs, act, alpha, out))
                val[i] = val[last.idx] = out
                done.add(last.idx)
            elif k == "act":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(AddOp(self, [x], n.attrs["fn"], n.attrs["alpha"], out))
                val[i] = out
            elif k == "mul":
                a, r = tensor_of(n.inputs[0]), tensor_of(n.inputs[1])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(MulOp(self, a, r, out))
                val[i] = out
            elif k == "upsample":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(UpsampleOp(self, x, n.attrs["size"], n.attrs["interpolation"] == "bilinear", out))
                val[i] = out
            elif k == "maxpool":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(PoolOp(self, x, n.attrs["size"], out))
                val[i] = out
            else:
                raise NotImplementedError(f"layer kind {k!r} ({n.name}) is not lowered to HIP yet")
        note_ranges()
        for oi in g.outputs:
            if not isinstance(val[oi], HeadOp):
                raise NotImplementedError("model outputs must be 1x1 conv heads")
        self.out_heads = [val[oi] for oi in g.outputs]
        # ---- parameters: ops allocate in creation order, so each op owns one contiguous slice of the flat buffers
        self.store.allocate(self.dtype)
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.finalize()
        self.src_buf = "P16" if self.dtype == "f16" else "P"      # the derived operands are transposed from the fp16 copy Adam refreshes
        self.gathers = [] if _DryRun.on else plan_gathers(self, self.ops)
        self._build_transpose_table()
        self.set_weights({p.name: init_param(p, self.seed) for p in g.param_specs()})
        # ---- static backward schedule (accumulate flags)
        self.producer_of = {id(op.out): op for op in self.ops if isinstance(op, ConvOp)}
        for op in reversed(self.ops):
            op.plan_backward()
        # ONE workspace for every split reduction: forward and data-gradient launches all run on the main stream, in order
        need = max([fwd_workspace_bytes(a) for a in self.fwd_ws_users] + [0])
        self.fwd_ws = torch.empty(need // 4, dtype=torch.float32, device=_dev()) if need and not _DryRun.on else None
        for a in self.fwd_ws_users:
            if a.splitk > 1:
                a.partial = self.fwd_ws.data_ptr() if self.fwd_ws is not None else None
        DevTensor.registry = None
        DevTensor.dtype = torch.float16
        if self.use_side_stream and not _DryRun.on:
            self.side_stream = torch.cuda.Stream()
        self.built = True
        self._init_optimizer_state()
        w0 = getattr(self, "_weights_at_build", None)          # keras_import.load_model: the checkpoint's weights, taken over once
        if w0 is not None and not _DryRun.on:
            self.set_weights(w0)
            self._weights_at_build = None
        return self

    def _fuse_bn_tail(self, bop: "BNOp", n: Node, cons):
        """BN(ReLU(a + b)) - the tail of a MultiResBlock and of every ResPath iteration (unet_variants.py:97-99, :118-120) - as two
        fused passes each way instead of five / seven: the add kernel applies the ReLU and delivers the statistics this BN needs
        (no seg_bn_stats pass), an input that is itself a BatchNormalization with no other consumer (MultiResBlock: BN(cat)) is
        normalised on the fly inside the add (its output is never written), and this BN's backward writes [s > 0] * dx to every
        input of the add (no seg_add_act_bwd pass).  Needs: this BN without an activation of its own, the add's output feeding
        only this BN, every input of the add feeding only the add."""
        g = self.graph
        if not F.FUSE_BN_TAILS or bop.act is not None:
            return
        if bop.x.cs > planning.TAIL_ROW_LIMIT:                # wider than the tail kernels' one column group: the separate passes
            return
        add = next((o for o in reversed(self.ops[:-1]) if isinstance(o, AddOp) and o.out is bop.x), None)
        if add is None or len(add.xs) < 2 or add.fused_bn is not None or ACT[add.act] not in (ACT[None], ACT["relu"]):
            return
        src = g.nodes[n.inputs[0]]                            # the activation node (or the add itself)
        addn = g.nodes[src.inputs[0]] if src.kind == "act" else src
        if addn.kind != "add" or list(cons[src.idx]) != [n.idx] or (src is not addn and list(cons[addn.idx]) != [src.idx]):
            return
        if any(list(cons[j]) != [addn.idx] for j in addn.inputs) or not all(x.needs_grad and x.cs == bop.x.cs for x in add.xs):
            return
        for k, x in enumerate(add.xs):
            b1 = next((o for o in self.ops if isinstance(o, BNOp) and o.out is x), None)
            if b1 is not None and b1.act is None and b1.fused_add is None and b1.folded_into is None:
                add.folded[k] = b1
                b1.folded_into = add
                b1.out.materialised = False                   # (never written: tests / tools must not read it)
                # BN over a concatenation (MultiResBlock's BN(cat[c3, c5, c7]), unet_variants.py:95-96) that only this add reads: the
                # concatenation is never materialised either - the tail kernels gather the pieces (a dense sibling input must carry
                # the add's gradient for the backward scatter: input 0, checked in AddOp.plan_backward)
                co = next((o for o in self.ops if isinstance(o, ConcatOp) and o.out is b1.x), None)
                bn1 = g.nodes[addn.inputs[k]]
                if (F.TAIL_VIRTUAL_CAT and F.TAIL_SHARED_GRAD and co is not None and k > 0 and len(co.segs) <= 4 and b1.C <= planning.TAIL_ROW_LIMIT
                        and list(cons[bn1.inputs[0]]) == [bn1.idx]
                        and any(isinstance(o, ConvOp) and o.out is add.xs[0] and o.has_bn for o in self.ops)):
                    co.virtual = True
                    b1.make_virtual(co.segs)
                    b1.out.buf = torch.empty((0,), dtype=b1.out.buf.dtype, device=b1.out.buf.device)
                continue
            # a Conv_Block (conv -> BN -> ReLU) whose only consumer is this add: normalised on load inside the add (training passes)
            cv = next((o for o in self.ops if isinstance(o, ConvOp) and o.out is x), None)
            if (F.TAIL_LAZY_INPUTS and cv is not None and cv.has_bn and cv.kind == "conv" and cv.pooled is None
                    and ACT[cv.act] in (ACT[None], ACT["relu"])):
                add.lazy[k] = cv
                cv.lazy = True
                cv.out.materialised = False                   # (written by inference passes only)
        bop.fused_add = add
        add.fuse(bop)

    def _build_transpose_table(self):
        """Device table of every derived (transposed) weight operand: refreshed by ONE launch after each optimiser step."""
        jobs = []
        for op in self.ops:
            if isinstance(op, ConvOp):
                jobs += op.transpose_jobs()
        for grp in self.gathers:
            jobs += grp.transpose_jobs()
        self.tr_n, self.tr_tiles, self.tr_table = len(jobs), 0, None
        src16 = 1 if self.src_buf == "P16" else 0
        if not jobs or _DryRun.on:
            return
        arr = (TransposeDesc * len(jobs))()
        for d, job in zip(arr, jobs):
            w, wt, rows, T, K, k0, kcount, rows_pad, dst_ld, dst_coff = job[:10]
            d.w, d.wt, d.rows, d.T, d.K, d.k0, d.kcount, d.rows_pad, d.dst_ld, d.dst_coff = w, wt, rows, T, K, k0, kcount, rows_pad, dst_ld, dst_coff
            d.dst_T, d.dst_t0 = job[10:12] if len(job) > 10 else (0, 0)
            d.tile0, d.src16 = self.tr_tiles, src16
            self.tr_tiles += T * _cdiv(kcount, 32) * _cdiv(rows_pad, 32)
        self.tr_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(_dev())

    def refresh_derived(self):
        """Re-derive the data-gradient operands from the master weights (after an optimiser step / weight import)."""
        if self.tr_table is not None:
            call("seg_transpose_w_multi", _p(self.tr_table), self.tr_n, self.tr_tiles, _stream())

    def _materialise(self, val, i) -> DevTensor:
        v = val[i]
        if isinstance(v, list):
            n = self.graph.nodes[i]
            out = DevTensor(n.name, self.N, *n.shape)
            self.ops.append(ConcatOp(self, v, out))
            val[i] = out
            return out
        return v

    # ---------------------------------------------------------------- weights
    def set_weights(self, weights: Dict[str, np.ndarray]):
        """weights: structural name -> numpy array in Keras layout (see graph.py)."""
        st = self.store
        if self.built:
            self._join_side()
        host_p = st.P.cpu().numpy()
        host_s = st.S.cpu().numpy()
        for name, arr in weights.items():
            s = st.slots.get(name)
            if s is None:
                raise KeyError(f"unknown parameter {name!r}")
            arr = np.asarray(arr, np.float32)
            if tuple(arr.shape) != tuple(s.spec.shape):
                raise ValueError(f"{name}: shape {arr.shape} != {s.spec.shape}")
            flat = np.asarray(s.to_engine(arr), np.float32).reshape(-1)
            assert flat.size == s.numel, (name, flat.size, s.numel)
            (host_p if s.trainable else host_s)[s.off:s.off + s.numel] = flat
        st.P.copy_(torch.from_numpy(host_p))
        st.S.copy_(torch.from_numpy(host_s))
        self._refresh_copies()

    def _join_side(self):
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
        self._param_events = []

    def get_weights(self) -> Dict[str, np.ndarray]:
        st = self.store
        self._join_side()
        host_p, host_s = st.P.cpu().numpy(), st.S.cpu().numpy()
        out = {}
        for name, s in st.slots.items():
            src = host_p if s.trainable else host_s
            out[name] = np.asarray(s.from_engine(src[s.off:s.off + s.numel]), np.float32).reshape(s.spec.shape)
        return out

    def save_weights(self, path: str):
        """By extension: `.keras` - what the reference's ModelCheckpoint writes (Train.py:375) - a Keras-layout archive of this model
        (graph + weights: `segamd.load_model(path)` rebuilds it from the file alone; `save_keras_file(path, like=)` instead fills
        an existing Keras checkpoint); `.safetensors` / `.npz` keyed by structural parameter names, Keras layouts."""
        from .harness import save_weights_file
        save_weights_file(self.get_weights(), path, self.graph)

    def load_weights(self, path: str, config: Optional[str] = None):
        """model.load_weights (Train.py:363,367): a weight file of this package (.npz / .safetensors) or a Keras checkpoint of the
        reference (.keras / .h5 / .hdf5: `load_keras_file`)."""
        from .harness import load_weights_file
        self.set_weights(load_weights_file(path, self.graph, config))

    def load_keras_file(self, path: str, config: Optional[str] = None):
        """Take over the weights of a Keras checkpoint FILE written by the reference for the same constructor call (`.keras` archive,
        legacy whole-model `.h5`; a weights-only `.h5` with `config=` the text of `model.to_json()`): read without TensorFlow or
        h5py (h5lite.py), matched layer by layer through the file's own `inbound_nodes` (keras_file.py, keras_map.py)."""
        from .keras_file import load_keras_file
        self.set_weights(load_keras_file(path, self.graph, config))

    def save_keras_file(self, path: str, like: str, config: Optional[str] = None):
        """This model's weights as a Keras checkpoint shaped like `like` (the checkpoint it was loaded from, or any checkpoint of
        the same constructor call): configuration, metadata and optimiser variables are `like`'s, the layer weights this model's."""
        from .keras_file import save_keras_file
        save_keras_file(self.get_weights(), path, like, self.graph, config)

    def load_keras_model(self, keras_model):
        """Take over the weights of a tf.keras model built by the reference for the same constructor call: every Keras layer
        is matched to its structural name by walking `inbound_nodes` (keras_map.py), never by list position or auto-name."""
        from .keras_map import keras_weights_by_structural_name
        self.set_weights(keras_weights_by_structural_name(keras_model, self.graph))

    def export_to_keras_model(self, keras_model):
        """The other direction: write this model's weights into the matching layers of a tf.keras model."""
        from .keras_map import export_to_keras
        export_to_keras(self.get_weights(), keras_model, self.graph)

    def get_grads(self) -> Dict[str, np.ndarray]:
        """Gradients of the last backward pass, Keras layout (already divided by the internal gradient scale).  Valid
        between backward() and optimizer_step(): the optimiser step clears the buffer for the next backward pass."""
        st = self.store
        host = st.G.cpu().numpy()
        return {name: np.asarray(s.from_engine(host[s.off:s.off + s.numel]), np.float32).reshape(s.spec.shape)
                for name, s in st.slots.items() if s.trainable}

    def nonfinite_updates(self, reset: bool = False) -> int:
        """Parameter elements whose update the optimiser skipped since the last reset because their gradient was NaN or
        +-Inf (an overflow of the fp16 backward pass; Keras would have propagated the NaN into the weights).  Reading it
        synchronises with the device; train_step() uses the asynchronous form (nonfinite_updates_async, one step late) and feeds it to
        loss_scale_feedback(), which halves `grad_target` (the loss scale) when it is non-zero, the way dynamic loss scaling backs off."""
        self._join_side()
        n = int(self.store.nonfinite.cpu())
        if reset and n:
            self.store.nonfinite.zero_()
        return n

    def nonfinite_updates_async(self) -> int:
        """The same counter without stalling the training loop: returns what the PREVIOUS call collected (one step late, 0 the
        first time) and queues, behind the optimiser slices of the step just issued, a copy of the counter into pinned host memory
        followed by its reset.  Nothing joins the side stream, so the optimiser keeps running under the next forward pass; the side
        stream first catches up with the current one (an optimiser that ran there - a graph replay, the single-stream schedule -
        must have finished counting before the counter is copied and cleared).  `nonfinite_drain()` collects what is still pending."""
        if _DryRun.on:
            return 0
        prev = 0
        pend = getattr(self, "_nf_pending", None)
        if pend is not None:
            pend[0].synchronize()                  # (queued a whole step ago: long finished)
            prev = int(pend[1][0])
        if getattr(self, "_nf_host", None) is None:
            self._nf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        cur = torch.cuda.current_stream()
        stream = self.side_stream if self.side_stream is not None else cur
        if stream is not cur:
            stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            self._nf_host.copy_(self.store.nonfinite, non_blocking=True)
            self.store.nonfinite.zero_()
            ev = torch.cuda.Event()
            ev.record(stream)
        self._nf_pending = (ev, self._nf_host)
        return prev

    def nonfinite_drain(self) -> int:
        """Everything the asynchronous counter has not reported yet - the value queued by the last nonfinite_updates_async() plus
        what the optimiser counted since - with one blocking read (end of an epoch / of fit: the logs and the loss-scale back-off
        see every step, also the last one)."""
        if _DryRun.on:
            return 0
        n = 0
        pend = getattr(self, "_nf_pending", None)
        if pend is not None:
            pend[0].synchronize()
            n += int(pend[1][0])
            self._nf_pending = None
        return n + self.nonfinite_updates(reset=True)

    def loss_scale_feedback(self, skipped: int):
        """Dynamic loss scaling: `skipped` = parameter elements whose update an optimiser step skipped because their gradient was
        NaN / Inf (an overflow of the fp16-range gradient storage).  Non-zero: the target halves (down to 2^-20: a network whose
        backward pass amplifies by more than 65 504 - BatchNormalization after BatchNormalization on a near-constant signal - still
        finds a scale at which nothing overflows, at the price of its smallest gradients); after
        `loss_scale_growth_interval` clean steps it doubles again, up to the mode's default.  train_step() feeds it from the
        asynchronous counter (one step late, no synchronisation); harness.fit adds what the end of an epoch drains and logs the
        total (`take_skipped`)."""
        if skipped:
            self._ls_skipped += int(skipped)
            self._ls_clean = 0
            self.grad_target = max(self.grad_target / 2.0, 2.0 ** -20)      # (a power of two: the device picks scale = target / max |dLoss/dOut| as one)
        else:
            self._ls_clean += 1
            if self._ls_clean >= self.loss_scale_growth_interval and self.grad_target < self.grad_target_max:
                self.grad_target = min(self.grad_target * 2.0, self.grad_target_max)
                self._ls_clean = 0

    def take_skipped(self) -> int:
        """Skipped parameter-element updates reported to loss_scale_feedback since the last call (for the epoch log)."""
        n, self._ls_skipped = self._ls_skipped, 0
        return n

    def check_input_range(self):
        """Raises when an input element handed to forward() since the last check was NaN, +-Inf or |x| >= 65504: the reference's fp32
        tensors hold such values, this engine's fp16 / split-pair elements do not - they would turn into Inf / NaN downstream.
        (Reads a device counter: synchronises.  predict() and harness.fit call it where they synchronise anyway.)"""
        if _DryRun.on or not self.built:
            return
        n = int(self.input_overflow.cpu())
        if n:
            self.input_overflow.zero_()
            raise ValueError(f"{n} input element(s) are NaN / Inf or outside |x| < 65504, the range of this engine's fp16 / split-pair storage "
                             "(docs/experiments.md section 2); normalise the input (the reference's loaders divide images by 255)")

    def _refresh_copies(self):
        st = self.store
        self._enter()
        if self.dtype == "f16":
            call("seg_cast_f32_f16", _p(st.P), _p(st.P16), st.P.numel(), _stream())
        else:
            call("seg_cast_f32_split", _p(st.P), _p(st.PS), st.P.numel(), _stream())
        for op in self.ops:
            op.post_step()
        self.refresh_derived()

    # ---------------------------------------------------------------- execution
    def _check_x(self, x):
        g = self.graph
        H, W, Cin = g.nodes[g.inputs[0]].shape
        if g.ndim == 1 and x.dim() == 3:
            x = x.unsqueeze(1)
        if not self.built:
            self.build(x.shape[0])
        if tuple(x.shape) != (self.N, H, W, Cin):
            raise ValueError(f"input shape {tuple(x.shape)} != {(self.N, H, W, Cin)} (NHWC; the batch size is fixed at build)")
        return x.to(device=_dev(), dtype=torch.float32).contiguous()

    def forward(self, x: torch.Tensor, training: Optional[bool] = None) -> List[torch.Tensor]:
        """x: (N,H,W,C) [2D] or (N,L,C) [1D] float32.  Returns [out, level1, ...] fp32 NHWC (views of internal buffers)."""
        training = self.training if training is None else training
        self._x = self._check_x(x)
        self._enter()
        # optimiser slices still being updated on the side stream: (need, event) in the order the side stream processes them, `need` =
        # the first op that READS a parameter of the slice - the slice's own first op, or the Conv_Block that carries a fused output
        # head whose weights live in the slice (_slice_need).  The side stream is in order, so the last event an op needs implies
        # the earlier ones
        ev = self._param_events
        main = None if _DryRun.on else torch.cuda.current_stream()
        plan = self._wait_plan(tuple(e[0] for e in ev)) if ev else {}
        for i, op in enumerate(self.ops):
            j = plan.get(i)
            if j is not None:
                main.wait_event(ev[j][1])
            op.forward(training)
        if ev and plan.get(-1) is not None:
            main.wait_event(ev[-1][1])
        self._param_events = []
        outs = [h.y for h in self.out_heads]
        return [o.squeeze(1) for o in outs] if self.graph.ndim == 1 else outs

    __call__ = forward

    def predict(self, x, batch_size=None, verbose=0):
        """model.predict (2DCNN/Test.py:153,161): inference-mode BN.  Returns numpy, a list when the model has several outputs.
        Any number of samples: the engine's batch is fixed at build, so the input is processed in chunks of that size, the
        last one padded with zeros (inference-mode BN keeps samples independent, the padding rows are dropped)."""
        x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x
        if not self.built:
            self.build(x.shape[0])
        n, N = x.shape[0], self.N
        if n == N:
            outs = [o.cpu().numpy() for o in self.forward(x, training=False)]
        else:
            parts = []
            for lo in range(0, n, N):
                xb = x[lo:lo + N]
                k = xb.shape[0]
                if k < N:
                    xb = torch.cat([xb, torch.zeros((N - k,) + tuple(xb.shape[1:]), dtype=xb.dtype, device=xb.device)], 0)
                parts.append([o[:k].cpu().numpy() for o in self.forward(xb, training=False)])
            outs = [np.concatenate([p[i] for p in parts], 0) for i in range(len(parts[0]))]
        self.check_input_range()
        return outs if len(outs) > 1 else outs[0]

    def adam_slices(self) -> List[Tuple[int, int, int]]:
        """[(lo, hi, first_op)]: op-aligned slices of the flat parameter buffers in forward order that tile [0, numel);
        an op's parameters are final once the slices with first_op <= its index have been updated."""
        if self._adam_slices is None:
            owners = sorted(self._op_grad_range.items())
            out, lo, first = [], None, None
            for oi, (a, b) in owners:
                if lo is None:
                    lo, first = a, oi
                # the first slices are what the next forward pass waits for at the step boundary (the main stream idles
                # until the first layer's weights are updated): they are kept short - first layer alone, then 256 K elements
                limit = F.ADAM_SLICE_ELEMS if (len(out) >= 2 or not F.ADAM_URGENT_FIRST) else (1 if not out else 1 << 18)
                if b - lo >= limit:
                    out.append([lo, b, first])
                    lo = None
            if lo is not None:
                out.append([lo, owners[-1][1][1], first])
            if not out:
                out = [[0, self.store.P.numel(), 0]]
            out[0][0], out[-1][1] = 0, self.store.P.numel()
            self._adam_slices = [tuple(o) for o in out]
        return self._adam_slices

    def _wait_plan(self, needs: Tuple[int, ...]) -> Dict[int, int]:
        """{op index: position of the event to wait for in front of that op} for optimiser events with these `need` indices (in
        side-stream order; the same every step, hence cached); key -1: events nobody needed are awaited at the end of the pass."""
        cache = getattr(self, "_wait_plan_cache", None)
        if cache is None or cache[0] != needs:
            plan, k = {}, 0
            for i in range(len(self.ops)):
                j = -1
                for q in range(k, len(needs)):
                    if needs[q] <= i:
                        j = q
                if j >= 0:
                    plan[i] = j
                    k = j + 1
            if k < len(needs):
                plan[-1] = len(needs) - 1
            self._wait_plan_cache = cache = (needs, plan)
        return cache[1]

    def _slice_need(self, lo: int, hi: int, first_op: int) -> int:
        """The first op of the forward pass that reads a parameter in [lo, hi) of the flat buffers: the slice's own first op, or -
        earlier - a Conv_Block whose bn_act_apply pass carries the forward of an output head (ConvOp.heads) whose kernel / bias sit
        in the slice.  The next forward pass may start an op only after the optimiser has updated every slice it needs."""
        if getattr(self, "_head_needs", None) is None:
            self._head_needs = []
            for i, op in enumerate(self.ops):
                for h in getattr(op, "heads", ()):
                    for nm in (h.wname, h.bname):
                        self._head_needs.append((self.store.slots[nm].off, i))
        return min([first_op] + [i for off, i in self._head_needs if lo <= off < hi])

    def grad_buckets(self) -> List[Tuple[int, int, int]]:
        """[(lo, hi, op_index)]: contiguous slices of the flat gradient buffer, each complete once the backward sweep
        (which walks the ops in reverse) has finished op `op_index`.  Slices follow op boundaries, are at least
        bucket_elems long (except the last) and tile [0, n_train) exactly."""
        owners = sorted(self._op_grad_range.items())          # ascending op index == ascending offsets
        out, lo, first = [], None, None
        for oi, (a, b) in owners:
            if lo is None:
                lo, first = a, oi
            # the first slice (the encoder's first layers) is what the next forward pass needs first and what the backward
            # sweep finishes last: with more than one rank it is kept small so that its exchange is short (F.DP_ASYNC)
            limit = min(self.bucket_elems, F.DP_URGENT_ELEMS) if (not out and F.DP_ASYNC and self.world_size > 1) else self.bucket_elems
            if b - lo >= limit:
                out.append((lo, b, first))
                lo = None
        if lo is not None:
            out.append((lo, owners[-1][1][1], first))
        return out

    def bucket_holds(self) -> List[bool]:
        """For every slice of grad_buckets(): hold its all-reduce back until the backward sweep has ended?  A collective that
        cannot finish before the sweep does would still be on the (in-order) RCCL stream when the more urgent slices - the
        encoder's, needed first by the next forward pass, finished last by this sweep - become ready, and delay them.  Held
        slices are issued after the sweep in forward order.  Estimate: ring all-reduce time 2 (w-1)/w x bytes / bus bandwidth
        (measured once by dp.DataParallel on the job's own fabric - `dp_busbw_gbs` -, else SEG_DP_BUSBW_GBS, default 300: xGMI rings are per-link bound), collectives queued one after the other, against the
        remaining GEMM work of the sweep at 600 TFLOP/s (fwd-shaped data gradient + weight gradient = 4 x MACs)."""
        bk = self.grad_buckets()
        w = self.world_size
        if w <= 1 or not F.DP_ASYNC:
            return [False] * len(bk)
        cost = [4.0 * op.M * op.T * op.Kpad * op.n_rows / 600e12 if isinstance(op, ConvOp) else 0.0 for op in self.ops]
        remaining = [0.0]
        for c in cost:
            remaining.append(remaining[-1] + c)                  # remaining[oi] = cost of the ops the sweep still has to run after op oi
        holds, busy_until = [False] * len(bk), -1e30             # times relative to the end of the sweep (negative = before it)
        for j in range(len(bk) - 1, -1, -1):                     # the order in which the sweep completes the slices
            lo, hi, oi = bk[j]
            start = max(-remaining[oi], busy_until)              # collectives run one after the other
            end = start + 2.0 * (w - 1) / w * (hi - lo) * 4 / ((self.dp_busbw_gbs or F.DP_BUSBW_GBS) * 1e9)
            if end > 0.0:
                holds[j] = True
            else:
                busy_until = end
        return holds

    def backward(self, grads: Optional[Sequence[torch.Tensor]] = None):
        """Backward pass.  grads[i] = dLoss/d output_i (fp32, same shape as the output); None: use the gradients the
        loss kernels left in the heads' dy buffers."""
        if self.side_stream is not None and not self._graph_mode:         # (a captured step joins the side stream at its end)
            torch.cuda.current_stream().wait_stream(self.side_stream)      # derived operands rebuilt after the last optimiser step
        s = _stream()
        self._enter()
        if grads is not None:
            for h, gt in zip(self.out_heads, grads):
                h.dy.copy_(gt.reshape(h.dy.shape))
        self.amax.zero_()
        for h in self.out_heads:
            call("seg_absmax", _p(h.dy), h.dy.numel(), _p(self.amax), s)
        call("seg_grad_scale", _p(self.amax), float(self.grad_target), _p(self.gs), s)
        if not self._g_clean:
            self.store.G.zero_()          # (after an optimiser step the side stream has already cleared it)
        self._g_clean = False
        ready = {}
        if self.bucket_hook is not None:
            for (lo, hi, oi), hold in zip(self.grad_buckets(), self.bucket_holds()):
                ready.setdefault(oi, []).append((lo, hi, hold))

        def exchange(lo, hi):
            if self.side_stream is not None:
                # the slice holds weight gradients (side stream) and BN / bias gradients (this stream): issue the
                # exchange from the side stream after it has caught up with this one - this stream never stalls
                self.side_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side_stream):
                    self.bucket_hook(lo, hi)
            else:
                self.bucket_hook(lo, hi)

        held = []
        for oi in range(len(self.ops) - 1, -1, -1):
            self.ops[oi].backward()
            for lo, hi, hold in ready.get(oi, ()):
                if hold:
                    held.append((lo, hi))          # see bucket_holds(): issued after the sweep, most urgent first
                else:
                    exchange(lo, hi)
        for lo, hi in sorted(held):
            exchange(lo, hi)
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)

    def compute_loss(self, targets: Sequence[torch.Tensor], loss: Optional[str] = None) -> torch.Tensor:
        """Sum over outputs of weight x mean-reduced loss (unit weights unless compile(loss_weights=...), 2DCNN/Train.py:322);
        fills the heads' dy buffers.  'sparse_cce' takes one class index per pixel as its target."""
        kind = LOSS[loss or self.loss_name]
        weights = self.loss_weights or [1.0] * len(self.out_heads)
        self.loss_buf.zero_()
        s = _stream()
        for h, t, w in zip(self.out_heads, targets, weights):
            t = t.to(device=_dev(), dtype=torch.float32).contiguous()
            npix = h.y.numel() // h.cout
            if t.numel() != (npix if kind == LOSS["sparse_cce"] else h.y.numel()):
                raise ValueError(f"target for {h.node.name}: {tuple(t.shape)} vs output {tuple(h.y.shape)}")
            flag = None
            if kind in (LOSS["hinge"], LOSS["squared_hinge"]):       # Keras reads {0,1} labels as {-1,+1} iff ALL of them are 0/1
                if self._label_flag is None:
                    self._label_flag = torch.ones(1, dtype=torch.int32, device=_dev())
                flag = self._label_flag
                call("seg_labels_binary", _p(t), t.numel(), _p(flag), s)
            call("seg_loss_fwd_bwd_w", kind, _p(h.y), _p(t), npix, h.cout, float(w), _p(flag) if flag is not None else None,
                 _p(self.loss_buf), _p(h.dy), s)
        return self.loss_buf

    def _init_optimizer_state(self):
        """The slot variables as Keras creates them: zeros, except the accumulators of Adagrad and FTRL (0.1)."""
        st = self.store
        self._join_side()
        st.M.zero_()
        st.V.zero_()
        if self.optimizer == "Adagrad":
            st.V.fill_(0.1)
        elif self.optimizer == "FTRL":
            st.M.fill_(0.1)
        self._u_product = 1.0

    def _optimizer_scalars(self):
        """The five per-step host scalars of seg_optimizer_step (include/seg_hip.h) for step self.adam_t."""
        t, lr, eps = max(self.adam_t, 1), self.lr, 1e-7
        o = self.optimizer
        if o == "RMSprop":
            c = [lr, 0.9, eps]
        elif o == "Adadelta":
            c = [lr, 0.95, eps]
        elif o == "Adamax":
            c = [lr / (1.0 - 0.9 ** t), 0.9, 0.999, eps]
        elif o == "Nadam":
            u_t = 0.9 * (1.0 - 0.5 * 0.96 ** t)
            u_t1 = 0.9 * (1.0 - 0.5 * 0.96 ** (t + 1))
            prod_t = self._u_product * u_t
            self._u_product = prod_t
            c = [lr, u_t1 / (1.0 - prod_t * u_t1), (1.0 - u_t) / (1.0 - prod_t), 1.0 / (1.0 - 0.999 ** t), eps]
        else:                           # SGD, Adagrad, FTRL
            c = [lr, 0.0, eps]
        return (C.c_float * 5)(*(c + [0.0] * (5 - len(c))))

    def optimizer_step(self):
        st = self.store
        self._enter()
        if not self._graph_mode:          # (a captured step gets its step count / size from graphed_train_step)
            self.adam_t += 1
        a_t = self.lr * math.sqrt(1.0 - self.beta_2 ** max(self.adam_t, 1)) / (1.0 - self.beta_1 ** max(self.adam_t, 1))
        other = None
        if self.optimizer != "Adam":
            if self._graph_mode:
                raise NotImplementedError("a captured training step supports the Adam optimiser only")
            other = (OPTIMIZER[self.optimizer], self._optimizer_scalars())

        def adam(lo, hi, exchanged=False):
            off = lambda t: C.c_void_p(t.data_ptr() + lo * t.element_size())
            p16, pfmt = (off(st.P16), _ffi.OPERAND_F16) if self.dtype == "f16" else (off(st.PS), _ffi.OPERAND_SPLIT)
            # a slice that went through the bf16 exchange is read from there (the sum over the ranks), everything else from G
            if exchanged and self.grad_dtype == "bf16":
                g, gdt = off(st.G16), _ffi.GRAD_BF16
            else:
                g, gdt = off(st.G), _ffi.GRAD_F32
            if other is not None:        # the rest of the registry: same slices, same streams, another update rule
                call("seg_optimizer_step_ex", other[0], off(st.P), g, gdt, off(st.M), off(st.V), p16, pfmt, hi - lo, other[1], 1.0 / self.world_size,
                     _p(st.nonfinite), _stream())
            else:                        # (a captured step reads its step size from device memory, refreshed before every replay)
                dev = self._graph_mode
                call("seg_adam_step_ex", off(st.P), g, gdt, off(st.M), off(st.V), p16, pfmt, hi - lo, 1.0 if dev else a_t,
                     _p(self._alpha_dev) if dev else None, self.beta_1, self.beta_2, self.epsilon, 1.0 / self.world_size, _p(st.nonfinite), _stream())

        n = st.P.numel()
        if self.bucket_waits is not None and F.DP_ASYNC and self.side_stream is not None and not self._graph_mode:
            # overlapped data parallel, cross-iteration: the waits for the exchanged slices and their optimiser updates go to
            # the side stream, in forward order, each followed by an event; the next forward pass starts at once and every op
            # waits only for the slice holding its own parameters.  The encoder's small first slice is exchanged first after
            # the sweep (bucket_holds), so the big bottleneck slices travel under the encoder's forward pass.
            first_of = {lo: oi for lo, _hi, oi in self.grad_buckets()}
            side = self.side_stream
            side.wait_stream(torch.cuda.current_stream())
            events, pos = [], 0
            with torch.cuda.stream(side):
                for lo, hi, work in sorted(self.bucket_waits(), key=lambda t: t[0]):
                    self._timed_wait(work, side)   # stream-side for RCCL: the side stream waits for this collective
                    if lo > pos:
                        adam(pos, lo)
                    adam(lo, hi, True)
                    pos = hi
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((self._slice_need(lo, hi, first_of.get(lo, 0)), e))
                if pos < n:                        # alignment padding behind the last parameter
                    adam(pos, n)
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((len(self.ops), e))
            self._param_events = events
        elif self.bucket_waits is not None:
            # overlapped data parallel: update each slice of the flat buffers as soon as ITS all-reduce has finished (the
            # waits are stream-side), so the optimiser of the decoder's slices runs under the exchange of the encoder's
            done = []
            for lo, hi, work in self.bucket_waits():
                self._timed_wait(work, None)
                adam(lo, hi, True)
                done.append((lo, hi))
            done.sort()
            pos = 0
            for lo, hi in done + [(n, n)]:            # whatever no bucket covered (alignment padding behind the last parameter)
                if lo > pos:
                    adam(pos, lo)
                pos = max(pos, hi)
        elif self.allreduce is None and self.side_stream is not None and F.ADAM_ASYNC and not self._graph_mode:
            # single GPU: the update itself moves under the next forward pass - slices in forward order on the side stream,
            # an event per slice; forward() makes each op wait only for the slice holding its own parameters (Adam
            # streams 30 B per parameter from HBM while the forward GEMMs leave the HBM mostly idle)
            side = self.side_stream
            side.wait_stream(torch.cuda.current_stream())
            events = []
            # (slices in the order the next forward pass needs them: a slice holding the weights of a fused output head is needed by
            # the Conv_Block that carries the head - level d of UNet++ rides on the FIRST encoder block)
            order = sorted(self.adam_slices(), key=lambda t: (self._slice_need(*t), t[0]))
            with torch.cuda.stream(side):
                for lo, hi, first_op in order:
                    adam(lo, hi)
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((self._slice_need(lo, hi, first_op), e))
            self._param_events = events
        else:
            if self.allreduce is not None:
                self.allreduce(st.G)
            adam(0, n)
        for op in self.ops:
            op.post_step()
        if self.side_stream is None:
            self.refresh_derived()
        else:
            # the transposed operands are first needed by the next backward pass: rebuild them on the side stream, under
            # the next forward pass (backward() joins the side stream before its first kernel)
            self.side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side_stream):
                self.store.G.zero_()      # likewise the gradient accumulators of the next backward pass
                self._g_clean = True
                self.refresh_derived()
            if self._graph_mode:          # a captured step has to end with every forked stream joined
                torch.cuda.current_stream().wait_stream(self.side_stream)

    def _timed_wait(self, work, stream):
        """Wait for one gradient slice's collective.  With `comm_timing` set (bench.py --gpus N) the wait is bracketed by two
        events on the waiting stream: their distance is the time that stream sat idle for the exchange, i.e. communication
        the backward sweep / the optimiser of earlier slices did not cover."""
        if not getattr(self, "comm_timing", False) or _DryRun.on:
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = stream if stream is not None else torch.cuda.current_stream()
        e0.record(stream)
        work.wait()
        e1.record(stream)
        self._comm_events.append((e0, e1))

    def exposed_comm_ms(self, reset=True) -> float:
        """Sum of the timed waits since the last call (see _timed_wait); synchronises."""
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self._comm_events)
        if reset:
            self._comm_events = []
        return ms

    def train_step(self, x, targets, loss: Optional[str] = None) -> torch.Tensor:
        """One optimiser step = Keras' default train_step (Train.py:394-415): forward (training BN) -> loss -> backward
        -> [gradient all-reduce] -> Adam.  Returns the (device) scalar loss; nothing synchronises with the host."""
        self.forward(x, training=True)
        if not isinstance(targets, (list, tuple)):
            targets = [targets]
        lv = self.compute_loss(targets, loss)
        self.backward(None)
        self.optimizer_step()
        if self.auto_loss_scale and not self._graph_mode:
            self.loss_scale_feedback(self.nonfinite_updates_async())
        return lv

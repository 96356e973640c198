"""Switches and measured constants of the engine (model.py / ops.py / planning.py read them as `F.<NAME>` at call time, so a test or an
A/B run can flip one: `monkeypatch.setattr(flags, "FUSE_BN_TAILS", False)`).  Environment variables are kept only where an
alternative path is needed for bisecting or for same-box A/Bs; tuning numbers whose sweeps are recorded in docs/experiments.md are plain
constants (round 4 removed SEG_BN_PARTS_BWD, SEG_WGRAD_OVERWRITE, SEG_BIAS_FROM_WGRAD, SEG_ADAM_SLICE_ELEMS, SEG_ADAM_URGENT_FIRST,
SEG_DP_URGENT_ELEMS, SEG_FWD_HALO_MIN_BLOCKS(_F32), SEG_FWD_SPLITK_{MAX_BLOCKS,TARGET,MIN_K}, SEG_HEAD_DW_SIDE, SEG_WGRAD_BLOCKS,
SEG_HALO_BLOCKS_TC and, in the library, SEG_FWD_PIPE, SEG_FWD4_TG2 / TG3, SEG_WGRAD_WS / TS / V1, SEG_IGEMM_V1, SEG_ADAM_BLOCKS,
SEG_NO_DEEP_RING, SEG_DEEP_RING_MAX)."""
import os

BN_PARTS_BWD = 512     # partial rows of the BN-backward reductions: 2 blocks per CU (same-box A/B against 1024: C4 47.4 -> 46.8 ms, C3 -0.1, C2 -0.06; 256 and 2048 lose)
GRAD_TARGET = 1024.0      # throughput mode (fp16 gradients): max |dLoss/dOut| is rescaled to [512, 1024) before the backward pass
GRAD_TARGET_F32 = 32.0    # parity mode: [16, 32).  The split pair keeps ~22 bits down to 2^-14 and an absolute floor of 2^-36 below, so
                          # the room is better spent ABOVE: a first layer whose input has a small variance (a one-channel 1-D signal
                          # through a 3-tap conv: 1 / sigma = 10 .. 25) amplifies its gradient past 65 504 at 1024 - 35 of 600 random
                          # builder configurations overflowed in their first step, 2 at 32 (tools/fuzz_models.py) - while every
                          # gradient figure of the full-size suite is unchanged (C1 .. C5 to 3 digits, gpurun_out/fullsize_report*.json)
LOSS_SCALE_GROWTH_INTERVAL = 200     # clean optimiser steps after which a backed-off loss scale doubles again (dynamic loss scaling)
WGRAD_VARIANT = 0         # 0: ds_read_b64_tr_b16 operand reads, 1: transposing LDS stores
WGRAD_OVERWRITE = True   # single-writer weight-gradient tiles store into the cleared buffer
WGRAD_HALO = os.environ.get("SEG_WGRAD_HALO", "1") != "0"      # variant 2 (LDS-DMA halo tiles) for the 3x3 stride-1 layers
WGRAD_TC2_F32 = os.environ.get("SEG_WGRAD_TC2", "1") != "0"   # parity mode: class-pair kernel for the transposed convs (the library reads the same switch)
WGRAD_DETERMINISTIC = os.environ.get("SEG_WGRAD_DETERMINISTIC", "1") != "0"   # pixel-split weight gradients through per-layer partial tiles + an ordered reduce (no float atomics)
FUSE_ACT_BWD = os.environ.get("SEG_FUSE_ACT_BWD", "1") != "0"
FUSE_BN_TAILS = os.environ.get("SEG_FUSE_BN_TAILS", "1") != "0"             # BN(ReLU(a + b)) tails of MultiResBlock / ResPath as fused passes
FUSE_INFERENCE_BN = os.environ.get("SEG_FUSE_INFERENCE_BN", "1") != "0"     # predict: conv -> BN(moving statistics) -> activation as ONE launch
# the fused tails, round 4 (each with its own switch for bisecting / same-box A/Bs):
TAIL_LAZY_INPUTS = os.environ.get("SEG_TAIL_LAZY_INPUTS", "1") != "0"       # a Conv_Block feeding only the add is normalised on load there (no bn_act_apply pass)
TAIL_SHARED_GRAD = os.environ.get("SEG_TAIL_SHARED_GRAD", "1") != "0"       # the add's gradient is written once; the Conv_Blocks behind it read the one buffer
TAIL_VIRTUAL_CAT = os.environ.get("SEG_TAIL_VIRTUAL_CAT", "1") != "0"       # MultiResBlock's cat[c3, c5, c7] exists only in the tail kernels' LDS tiles
TAIL_LAZY_SUMS = os.environ.get("SEG_TAIL_LAZY_SUMS", "1") != "0"           # the BN-backward sums of those Conv_Blocks ride on the tail's fan-out launch (no reduce pass)
TAIL_PIECE_STATS = os.environ.get("SEG_TAIL_PIECE_STATS", "1") != "0"       # the pieces of a virtual concatenation hand BN(cat) their sums from their own apply pass
TAIL_PAIR_APPLY = os.environ.get("SEG_TAIL_PAIR_APPLY", "1") != "0"         # ... and the two blocks of a ResPath tail get their dz from ONE pass over the shared dOut
GATHER_DGRAD = os.environ.get("SEG_GATHER_DGRAD", "1") != "0"
GATHER_DGRAD_1X1 = os.environ.get("SEG_GATHER_DGRAD_1X1", "1") != "0"       # the 1x1 shortcut conv joins the gathered data gradient of its 3x3 sibling
GATHER_1X1_MAX_C = 64         # ... for layers of at most this many channels (HBM-bound ones)
BIAS_FROM_WGRAD = True
ADAM_ASYNC = os.environ.get("SEG_ADAM_ASYNC", "1") != "0"
DP_ASYNC = os.environ.get("SEG_DP_ASYNC", "1") != "0"           # data parallel: exchanged slices are awaited + updated on the side stream under the next forward
DP_URGENT_ELEMS = 4 << 20     # size at which the first (most urgent) gradient slice closes
DP_BUSBW_GBS = float(os.environ.get("SEG_DP_BUSBW_GBS", "300"))                 # all-reduce bus bandwidth the hold-back estimate assumes ...
DP_BUSBW_MEASURE = os.environ.get("SEG_DP_BUSBW_GBS") is None                   # ... unless dp.DataParallel measured the fabric (one timed all-reduce at construction, the minimum over the ranks)
ADAM_SLICE_ELEMS = 4 << 20
ADAM_URGENT_FIRST = True
FWD_HALO = os.environ.get("SEG_FWD_HALO", "1") != "0"       # variant 3 (halo tiles) for the 3x3 stride-1 layers
FWD_HALO_MIN_BLOCKS = 384
FWD_HALO_MIN_BLOCKS_F32 = 128
FWD_SPLITK = os.environ.get("SEG_FWD_SPLITK", "1") != "0"              # split reduction of under-filled forward / data-gradient grids
FWD_SPLITK_MAX_BLOCKS = 256     # ... grids of at most one block per CU
FWD_SPLITK_TARGET = 512             # ... are split towards two blocks per CU
FWD_SPLITK_MIN_K = 3072              # ... while a slice keeps this much of the reduction
FWD_HALO_NARROW = os.environ.get("SEG_FWD_HALO_NARROW", "1") != "0"    # parity mode: the 256 x 32 halo tile for layers of <= 32 output channels
WGRAD_FORK_EARLY = os.environ.get("SEG_WGRAD_FORK_EARLY", "1") != "0"   # side stream forks when dz is final, before the op's data gradients are enqueued
HEAD_DW_SIDE = True          # output heads: weight / bias gradient on the side stream
HEAD_FUSE_FWD = os.environ.get("SEG_HEAD_FUSE_FWD", "1") != "0"        # a one-channel head's forward rides on the bn_act_apply pass of the Conv_Block it reads
HEAD_INJECT = os.environ.get("SEG_HEAD_INJECT", "1") != "0"            # ... and its data gradient is added on the fly in that block's BN-backward passes
WGRAD_BLOCKS = 1536       # block target of the split-K of the per-tap weight-gradient kernels
HALO_BLOCKS_TC = 256   # the same for the transposed-conv class-pair kernel
HALO_BLOCKS = int(os.environ.get("SEG_HALO_BLOCKS", "256"))          # blocks per round of the variant-2 kernel: ONE per CU - it shares the GPU with the main stream (measured: 512 = 2 per CU is faster alone, slower in the step)
# parity mode, the BACKWARD GEMMs only (SegModel.build(backward=)): how many fp16 MFMAs a product costs there.  The forward
# pass - outputs, masks, `predict` - is the same bits in every mode.  "<d|b><D>w<W>": D = 3 | 2 MFMAs per product in the data gradients
# (d2: the incoming gradient dz read as its fp16 hi half; b2: the transposed WEIGHTS read as theirs - O(1) numbers in fp16's normal
# range, where dz spans everything below the loss scale), W = 3 | 2 | 1 in the weight gradients (2: the layer input as its hi half; 1: dz as
# well).  "3mfma" = d3w3: every product a b ~= ah bh + (al bh + ah bl) / 2048 as in the forward pass; "2mfma" = b2w2.
# "auto" (the default): b2w2 when the model's longest chain of convolutions (Graph.longest_conv_chain) is at most BACKWARD_AUTO_CHAIN
# layers AND the graph is a plain Conv_Block stack (Graph.plain_conv_bn_stack: no attention gate, every BatchNormalization directly
# behind a convolution), else d3w2.  A weight gradient's rounding lands once, in dW, and leaves dz - hence every bias / gamma / beta
# gradient - exact; a data gradient's is inherited by every layer below it, grows with the chain, and is amplified where a
# normalisation follows a normalisation or a gate multiplies.  Measured, relative L2 per parameter tensor against the oracle with the
# decisions imposed (tolerance 1e-3): at the BASELINE geometries (tests/test_gpu_fullsize.py) b2w2 6.3e-4 / 4.8e-4 / 6.8e-4 on C2 / C3 /
# C5 (chains of 19 - 20, plain stacks), 1.5e-3 on C4 (52, MultiRes tails + gates); d3w2 3.8e-4 / 3.1e-4 / 3.8e-4 and 2.8e-4 on C4;
# d3w3 7e-6 / 3e-6 / 7e-6 and 9e-5.  Over random builder calls at 64 - 256 pixels (tools/fuzz_models.py --big, seed 62): with b2w2 for
# every chain <= 32, 10 of 133 configurations exceed 1e-3 - eight with attention gates (the gate's one-channel transposed-conv bias,
# a scalar sum of cancelling terms: 1.0e-3 .. 3.1e-3), two MultiResUNets (BatchNormalization gammas 3.7e-3, 2.9e-2); the rule above
# sends exactly those to d3w2 (profiles/r06_fuzz_campaigns.txt).
BACKWARD = os.environ.get("SEG_BACKWARD", "auto")
BACKWARD_AUTO_CHAIN = 32
# b2 modes: also the data gradients on the per-tap LDS-DMA kernel (deep levels, 1x1 convs)?  Measured (round 6, same box): C3 41.64 -> 41.50 ms,
# worst parameter gradient C2 6.3e-4 -> 7.9e-4, C5 6.8e-4 -> 8.4e-4 (the 8x8 / 16x16 levels are where the chain is longest) - 0.3 % of the
# step for a quarter of the margin to 1e-3: off.
BWD_B2_PER_TAP = os.environ.get("SEG_BWD_B2_PER_TAP", "0") != "0"

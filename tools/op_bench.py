#!/usr/bin/env python3
"""Times single conv layers of a bench workload (forward / data-gradient / weight-gradient launches in isolation),
optionally sweeping the tile and the K-step of the LDS-DMA kernel.  Tuning aid for csrc/igemm.hip:

    python tools/op_bench.py --ops X05/tconv,X05/conv --phases fwd,dgrad,wgrad --tiles 128x128,256x64,128x64 --bk 32,64
"""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

PKG = bench.PKG


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--ops", default="")
    ap.add_argument("--phases", default="fwd,dgrad,wgrad")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--bk", default="")
    ap.add_argument("--splitk", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f16", choices=("f16", "f32"), help="f32 = the parity mode's split-fp16 kernels")
    ap.add_argument("--dbg", default="", help="comma list of timing-experiment bit masks for the pipelined halo kernel (fwd / dgrad phases; results are wrong)")
    args = ap.parse_args()
    pkg = importlib.import_module(PKG)
    M = importlib.import_module(PKG + ".model")
    kw, H, W, Cin, bs, loss, _ = bench.WORKLOADS[args.workload]
    model = pkg.unet_model_builder(length=H, width=W, num_channels=Cin, train_mode="from_scratch", **kw).ResNet50()
    model.build(bs, dtype=args.dtype)
    model.compile(loss=loss, lr=2e-4)
    x, tg = bench.make_batch(model, bs, H, W, Cin, loss, 0)
    model.train_step(x, tg)
    model.train_step(x, tg)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    convs = {op.node.name: op for op in model.ops if isinstance(op, M.ConvOp)}
    names = [n for n in args.ops.split(",") if n] or list(convs)
    tiles = [tuple(int(v) for v in t.split("x")) for t in args.tiles.split(",") if t] or [None]
    bks = [b for b in args.bk.split(",") if b] or [None]
    sks = [int(v) for v in args.splitk.split(",") if v] or [None]

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3

    for name in names:
        op = convs[name]
        for ph in args.phases.split(","):
            if ph == "fwd":
                a, fl = op.fwd_args, op.flops_fwd
                fn = lambda: M.call("seg_igemm_fwd", C.byref(a), s)
            elif ph == "dgrad":
                if not getattr(op, "dg_args", None):
                    continue
                a, _z, fl = op.dg_args[0]
                fn = lambda: M.call("seg_igemm_fwd", C.byref(a), s)
            else:
                a, fl = op.wg_args, op.flops_fwd
                fn = lambda: M.call("seg_igemm_wgrad", C.byref(a), s)
            t0 = (a.tile_m, a.tile_n)
            sk0 = a.splitk if ph == "wgrad" else None
            part0 = a.partial if ph == "wgrad" else None
            if args.dbg and (a.variant & 255) == (2 if ph == "wgrad" else 3):
                v0 = a.variant & 255
                for sk in (sks if ph == "wgrad" else [None]):
                    if sk is not None:
                        a.splitk, a.partial = sk, None        # (the layer's own partial workspace is sized for its planned split)
                    for d in [int(v) for v in args.dbg.split(",")]:
                        a.variant = v0 | (d << 8)
                        us = timeit(fn)
                        print(f"{name:16s} {ph:6s} tile {a.tile_m}x{a.tile_n} splitk {a.splitk if ph == 'wgrad' else '-'} dbg {d:2d}: {us:8.1f} us  {fl / us / 1e6:7.0f} TF", flush=True)
                a.variant = v0
                if sk0 is not None:
                    a.splitk, a.partial = sk0, part0
                continue
            v0, sk_f0, part_f0 = a.variant, (a.splitk if ph != "wgrad" else None), (a.partial if ph != "wgrad" else None)
            for tile in tiles:
                for bk in (bks if ph != "wgrad" else [None]):
                    for sk in sks:
                        if tile is not None:
                            a.tile_m, a.tile_n = tile
                            if ph == "fwd" and tile[0] < t0[0]:
                                a.stats = None          # the BN partial-sum buffer is sized for the planned tile
                        if sk is not None:
                            a.splitk = sk
                            if ph == "wgrad":
                                a.partial = None          # (the layer's partial workspace is sized for its planned split)
                            else:                         # split reduction of the per-tap kernel (variant 0)
                                a.variant = 0
                                if tile is None and a.tile_m == 256 and a.tile_n != 64:
                                    a.tile_n = 64
                                nb = M._ffi.lib().seg_igemm_fwd_workspace_bytes(C.byref(a))
                                ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device="cuda")
                                a.partial = ws.data_ptr() if sk > 1 else None
                        if bk is None:
                            os.environ.pop("SEG_V2_BK", None)
                        else:
                            os.environ["SEG_V2_BK"] = bk
                        try:
                            us = timeit(fn)
                            print(f"{name:16s} {ph:6s} tile {a.tile_m}x{a.tile_n} bk {bk or 'auto':4s} splitk {a.splitk:>4} "
                                  f"M={op.M} n_rows={a.n_rows} Kpad={a.Kpad} v{a.variant & 255}: {us:8.1f} us  {fl / us / 1e6:7.0f} TF", flush=True)
                        except Exception as e:
                            print(f"{name:16s} {ph:6s} tile {tile} bk {bk}: {e}", flush=True)
            a.tile_m, a.tile_n = t0
            a.variant = v0
            if sk0 is not None:
                a.splitk, a.partial = sk0, part0
            if sk_f0 is not None:
                a.splitk, a.partial = sk_f0, part_f0
            os.environ.pop("SEG_V2_BK", None)


if __name__ == "__main__":
    main()

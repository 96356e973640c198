#!/usr/bin/env python3
"""A BASELINE workload through its own checkpoint: build it with the builder, `save_weights('x.keras')` (Keras-layout archive of graph +
weights), `segamd.load_model('x.keras')` (the reference's Test.py:114 - no constructor call), and compare: file size and write / read
times, predictions of the two models, and the time of an optimiser step of each (the imported graph is lowered like the builder's -
same launches, same step time - or this prints the difference).

    python tools/keras_roundtrip_bench.py [--workload C3] [--steps 8]

Build-authored test tooling; needs an MI355X."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import segamd


def step_ms(model, x, tg, loss, steps):
    for _ in range(3):
        model.train_step(x, tg, loss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(x, tg, loss)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    kw, H, W, Cin, bs, loss, desc = bench.WORKLOADS[args.workload]
    a = bench.build_model(args.workload, "f32")
    x, tg = bench.make_batch(a, bs, H, W, Cin, loss, 0)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"{args.workload}.keras")
        t0 = time.perf_counter()
        a.save_weights(path)
        t1 = time.perf_counter()
        b = segamd.load_model(path, batch_size=bs)
        t2 = time.perf_counter()
        size = os.path.getsize(path)
    b.compile(loss=loss, lr=2e-4)
    print(f"{desc}: archive {size / 1e6:.1f} MB, written in {t1 - t0:.1f} s, load_model + build in {t2 - t1:.1f} s; "
          f"{len(a.graph.live_nodes())} builder nodes, {len(b.graph.live_nodes())} imported nodes")
    pa, pb = a.predict(x.cpu()), b.predict(x.cpu())
    pa, pb = (pa if isinstance(pa, (list, tuple)) else [pa]), (pb if isinstance(pb, (list, tuple)) else [pb])
    print("max |prediction difference| per output:", [float(np.abs(np.asarray(p) - np.asarray(q)).max()) for p, q in zip(pa, pb)])
    ta, tb = step_ms(a, x, tg, loss, args.steps), step_ms(b, x, tg, loss, args.steps)
    ta2, tb2 = step_ms(a, x, tg, loss, args.steps), step_ms(b, x, tg, loss, args.steps)
    print(f"optimiser step: builder model {min(ta, ta2):.2f} ms, imported model {min(tb, tb2):.2f} ms")
    print("launch groups:", len(getattr(a, "ops", [])), "vs", len(getattr(b, "ops", [])))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""CPU-only fuzz of the checkpoint-file path over the builder API: random constructor calls (tools/fuzz_models.py's sampler - every decoder,
1-D and 2-D, ds / ae / ag / lstm, any width and kernel size) -> `save_weights('x.keras')` (keras_import.save_model: configuration +
model.weights.h5 through h5lite) -> the archive alone back to a graph and weights (`load_model`'s path) and onto the constructor call
(`load_weights`'s path).  The imported graph must compute what the builder's graph computes, bit for bit, in the oracle's interpreter.

    python tools/fuzz_archives.py <seed> <n>          # ~6 configurations a second

Build-authored test tooling (uses the oracle's interpreter: not part of the product path)."""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import segamd
from segamd import keras_file as KF, keras_import as KI, harness
from ir_interp import run_graph
import fuzz_models as FZ


def run(seed, n, verbose=True):
  rng = random.Random(seed)
  ok = refused = 0
  fails = []
  tmp = tempfile.mkdtemp()
  for i in range(n):
      two_d = rng.random() < 0.65
      call, make, xf, loss = (FZ.sample_2d if two_d else FZ.sample_1d)(rng)
      try:
          try:
              m = make()
          except Exception as e:
              text = f"{type(e).__name__}: {e}"
              if any(text.startswith(t) and frag in text for t, frag in FZ.EXPECTED):
                  refused += 1; continue
              raise
          g = m.graph
          wr = np.random.default_rng(i)
          w = {}
          for p in g.param_specs():
              v = wr.standard_normal(p.shape).astype(np.float32)
              w[p.name] = (np.abs(v) + 0.5 if p.name.endswith("moving_variance") else v / np.sqrt(np.prod(p.shape[:-1])) if len(p.shape) >= 2 else v).astype(np.float32)
          path = os.path.join(tmp, "m.keras")
          harness.save_weights_file(w, path, g)
          kfile = KF.KerasFileModel(path)
          gi = KI.graph_from_keras_config(kfile.config)
          wi = KI.weights_of_imported_graph(kfile, gi)
          back = harness.load_weights_file(path, g)
          assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w), "constructor-call route"
          torch.manual_seed(i)
          x = xf()
          x = x[:2]
          t = lambda d: {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in d.items()}
          a = run_graph(gi, x, t(wi), training=True)
          b = run_graph(g, x, t(w), training=True)
          assert len(a) == len(b) and all(torch.equal(p, q) or (torch.isnan(p) == torch.isnan(q)).all() and torch.equal(torch.nan_to_num(p), torch.nan_to_num(q)) for p, q in zip(a, b)), "imported graph computes something else"
          ok += 1
      except Exception as e:
          fails.append((i, call, f"{type(e).__name__}: {str(e)[:200]}"))
          if verbose and len(fails) <= 5:
              print(f"[{i}] FAIL {call}\n     {type(e).__name__}: {str(e)[:300]}", flush=True)
  return ok, refused, fails


if __name__ == "__main__":
    torch.set_num_threads(8)
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    ok, refused, fails = run(seed, n)
    print(f"==== archive round trip over random builder calls: {ok} ok, {refused} refused as the reference refuses them, {len(fails)} failed, in {time.time() - t0:.0f} s (seed {seed})")

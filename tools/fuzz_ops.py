#!/usr/bin/env python3
"""Differential fuzzing of the GEMM entry points THROUGH THE C ABI (seg_igemm_fwd / seg_igemm_wgrad, both storage modes): random
geometries - image size, batch, virtual-concat segments of any channel count, output channels, kernel 1x1 .. 5x5 / k x 1, stride 1 / 2,
conv and 4x4 / 2x2 stride-2 transposed conv -, every tile / variant / split the argument checks admit, bias, activation, BN statistics,
accumulating output - against the oracle's convolution.  An external caller of the boundary may hand over any of this: the answer must
be the right numbers or SEG_ERR_ARG with a message, never a wrong result, a fault or non-zero padding channels.

    python tools/fuzz_ops.py --n 2000 --seed 1 [--seconds 600]

Build-authored test tooling; needs an MI355X."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gpu_util as U
from oracle import keras_ref as K

ffi, M = U.ffi, U.M


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def sample(rng):
    kind = rng.choice(["conv", "conv", "conv", "convT"])
    N = rng.choice([1, 2, 3, 5])
    H = rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 40, 64])
    W = rng.choice([2, 4, 6, 8, 10, 16, 24, 32, 48, 64, 96])
    if kind == "convT":
        k, s = rng.choice([((4, 4), (2, 2)), ((2, 2), (2, 2)), ((1, 2), (1, 2)), ((1, 4), (1, 2))])
        if k[0] == 1:
            H = 1
    else:
        k = rng.choice([(3, 3), (3, 3), (1, 1), (1, 1), (4, 4), (1, 3), (1, 5), (1, 7), (2, 2), (1, 2), (1, 11), (1, 16), (2, 3), (1, 17), (1, 25), (1, 32), (5, 5)])      # (at most SEG_MAX_TAPS = 32 taps)
        s = rng.choice([(1, 1), (1, 1), (1, 1), (2, 2)]) if k != (1, 1) else rng.choice([(1, 1), (2, 2), (1, 2)])
        if k[0] == 1 and k != (1, 1):
            H, s = 1, (1, s[1])
    nseg = rng.choice([1, 1, 2, 3, 5, 8])
    cins = [rng.choice([1, 3, 5, 8, 12, 16, 17, 24, 32, 40, 64, 96]) for _ in range(nseg)]
    cout = rng.choice([1, 2, 5, 8, 12, 16, 24, 31, 32, 33, 48, 64, 72, 128, 136])
    while N * H * W * sum(cins) > 6e6 or sum(cins) * cout * k[0] * k[1] > 4e6:
        if W > 8:
            W //= 2
        elif H > 1:
            H = max(1, H // 2)
        else:
            cins = cins[:max(1, len(cins) // 2)]
    f32 = rng.random() < 0.6
    return dict(kind=kind, N=N, H=H, W=W, k=k, s=s, cins=cins, cout=cout, f32=f32,
                bias=rng.random() < 0.5, act=rng.choice([None, None, "relu", "leaky_relu", "sigmoid"]), stats=rng.random() < 0.4,
                acc=rng.random() < 0.2)


def launch_options(c, rng):
    """(tile, variant, splitk) combinations to try for this case: the engine's own plan first, then forced alternatives."""
    g = M.geom_convT(c["N"], c["H"], c["W"], *c["k"], *c["s"]) if c["kind"] == "convT" else M.geom_conv(c["N"], c["H"], c["W"], *c["k"], *c["s"])
    n_rows = U.r8(c["cout"])
    Mtot = c["N"] * g.Hl * g.Wl
    ktot = g.ntaps * sum(U.r32(U.r8(ci)) for ci in c["cins"])
    tm, tn, variant, sk = M.plan_fwd(Mtot, n_rows, g, ktot, not c["f32"])
    opts = [((tm, tn), variant, sk)]
    for t in ((128, 128), (128, 64), (64, 64), (64, 128), (256, 64)):
        opts.append((t, 0, rng.choice([0, 0, 2, 3])))
    if M.fwd_halo_patch(g):
        opts.append(((256, 64), 3, 0))
        opts.append(((256, 32 if (c["f32"] and n_rows <= 32) else 128 if not c["f32"] else 64), 3, 0))
    rng.shuffle(opts)
    return opts[:4]


def run_fwd(c, rng):
    xs = [torch.randn(c["N"], c["H"], c["W"], ci) for ci in c["cins"]]
    cin = sum(c["cins"])
    kh, kw = c["k"]
    transposed = c["kind"] == "convT"
    kern = ((np.random.randn(kh, kw, c["cout"], cin) if transposed else np.random.randn(kh, kw, cin, c["cout"])) / np.sqrt(cin * kh * kw)).astype(np.float32)
    bias = np.random.randn(c["cout"]).astype(np.float32) if c["bias"] else None
    alpha = 0.3 if c["act"] == "leaky_relu" else 0.0
    if not c["f32"]:
        xs = [U.h16(x) for x in xs]
        kern = U.h16(torch.from_numpy(kern)).numpy()
    f = K.conv2d_transpose_same if transposed else K.conv2d_same
    pre = f(torch.cat(xs, -1), torch.from_numpy(kern), torch.from_numpy(bias) if bias is not None else None, c["s"])
    want = K.activation(pre, c["act"], alpha) if c["act"] not in (None, "leaky_relu") else (torch.where(pre > 0, pre, pre * alpha) if c["act"] == "leaky_relu" else pre)
    base = torch.randn_like(want) if c["acc"] else None
    tol = 2e-5 if c["f32"] else 4e-3
    msgs = []
    for tile, variant, sk in launch_options(c, rng):
        what = f"tile {tile} variant {variant} splitk {sk}"
        try:
            out, st, pad = U.igemm_conv(xs, kern, bias, c["kind"], c["s"], c["act"], alpha, c["stats"] and not c["acc"], tile,
                                        accumulate_into=base, f32=c["f32"], variant=variant, splitk=sk)
        except ffi.SegHipError as e:
            msgs.append(("refused", what, str(e)[:120]))
            continue
        ref = want + base if base is not None else want
        if not c["f32"]:
            ref = ref if base is None else want + U.h16(base)
        e = rel(out, ref)
        bad = not (e < tol) or pad != 0.0
        if st is not None and not bad:
            # BN statistics of the pre-activation values (bias included), summed over the partial rows
            s_ref = pre.reshape(-1, c["cout"]).sum(0)
            q_ref = (pre.reshape(-1, c["cout"]) ** 2).sum(0)
            es = float((st[0][:c["cout"]] - s_ref).abs().max() / (s_ref.abs().max() + 1.0))
            eq = float((st[1][:c["cout"]] - q_ref).abs().max() / (q_ref.abs().max() + 1.0))
            bad = not (es < 5 * tol and eq < 5 * tol)
            if bad:
                what += f" statistics {es:.1e} / {eq:.1e}"
        msgs.append(("FAIL" if bad else "ok", what, f"err {e:.1e} pad {pad:.1e}"))
    return msgs


def run_wgrad(c, rng):
    xs = [torch.randn(c["N"], c["H"], c["W"], ci) for ci in c["cins"]]
    cin = sum(c["cins"])
    kh, kw = c["k"]
    transposed = c["kind"] == "convT"
    kshape = (kh, kw, c["cout"], cin) if transposed else (kh, kw, cin, c["cout"])
    if not c["f32"]:
        xs = [U.h16(x) for x in xs]
    kz = torch.zeros(kshape, requires_grad=True)
    f = K.conv2d_transpose_same if transposed else K.conv2d_same
    y = f(torch.cat(xs, -1), kz, None, c["s"])
    dz = torch.randn_like(y)
    if not c["f32"]:
        dz = U.h16(dz)
    (y * dz).sum().backward()
    want = kz.grad
    tol = 2e-5 if c["f32"] else 4e-3
    g = M.geom_convT(c["N"], c["H"], c["W"], kh, kw, *c["s"]) if transposed else M.geom_conv(c["N"], c["H"], c["W"], kh, kw, *c["s"])
    msgs = []
    variants = [0] + ([2] if M.wgrad_halo_patch(g) else [])
    for variant in variants:
        for sk, partial in ((None, False), (rng.choice([2, 3, 7]), True), (rng.choice([2, 5]), False)):
            what = f"wgrad variant {variant} splitk {sk} partial {partial}"
            try:
                got = U.igemm_wgrad(xs, dz, kshape, c["kind"], c["s"], variant, sk, None, f32=c["f32"], partial=partial)
            except ffi.SegHipError as e:
                msgs.append(("refused", what, str(e)[:120]))
                continue
            e = rel(torch.from_numpy(got), want)
            msgs.append(("FAIL" if not (e < tol) else "ok", what, f"err {e:.1e}"))
    return msgs


def run(n, seed, seconds=0.0, start=0, verbose=True):
    """Cases [start, n) of the seed's sequence: (launches right, launches refused with an argument error, [(case, description, what)])."""
    rng = random.Random(seed)
    t0 = time.time()
    n_ok = n_ref = 0
    fails = []
    say = print if verbose else (lambda *a, **k: None)
    for i in range(n):
        if seconds and time.time() - t0 > seconds:
            break
        c = sample(rng)
        torch.manual_seed(seed * 7919 + i)
        np.random.seed(seed * 7919 + i)
        if i < start:
            continue
        desc = f"{c['kind']} {c['N']}x{c['H']}x{c['W']} cins {c['cins']} -> {c['cout']} k {c['k']} s {c['s']} {'f32' if c['f32'] else 'f16'} bias {c['bias']} act {c['act']} stats {c['stats']} acc {c['acc']}"
        say(f"[{i}] BEGIN {desc}", flush=True)
        try:
            msgs = run_fwd(c, rng) + run_wgrad(c, rng)
        except Exception as e:          # noqa: BLE001
            fails.append((i, desc, f"RAISED {type(e).__name__}: {str(e)[:200]}"))
            say(f"[{i}] RAISED {desc}: {type(e).__name__}: {str(e)[:200]}", flush=True)
            torch.cuda.synchronize()
            continue
        for st, what, info in msgs:
            if st == "ok":
                n_ok += 1
            elif st == "refused":
                n_ref += 1
            else:
                fails.append((i, desc, what + " " + info))
        bad = [m for m in msgs if m[0] == "FAIL"]
        say(f"[{i}] {'FAIL' if bad else 'ok  '} {desc} | " + "; ".join(f"{w}: {inf}" for s_, w, inf in msgs if s_ != "ok" or bad)[:400], flush=True)
    return n_ok, n_ref, fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--start", type=int, default=0)
    args = ap.parse_args()
    torch.set_num_threads(16)
    t0 = time.time()
    n_ok, n_ref, fails = run(args.n, args.seed, args.seconds, args.start)
    print(f"\n==== {n_ok} launches ok, {n_ref} refused with an argument error, {len(fails)} failed, in {time.time() - t0:.0f} s (seed {args.seed})")
    seen = set()
    for i, desc, info in fails:
        key = info[:60]
        if key in seen:
            continue
        seen.add(key)
        print(f"FAIL [{i}] {desc}\n     {info}")


if __name__ == "__main__":
    main()

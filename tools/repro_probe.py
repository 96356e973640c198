"""Bit-reproducibility probe (GPU): two identical forward + backward passes of C3 at batch 16 in one process; reports the first
tensors (activations, activation gradients, parameter gradients) that differ.  tools/lib_run.sh runs it against several builds."""
import sys, os, torch, numpy as np
sys.path.insert(0, "tests"); import conftest
import test_gpu_fullsize as T

def run():
    torch.manual_seed(5)
    model, x, targets, loss = T.CONFIGS["C3"](16)
    model.build(16, dtype="f16")
    model.compile(loss=loss, lr=2e-4)
    for _ in range(PRE_STEPS):                      # full optimiser steps first: the next pass then runs beside the update
        model.train_step(x, targets)
    model.forward(x, training=True)
    model.compute_loss(targets, loss)
    model.backward(None)
    model._join_side()
    torch.cuda.synchronize()
    acts = {n: t.buf.clone() for n, t in model.tensors.items()}
    acts["P16"] = model.store.P16.clone(); acts["P"] = model.store.P.clone(); acts["S"] = model.store.S.clone()
    for i, h in enumerate(model.out_heads):
        acts[f"head{i}.y"] = h.y.clone(); acts[f"head{i}.dy"] = h.dy.clone()
    agr = {n: t.grad.clone() for n, t in model.tensors.items() if t.grad is not None}
    return model, model.store.G.clone(), acts, agr

PRE_STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    m0, g0, a0, d0 = run()
    m1, g1, a1, d1 = run()
    bad_a = [n for n in a0 if not torch.equal(a0[n], a1[n])]
    bad_d = [n for n in d0 if not torch.equal(d0[n], d1[n])]
    st = m0.store
    bad_g = []
    for n in sorted(st.slots, key=lambda n: st.slots[n].off):
        sl = st.slots[n]
        if sl.trainable and not torch.equal(g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]):
            bad_g.append(n)
    if not (bad_a or bad_d or bad_g):
        print(f"trial {trial}: identical", flush=True)
        del m0, m1, g0, g1, a0, a1, d0, d1
        torch.cuda.empty_cache()
        continue
    order = [getattr(getattr(op, "node", None), "name", type(op).__name__) for op in m0.ops]
    print("   parameter gradients: relative L2 difference %.3e, max abs %.3e of max %.3e" % (float((g0.double() - g1.double()).norm() / g0.double().norm()), float((g0 - g1).abs().max()), float(g0.abs().max())))
    print("   master weights / fp16 weights / BN state differ:", "P" in bad_a, "P16" in bad_a, "S" in bad_a, " enc1/conv:z differs:", "enc1/conv:z" in bad_a)
    print("   first differing activations in forward order:", [n for n in order if n in bad_a or (n + ":z") in bad_a][:5])
    for n in bad_d:
        dd = (d0[n].float() - d1[n].float()).abs()
        print(f"      d/{n:20s} ndiff {int((dd > 0).sum()):9d} of {dd.numel():9d}  max {float(dd.max()):.3e}  ref max {float(d0[n].float().abs().max()):.3e}")
    if "X41/relu" in bad_d:
        dd = (d0["X41/relu"].float() - d1["X41/relu"].float()).abs()          # (N, H, W, C)
        chs = torch.nonzero(dd.sum((0, 1, 2)) > 0).flatten().tolist()
        pix = dd.sum(3) > 0
        print("      X41/relu: channels with differences", len(chs), chs[:40], "| pixels with differences", int(pix.sum()), "of", pix.numel(),
              "| images", torch.nonzero(pix.sum((1, 2)) > 0).flatten().tolist())
        for opx in m0.ops:
            if getattr(getattr(opx, "node", None), "name", "") == "X41/conv":
                opy = [o for o in m1.ops if getattr(getattr(o, "node", None), "name", "") == "X41/conv"][0]
                for nm in ("bpart", "c1", "c2"):
                    t0, t1 = getattr(opx, nm), getattr(opy, nm)
                    dv = (t0 - t1).abs()
                    print("      X41/conv", nm, "equal", torch.equal(t0, t1), "ndiff", int((dv > 0).sum()), "of", dv.numel(), "max", float(dv.max()))
    print("   differing activation grads, backward order:", [n for n in reversed(order) if n in bad_d or (n + ":z") in bad_d][:8])
    print(f"trial {trial}: activations differing {len(bad_a)} {bad_a[:4]}; activation grads differing {len(bad_d)} {bad_d[:6]}; "
          f"param grads differing {len(bad_g)} {bad_g[-6:]}", flush=True)
    del m0, m1, g0, g1, a0, a1, d0, d1
    torch.cuda.empty_cache()

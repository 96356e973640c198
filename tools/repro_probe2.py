"""The reproducibility test's own protocol (3 optimiser steps of C3, compare P and S), repeated; names what differs."""
import sys, torch
sys.path.insert(0, "tests"); import conftest
import test_gpu_fullsize as T
def run():
    torch.manual_seed(5)
    model, x, targets, loss = T.CONFIGS["C3"](16)
    model.build(16, dtype="f16")
    model.compile(loss=loss, lr=2e-4)
    snaps = []
    for _ in range(3):
        model.train_step(x, targets)
        model._join_side(); torch.cuda.synchronize()
        snaps.append((model.store.P.clone(), model.store.S.clone())) if SNAP else None
    model._join_side()
    torch.cuda.synchronize()
    return model, model.store.P.clone(), model.store.S.clone(), snaps
SNAP = len(sys.argv) > 2 and sys.argv[2] == "snap"
ref = None
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    m, p, s, snaps = run()
    if ref is None:
        ref = (p, s, snaps); print("trial 0 = reference", flush=True); continue
    st = m.store
    badp = [n for n in sorted(st.slots, key=lambda n: st.slots[n].off) if st.slots[n].trainable and not torch.equal(p[st.slots[n].off:st.slots[n].off + st.slots[n].numel], ref[0][st.slots[n].off:st.slots[n].off + st.slots[n].numel])]
    first = [i for i, (a, b) in enumerate(zip(snaps, ref[2])) if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))]
    print(f"trial {trial}: P equal {torch.equal(p, ref[0])} S equal {torch.equal(s, ref[1])} differing params {len(badp)} {badp[:3]} ... {badp[-3:]} first differing step {first[:1]}", flush=True)
    del m
    torch.cuda.empty_cache()

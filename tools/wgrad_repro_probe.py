"""Bit-reproducibility stress of the halo weight-gradient kernels at op level (GPU): every generation, deep-level and level-0
shapes, REPS identical launches each, optionally with another kernel running beside on a second stream."""
import sys, torch, numpy as np
sys.path.insert(0, "tests"); import conftest
import gpu_util as U
from gpu_util import h16
torch.manual_seed(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
def check(name, xs, dz, kshape, kind, strides, variant, sk, partial=True):
    outs = [U.igemm_wgrad(xs, dz, kshape, kind, strides, variant, sk, None, partial=partial and sk > 1) for _ in range(REPS)]
    nbad = sum(0 if np.array_equal(outs[0], o) else 1 for o in outs[1:])
    md = max(float(np.abs(outs[0] - o).max()) for o in outs[1:])
    print(f"{name:44s} differing launches {nbad}/{REPS - 1} maxdiff={md:.3e}", flush=True)
GEN = (("auto", 0), ("r1", 0x80), ("ts", 0x100), ("ws", 0x40))
# deep level: 16 x 16 x 16 images, 1024 + 1024 -> 1024 channels (X41-like), split 1
xs = [h16(torch.randn(16, 16, 16, 1024)), h16(torch.randn(16, 16, 16, 1024))]
dz = h16(torch.randn(16, 16, 16, 1024))
for gen, bits in GEN:
    check("deep 16x16 2048->1024 sk1 " + gen, xs, dz, (3, 3, 2048, 1024), "conv", (1, 1), 2 | (bits << 8), 1)
xs = [h16(torch.randn(16, 32, 32, 512)), h16(torch.randn(16, 32, 32, 512))]
dz = h16(torch.randn(16, 32, 32, 512))
for gen, bits in GEN:
    check("level3 32x32 1024->512 sk2 " + gen, xs, dz, (3, 3, 1024, 512), "conv", (1, 1), 2 | (bits << 8), 2)
xt = [h16(torch.randn(16, 16, 16, 1024))]
dzt = h16(torch.randn(16, 32, 32, 512))
check("convT 16x16 1024->512 (tc2) sk2", xt, dzt, (4, 4, 512, 1024), "convT", (2, 2), 2, 2)
xt = [h16(torch.randn(16, 8, 8, 2048))]
dzt = h16(torch.randn(16, 16, 16, 1024))
check("convT 8x8 2048->1024 (tc2) sk1", xt, dzt, (4, 4, 1024, 2048), "convT", (2, 2), 2, 1)

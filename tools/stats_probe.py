import sys, torch, numpy as np
sys.path.insert(0, "tests"); import conftest
import gpu_util as U
from gpu_util import h16
from oracle import keras_ref as K
torch.manual_seed(0)
rng = np.random.default_rng(1)
N, H, W = 8, 256, 256
x64 = [h16(torch.randn(N, H, W, 64))]
k64 = (rng.standard_normal((3, 3, 64, 64)) * 0.05).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
pre = K.conv2d_same(x64[0], torch.from_numpy(k64.astype(np.float16).astype(np.float32)), torch.from_numpy(b)).double()
ref = torch.stack([pre.sum((0, 1, 2)), (pre * pre).sum((0, 1, 2))])
for rep in range(6):
    out, st, pad = U.igemm_conv(x64, k64, b, "conv", (1, 1), None, 0.0, True, (256, 64), variant=3)
    err = ((st[:, :64].double() - ref).abs() / ref.abs().clamp_min(1.0))
    print(rep, "max rel err sum %.3e sumsq %.3e" % (float(err[0].max()), float(err[1].max())), "worst channels", torch.topk(err[1], 4).indices.tolist())

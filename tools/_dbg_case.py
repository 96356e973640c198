import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import segamd
import gpu_util as U
which = sys.argv[1]
torch.manual_seed(3)
if which == "22":
    m = segamd.UNet1D(64, 3, 2, 8, 3, 'Regression', output_nums=2, ds=0, ae=1, ag=1, lstm=0, feature_number=8, is_transconv=True, alpha=1).UNetPP()
    x = torch.rand(4, 64, 2)
elif which == "4":
    m = segamd.BCDUNet(40, 3, 1, 16, 3, 'Regression', output_nums=3, ds=0, ae=0, ag=1, lstm=1, feature_number=40, is_transconv=False, dense_loop=1).BCDUNet()
    x = torch.rand(2, 40, 1)
print("build", flush=True)
m.build(x.shape[0], dtype="f32")
g = m.graph
tg = []
for o in g.outputs:
    H, W, C = g.nodes[o].shape
    tg.append(torch.rand(x.shape[0], W, C) if g.ndim == 1 else torch.rand(x.shape[0], H, W, C))
print("forward", flush=True)
outs = m.forward(x, training=True); torch.cuda.synchronize()
print("loss", flush=True)
lv = m.compute_loss(tg, "mse"); torch.cuda.synchronize()
print("backward", flush=True)
m.backward(None); torch.cuda.synchronize()
print("grads", flush=True)
gr = m.get_grads()
bad = [k for k, v in gr.items() if not np.isfinite(v).all()]
print("non-finite gradient tensors:", bad, flush=True)
print("compare", flush=True)
r = U.compare_model(m, x, tg, "mse", verbose=(which == "4"), impose_decisions=True)
for k, (e, n) in r["gl2_imposed"].items():
    if not (e == e) or e > 1e-3 * (n + 1e-9):
        print("  bad grad", k, e, n)
print("done", flush=True)

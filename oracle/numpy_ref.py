"""ORACLE (test infrastructure) - second, loop-level restatement of the Keras-2.x primitives in plain numpy.
Used only to pin oracle/keras_ref.py (the torch-CPU implementation) on small cases; PARITY UNPINNED against the
reference itself (see keras_ref.py).  Index formulas follow SURVEY.md section 8(c)."""
import numpy as np


def same_pad_before(n_in, k, s):
    out = -(-n_in // s)
    return max((out - 1) * s + k - n_in, 0) // 2


def conv2d_same(x, kern, bias=None, strides=(1, 1)):
    """out[n,i,j,co] = b[co] + sum_{ky,kx,ci} x[n, i*sh + ky - pt, j*sw + kx - pl, ci] * kern[ky,kx,ci,co]  (zero outside)."""
    N, H, W, Ci = x.shape
    kh, kw, _, Co = kern.shape
    sh, sw = strides
    pt, pl = same_pad_before(H, kh, sh), same_pad_before(W, kw, sw)
    Ho, Wo = -(-H // sh), -(-W // sw)
    y = np.zeros((N, Ho, Wo, Co), np.float64)
    for i in range(Ho):
        for j in range(Wo):
            for ky in range(kh):
                for kx in range(kw):
                    yy, xx = i * sh + ky - pt, j * sw + kx - pl
                    if 0 <= yy < H and 0 <= xx < W:
                        y[:, i, j, :] += x[:, yy, xx, :].astype(np.float64) @ kern[ky, kx].astype(np.float64)
    if bias is not None:
        y += bias
    return y


def conv2d_transpose_same(x, kern, bias=None, strides=(2, 2)):
    """kern (kh,kw,Cout,Cin); y[n, s*i + ky - pb, ...] += x[n,i,...,ci] * kern[ky,kx,co,ci]; out = in * stride."""
    N, H, W, Ci = x.shape
    kh, kw, Co, _ = kern.shape
    sh, sw = strides
    pbh, pbw = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
    y = np.zeros((N, H * sh, W * sw, Co), np.float64)
    for i in range(H):
        for j in range(W):
            for ky in range(kh):
                for kx in range(kw):
                    Y, X = i * sh + ky - pbh, j * sw + kx - pbw
                    if 0 <= Y < H * sh and 0 <= X < W * sw:
                        y[:, Y, X, :] += x[:, i, j, :].astype(np.float64) @ kern[ky, kx].astype(np.float64).T
    if bias is not None:
        y += bias
    return y


def maxpool(x, size):
    ph, pw = size
    N, H, W, C = x.shape
    y = np.zeros((N, H // ph, W // pw, C), x.dtype)
    for i in range(H // ph):
        for j in range(W // pw):
            y[:, i, j] = x[:, i * ph:(i + 1) * ph, j * pw:(j + 1) * pw].max((1, 2))
    return y


def upsample_bilinear(x, f):
    """tf.image.resize bilinear, half-pixel centres, edge clamp."""
    N, H, W, C = x.shape
    fh, fw = f

    def src(o, fac, n):
        s = (o + 0.5) / fac - 0.5
        lo = int(np.floor(s))
        w = s - lo
        return min(max(lo, 0), n - 1), min(max(lo + 1, 0), n - 1), w
    y = np.zeros((N, H * fh, W * fw, C), np.float64)
    for i in range(H * fh):
        y0, y1, wy = src(i, fh, H)
        for j in range(W * fw):
            x0, x1, wx = src(j, fw, W)
            top = x[:, y0, x0] * (1 - wx) + x[:, y0, x1] * wx
            bot = x[:, y1, x0] * (1 - wx) + x[:, y1, x1] * wx
            y[:, i, j] = top * (1 - wy) + bot * wy
    return y


def batchnorm_train(x, gamma, beta, mm, mv, eps=1e-3, momentum=0.99, unbiased_moving=True):
    n = x.shape[0] * x.shape[1] * x.shape[2]
    mean = x.reshape(n, -1).mean(0)
    var = x.reshape(n, -1).var(0)
    y = (x - mean) / np.sqrt(var + eps) * gamma + beta
    upd = var * n / (n - 1) if unbiased_moving else var
    return y, mm * momentum + mean * (1 - momentum), mv * momentum + upd * (1 - momentum)


def hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def convlstm_t1(x, kern, bias):
    z = conv2d_same(x, kern, bias)
    F = z.shape[-1] // 4
    zi, zc, zo = z[..., :F], z[..., 2 * F:3 * F], z[..., 3 * F:]
    return hard_sigmoid(zo) * np.tanh(hard_sigmoid(zi) * np.tanh(zc))

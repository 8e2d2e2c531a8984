"""Oracle ops: torch-CPU restatement of the reference op layer (NHWC everywhere).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows; [TF1] marks semantics that live inside TensorFlow
1.x itself and are therefore restated, not imported.
All tensors are NHWC like the reference; weights are in TF layouts
(conv HWIO `[kh,kw,Cin,Cout]`, conv_transpose `[kh,kw,Cout,Cin]`).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# value-range helpers -- lib/ops.py:13-32
# --------------------------------------------------------------------------- #
def preprocess(image):          # lib/ops.py:13-16   [0,1] -> [-1,1]
    return image * 2 - 1


def deprocess(image):           # lib/ops.py:19-22   [-1,1] -> [0,1]
    return (image + 1) / 2


def preprocessLR(image):        # lib/ops.py:25-27   identity
    return image


def deprocessLR(image):         # lib/ops.py:30-32   identity
    return image


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------- #
# conv / conv_transpose -- lib/ops.py:35-56 (slim.conv2d / conv2d_transpose)
# --------------------------------------------------------------------------- #
def same_pad(in_size, k, s):
    """[TF1] SAME padding (Appendix A.1): returns (out, pad_before, pad_after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv2(x, w, b=None, stride=1):
    """lib/ops.py:47-56: slim.conv2d(k, stride, 'SAME', NHWC, no activation).

    x [N,H,W,Cin], w HWIO [kh,kw,Cin,Cout], b [Cout] or None.  [TF1] A.1.
    """
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pad(x.shape[1], kh, stride)
    _, pl, pr = same_pad(x.shape[2], kw, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=stride)
    return _nhwc(y)


def conv2_tran(x, w, b=None, stride=2):
    """lib/ops.py:35-44: slim.conv2d_transpose(k, stride, 'SAME').

    w is TF layout [kh,kw,Cout,Cin].  [TF1] A.2: the op is the input-gradient of
    a SAME conv on the (stride*n)-sized output, i.e.
    y[s*i+ky-pb, s*j+kx-pb'] += x[i,j,ci]*w[ky,kx,co,ci] with pb = the SAME
    pad_before of that forward conv (0 for k3 s2), cropped to [0, s*n).
    """
    kh, kw = w.shape[0], w.shape[1]
    n_h, n_w = x.shape[1] * stride, x.shape[2] * stride
    _, pt, _ = same_pad(n_h, kh, stride)
    _, pl, _ = same_pad(n_w, kw, stride)
    # torch ConvTranspose weight is [Cin, Cout, kh, kw]
    y = F.conv_transpose2d(_nchw(x), w.permute(3, 2, 0, 1), None, stride=stride)
    y = y[:, :, pt:pt + n_h, pl:pl + n_w]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


class _ReluTF(torch.autograd.Function):
    """tf.nn.relu: gradient is (y > 0), i.e. 0 at exactly 0."""

    @staticmethod
    def forward(ctx, x):
        y = x.clamp_min(0)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g * (y > 0).to(g.dtype)


def relu(x):
    return _ReluTF.apply(x)


class _LReluTF(torch.autograd.Function):
    """keras LeakyReLU(alpha) (lib/ops.py:84-85): x>0 ? x : alpha*x; grad at 0 is alpha."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return torch.where(x > 0, x, x * alpha)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return torch.where(x > 0, g, g * ctx.alpha), None


def lrelu(x, alpha):            # lib/ops.py:84-85
    return _LReluTF.apply(x, alpha)


def maxpool(x):
    """lib/ops.py:92-93: slim.max_pool2d([2,2]) -> stride 2, VALID [TF1] A.3."""
    return _nhwc(F.max_pool2d(_nchw(x), 2, 2))


def batchnorm(x, beta, eps=1e-3):
    """lib/ops.py:88-90 with is_training=True always (lib/Teco.py:38), scale=False.

    [TF1] A.7: normalise with the biased batch variance over (N,H,W), add beta.
    Returns (y, batch_mean, batch_var) (the moving stats are updated by the
    caller with decay 0.9 but never used by the path).
    """
    mean = x.mean(dim=(0, 1, 2))
    var = x.var(dim=(0, 1, 2), unbiased=False)
    y = (x - mean) * torch.rsqrt(var + eps) + beta
    return y, mean, var


def denselayer(x, kernel, bias):
    """lib/ops.py:96-103: tf.layers.Dense(output_size) on the last axis, with bias (A.8)."""
    return x @ kernel + bias


# --------------------------------------------------------------------------- #
# legacy resizes
# --------------------------------------------------------------------------- #
def resize_bilinear_legacy(x, out_h, out_w):
    """[TF1] A.4: tf.image.resize_images bilinear, align_corners=False, no half-pixel.

    src = dst * in/out ; lo = floor(src); hi = min(lo+1, in-1); lerp (src-lo).
    Used at lib/frvsr.py:21-22 (x2) and lib/Teco.py:244 (x4).
    """
    n, h, w, c = x.shape

    def axis(in_size, out_size):
        src = torch.arange(out_size, dtype=x.dtype) * (in_size / out_size)
        lo = src.floor().long()
        hi = torch.clamp(lo + 1, max=in_size - 1)
        return lo, hi, (src - lo.to(x.dtype))

    ylo, yhi, ya = axis(h, out_h)
    xlo, xhi, xa = axis(w, out_w)
    top = x[:, ylo]
    bot = x[:, yhi]
    ya = ya.view(1, -1, 1, 1)
    xa = xa.view(1, 1, -1, 1)
    tl, tr = top[:, :, xlo], top[:, :, xhi]
    bl, br = bot[:, :, xlo], bot[:, :, xhi]
    t = tl + (tr - tl) * xa
    b = bl + (br - bl) * xa
    return t + (b - t) * ya


def upsample2_legacy(x):
    """lib/frvsr.py:21-22: resize_images(net, 2*shape) (bilinear legacy)."""
    return resize_bilinear_legacy(x, x.shape[1] * 2, x.shape[2] * 2)


def upscale_four(x):
    """lib/ops.py:126-163: fixed-ratio bilinear x4 built from slices.

    Restated op-for-op: pad bottom/right by replication (134-135), 16 phase
    blends with weights (1-.25hi)(1-.25wj) etc. (149-156), interleave (158-161).
    """
    b, h, w, c = x.shape
    p = torch.cat((x, x[:, -1:]), dim=1)
    p = torch.cat((p, p[:, :, -1:]), dim=2)
    tl, tr = x, p[:, :-1, 1:]
    bl, br = p[:, 1:, :-1], p[:, 1:, 1:]
    arr = []
    for hi in range(4):
        for wj in range(4):
            arr.append(tl * (1.0 - 0.25 * hi) * (1.0 - 0.25 * wj)
                       + tr * (1.0 - 0.25 * hi) * (0.25 * wj)
                       + bl * (0.25 * hi) * (1.0 - 0.25 * wj)
                       + br * (0.25 * hi) * (0.25 * wj))
    hr = torch.stack(arr, dim=3).reshape(b, h, w, 4, 4, c)
    return hr.permute(0, 1, 3, 2, 4, 5).reshape(b, h * 4, w * 4, c)


def bicubic_weights(dtype=torch.float32):
    """lib/ops.py:186-188: Keys a=-0.75 weights for t in {0,.25,.5,.75} (float32 math)."""
    import numpy as np
    r = 0.75
    mat = np.float32([[0, 1, 0, 0], [-r, 0, r, 0], [2 * r, r - 3, 3 - 2 * r, -r], [-r, 2 - r, r - 2, r]])
    ws = [np.float32([1.0, t, t * t, t * t * t]).dot(mat) for t in [0.0, 0.25, 0.5, 0.75]]
    return torch.tensor(np.stack(ws), dtype=dtype)      # [phase, tap]


def bicubic_four(x):
    """lib/ops.py:166-212: Keys(-0.75) x4, replicate pad 1 top/left, 2 bottom/right.

    Rows first (taps p[i..i+3]) then columns, no clamping of the result (A.6).
    """
    b, h, w, c = x.shape
    wts = bicubic_weights(x.dtype)
    p = torch.cat((x[:, :1], x), dim=1)
    p = torch.cat((p[:, :, :1], p), dim=2)
    p = torch.cat((p, p[:, -1:], p[:, -1:]), dim=1)
    p = torch.cat((p, p[:, :, -1:], p[:, :, -1:]), dim=2)
    bins = [p[:, bi:bi + h] for bi in range(4)]
    rows = []
    for hi in range(4):
        cw = wts[hi]
        rows.append(cw[0] * bins[0] + cw[1] * bins[1] + cw[2] * bins[2] + cw[3] * bins[3])
    hy = torch.stack(rows, dim=2).reshape(b, h * 4, w + 3, c)
    bins = [hy[:, :, bj:bj + w] for bj in range(4)]
    cols = []
    for hj in range(4):
        cw = wts[hj]
        cols.append(cw[0] * bins[0] + cw[1] * bins[1] + cw[2] * bins[2] + cw[3] * bins[3])
    return torch.stack(cols, dim=3).reshape(b, h * 4, w * 4, c)


# --------------------------------------------------------------------------- #
# space-to-depth and D-input packing (index reshuffles: bit-exact)
# --------------------------------------------------------------------------- #
def space_to_depth4(x):
    """lib/Teco.py:145-148 (reshape/transpose form) == tf.space_to_depth(x,4) main.py:201.

    out[b,i,j,(dy*4+dx)*C+c] = x[b,4i+dy,4j+dx,c].
    """
    b, hh, ww, c = x.shape
    h, w = hh // 4, ww // 4
    y = x.reshape(b, h, 4, w, 4, c).permute(0, 1, 3, 2, 4, 5)
    return y.reshape(b, h, w, 16 * c)


def depth_to_space4(y, c=3):
    b, h, w, _ = y.shape
    x = y.reshape(b, h, w, 4, 4, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(b, h * 4, w * 4, c)


def pack_triplets(frames, t_batch):
    """lib/Teco.py:227-229 / 236-238: [tb*3,H,W,3] -> [tb,H,W,9], channel = c*3 + t."""
    n, hh, ww, c = frames.shape
    x = frames.reshape(t_batch, 3, hh, ww, c).permute(0, 2, 3, 4, 1)
    return x.reshape(t_batch, hh, ww, c * 3)


def crop_pad_dt(x, offset):
    """lib/Teco.py:230-234: crop_to_bounding_box(offset,offset,size,size) then zero-pad back."""
    if offset == 0:
        return x
    hh, ww = x.shape[1], x.shape[2]
    y = torch.zeros_like(x)
    y[:, offset:hh - offset, offset:ww - offset] = x[:, offset:hh - offset, offset:ww - offset]
    return y


# --------------------------------------------------------------------------- #
# dense_image_warp -- lib/Teco.py:120,140,224,254 ; main.py:215
# --------------------------------------------------------------------------- #
class _WarpAlpha(torch.autograd.Function):
    """[TF1] alpha = minimum(maximum(0, q - floor), 1) with TF's tie rules.

    tf.maximum(min_alpha, alpha) sends the gradient to its FIRST argument on a
    tie, so alpha gets gradient only for 0 < alpha_raw; tf.minimum(x, max_alpha)
    passes it for alpha_raw <= 1.  Net: d(alpha)/d(q) = 1 iff 0 < alpha_raw <= 1.
    """

    @staticmethod
    def forward(ctx, raw):
        ctx.save_for_backward(raw)
        return raw.clamp(0, 1)

    @staticmethod
    def backward(ctx, g):
        (raw,) = ctx.saved_tensors
        return g * ((raw > 0) & (raw <= 1)).to(g.dtype)


def dense_image_warp(image, flow):
    """[TF1] A.5: tf.contrib.image.dense_image_warp(image[B,H,W,C], flow[B,H,W,2]).

    query = (y - flow[...,0], x - flow[...,1]); per axis floor clamped to
    [0,size-2], alpha = clamp(q - floor, 0, 1); interp = top + ay*(bot-top) with
    top = tl + ax*(tr-tl).  Gradient to the image via the 4 gathers, to the flow
    via alpha only.
    """
    b, h, w, c = image.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=image.dtype), torch.arange(w, dtype=image.dtype),
                            indexing="ij")
    qy = gy.unsqueeze(0) - flow[..., 0]
    qx = gx.unsqueeze(0) - flow[..., 1]

    def axis(q, size):
        fl = q.detach().floor().clamp(0, size - 2)
        alpha = _WarpAlpha.apply(q - fl)
        return fl.long(), alpha

    fy, ay = axis(qy, h)
    fx, ax = axis(qx, w)
    flat = image.reshape(b, h * w, c)

    def gather(iy, ix):
        idx = (iy * w + ix).reshape(b, h * w, 1).expand(-1, -1, c)
        return torch.gather(flat, 1, idx).reshape(b, h, w, c)

    tl = gather(fy, fx)
    tr = gather(fy, fx + 1)
    bl = gather(fy + 1, fx)
    br = gather(fy + 1, fx + 1)
    ax = ax.unsqueeze(-1)
    ay = ay.unsqueeze(-1)
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


# --------------------------------------------------------------------------- #
# VGG helpers -- lib/Teco.py:5-24
# --------------------------------------------------------------------------- #
VGG_MEAN = (123.68, 116.78, 103.94)     # lib/Teco.py:3


def vgg_preprocess(x):
    """lib/Teco.py:9-10: deprocess, *255, minus RGB mean."""
    return deprocess(x) * 255.0 - torch.tensor(VGG_MEAN, dtype=x.dtype)


def vgg_norm(f):
    """lib/Teco.py:20-21: f / sqrt(sum_c f^2 + 1e-12)."""
    return f / torch.sqrt((f * f).sum(dim=3, keepdim=True) + 1e-12)


# --------------------------------------------------------------------------- #
# optimiser pieces -- lib/Teco.py:95-99,415-417,425,439-440
# --------------------------------------------------------------------------- #
def exponential_decay(lr0, step, decay_steps, decay_rate, staircase=False):
    """[TF1] tf.train.exponential_decay: lr0 * rate^(step/decay_steps) (lib/Teco.py:97-98)."""
    p = step / decay_steps
    if staircase:
        p = math.floor(p)
    return lr0 * decay_rate ** p


def adam_tf_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """[TF1] tf.train.AdamOptimizer update for step t (1-based), in place.

    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= lr_t * m / (sqrt(v) + eps)   (eps OUTSIDE the bias-corrected root).
    """
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    p.sub_(lr_t * m / (v.sqrt() + eps))


def ema_tf(shadow, value, decay=0.99):
    """[TF1] ExponentialMovingAverage.apply without num_updates: shadow -= (1-d)(shadow-value).

    The shadow starts at 0 for a tensor that is not a Variable (zero_debias off),
    lib/Teco.py:415-417,433-435.
    """
    return shadow - (1.0 - decay) * (shadow - value)

"""Drop-in for the reference's `lib/ops.py` operator layer, backed by libtecogan_hip.so on MI355X.

Same function names, argument order and tensor conventions as the reference (NHWC, fp32 at the boundary,
LR in [0,1], HR in [-1,1]); tensors are torch CUDA tensors instead of tf.Tensors, and the TF variable
scopes are emulated by a small registry (`variable_scope`, `get_variable`) that produces the SAME variable
names (`<scope>/Conv/weights`, ... -- SURVEY.md Appendix B), so checkpoints keyed by TF names interchange.

These op-level entry points run eagerly in fp32 (parity mode) and are forward-only; the fused,
hipGraph-captured training / inference programs live in `tecogan_amd.engine` / `tecogan_amd.infer` and are
what `lib.Teco.TecoGAN`, `lib.Teco.FRVSR` and `main.py` drive.  There is no CPU path: a missing HIP
library or a CPU tensor raises.
"""
import contextlib
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from tecogan_amd import kernels as K
from tecogan_amd._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, TG_F32, TecoHipError)
from tecogan_amd.params import pad8

# ------------------------------------------------------------------------------------------------
# variable-scope emulation (tf.variable_scope / slim naming)
# ------------------------------------------------------------------------------------------------
_VARS = OrderedDict()          # full TF name -> fp32 CUDA tensor in TF layout
_SCOPE = []                    # stack of (name, reuse)
_SEED = [1234]


def reset_default_graph(seed=1234):
    _VARS.clear()
    del _SCOPE[:]
    _SEED[0] = seed


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    inherited = _SCOPE[-1][1] if _SCOPE else False
    _SCOPE.append((name, inherited if reuse is None else reuse))
    try:
        yield
    finally:
        _SCOPE.pop()


def _full(name):
    return "/".join([s for s, _ in _SCOPE] + [name])


def global_variables():
    return _VARS


def get_variable(name, shape, initializer="xavier", device="cuda"):
    full = _full(name)
    reuse = _SCOPE[-1][1] if _SCOPE else False
    if full in _VARS:
        if not reuse:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % full)
        return _VARS[full]
    if reuse:
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % full)
    if initializer == "zeros":
        v = torch.zeros(shape)
    elif initializer == "ones":
        v = torch.ones(shape)
    else:                                                     # tf.contrib.layers.xavier_initializer (uniform)
        g = torch.Generator().manual_seed(_SEED[0] + len(_VARS))
        if len(shape) == 4:
            rf = shape[0] * shape[1]
            fan_in, fan_out = rf * shape[2], rf * shape[3]
        else:
            fan_in, fan_out = shape[0], shape[1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        v = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul(lim).float()
    _VARS[full] = v.to(device)
    return _VARS[full]


def _need_cuda(x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise TecoHipError("lib.ops works on CUDA tensors (NHWC fp32): there is no CPU path")
    return x.float().contiguous()


def _pad_channels(x, cpad):
    if x.shape[-1] == cpad:
        return x
    return K.concat2_pad(x, None, torch.empty(*x.shape[:-1], cpad, device=x.device))


# ------------------------------------------------------------------------------------------------
# value ranges (reference lib/ops.py:13-32)
# ------------------------------------------------------------------------------------------------
def preprocess(image):
    return image * 2 - 1            # [0, 1] => [-1, 1]


def deprocess(image):
    return (image + 1) / 2          # [-1, 1] => [0, 1]


def preprocessLR(image):
    return image


def deprocessLR(image):
    return image


# ------------------------------------------------------------------------------------------------
# convolutions (reference lib/ops.py:35-56)
# ------------------------------------------------------------------------------------------------
def _conv_forward(x, w_operand, bias, desc_args, cout, act=ACT_NONE, alpha=0.0):
    N, Hin, Win, Cin, Hout, Wout, k, stride, pt, pl, mode = desc_args
    out = torch.empty(N, Hout, Wout, cout, device=x.device)
    d = K.conv_desc(N, Hin, Win, Cin, Hout, Wout, cout, k, k, stride, pt, pl, mode, TG_F32, TG_F32, act, alpha)
    K.conv_forward(d, x, w_operand, bias, None, None, out)
    return out


def conv2(batch_input, kernel=3, output_channel=64, stride=1, use_bias=True, scope='conv'):
    """slim.conv2d(kernel, stride, 'SAME', NHWC, activation_fn=None, xavier)."""
    x = _need_cuda(batch_input)
    N, H, W, Cin = x.shape
    with variable_scope(scope):
        with variable_scope("Conv"):
            w = get_variable("weights", (kernel, kernel, Cin, output_channel))
            b = get_variable("biases", (output_channel,), "zeros") if use_bias else None
    cp = pad8(Cin)
    wt = torch.zeros(kernel * kernel, output_channel, cp, device=x.device)
    wt[:, :, :Cin] = w.permute(0, 1, 3, 2).reshape(kernel * kernel, output_channel, Cin)
    Ho, pt = K.same_pad(H, kernel, stride)
    Wo, pl = K.same_pad(W, kernel, stride)
    return _conv_forward(_pad_channels(x, cp), wt, b, (N, H, W, cp, Ho, Wo, kernel, stride, pt, pl, 0), output_channel)


def conv2_tran(batch_input, kernel=3, output_channel=64, stride=1, use_bias=True, scope='conv'):
    """slim.conv2d_transpose(kernel, stride, 'SAME'); filter layout [kh,kw,Cout,Cin]."""
    x = _need_cuda(batch_input)
    N, H, W, Cin = x.shape
    with variable_scope(scope):
        with variable_scope("Conv2d_transpose"):
            w = get_variable("weights", (kernel, kernel, output_channel, Cin))
            b = get_variable("biases", (output_channel,), "zeros") if use_bias else None
    cp = pad8(Cin)
    wn = torch.zeros(kernel * kernel, output_channel, cp, device=x.device)
    wn[:, :, :Cin] = w.reshape(kernel * kernel, output_channel, Cin)
    _, pt = K.same_pad(H * stride, kernel, stride)           # pad_before of the forward conv this op transposes
    _, pl = K.same_pad(W * stride, kernel, stride)
    return _conv_forward(_pad_channels(x, cp), wn, b, (N, H, W, cp, H * stride, W * stride, kernel, stride, pt, pl, 1),
                         output_channel)


def conv2_NCHW(*a, **k):
    raise NotImplementedError("conv2_NCHW is dead code in the reference (never called); use conv2 (NHWC)")


def prelu_tf(inputs, name='Prelu'):
    raise NotImplementedError("prelu_tf is dead code in the reference (never called)")


def lrelu(inputs, alpha):
    x = _need_cuda(inputs)
    return torch.where(x > 0, x, x * alpha)


def batchnorm(inputs, is_training):
    """slim.batch_norm(decay=.9, eps=1e-3, scale=False); the path always calls it with is_training=True."""
    if not is_training:
        raise NotImplementedError("the reference path only uses batchnorm(is_training=True) (lib/Teco.py:38)")
    x = _need_cuda(inputs)
    Cn = x.shape[-1]
    with variable_scope("BatchNorm"):
        beta = get_variable("beta", (Cn,), "zeros")
        mm = get_variable("moving_mean", (Cn,), "zeros")
        mv = get_variable("moving_variance", (Cn,), "ones")
    y = torch.empty_like(x)
    stats = torch.empty(2, Cn, device=x.device)
    moving = torch.stack((mm, mv)).contiguous()
    K.bn_lrelu_forward(x, y, beta, 1e-3, 1.0, stats, moving)          # alpha=1: plain BN (LeakyReLU(1) = id)
    mm.copy_(moving[0])
    mv.copy_(moving[1])
    return y


def maxpool(inputs, scope='maxpool'):
    x = _need_cuda(inputs)
    N, H, W, Cn = x.shape
    return K.maxpool2_forward(x, torch.empty(N, H // 2, W // 2, Cn, device=x.device))


def denselayer(inputs, output_size):
    """tf.layers.Dense(output_size) on the last axis (kernel + bias)."""
    x = _need_cuda(inputs)
    Cin = x.shape[-1]
    with variable_scope("dense"):
        kern = get_variable("kernel", (Cin, output_size))
        bias = get_variable("bias", (output_size,), "zeros")
    lead = x.shape[:-1]
    x4 = x.reshape(-1, 1, 1, Cin)
    cp = pad8(Cin)
    wt = torch.zeros(1, output_size, cp, device=x.device)
    wt[0, :, :Cin] = kern.t()
    out = _conv_forward(_pad_channels(x4, cp), wt, bias, (x4.shape[0], 1, 1, cp, 1, 1, 1, 1, 0, 0, 0), output_size)
    return out.reshape(*lead, output_size)


def pixelShuffler(inputs, scale=2):
    raise NotImplementedError("pixelShuffler is dead code in the reference (never called)")


def phaseShift(inputs, scale, shape_1, shape_2):
    raise NotImplementedError("phaseShift is dead code in the reference (never called)")


def upscale_four(inputs, scope='upscale_four'):
    """Fixed-ratio bilinear x4 (== legacy tf.image.resize_bilinear)."""
    x = _need_cuda(inputs)
    B, h, w, Cn = x.shape
    return K.upscale4_forward(x, torch.empty(B, 4 * h, 4 * w, Cn, device=x.device), 1.0)


def bicubic_four(inputs, scope='bicubic_four'):
    """Keys(-0.75) bicubic x4 == tf.image.resize_bicubic for API <= 1.13 (3-channel images)."""
    x = _need_cuda(inputs)
    B, h, w, Cn = x.shape
    if Cn != 3:
        raise ValueError("bicubic_four: the HIP kernel handles RGB (3-channel) images, got %d channels" % Cn)
    zero = torch.zeros(B, 4 * h, 4 * w, 3, device=x.device)
    y = K.bicubic_add_preprocess(zero, _pad_channels(x, 8), torch.empty_like(zero))     # (0 + bicubic)*2 - 1
    return (y + 1) / 2


def dense_image_warp(image, flow):
    """tf.contrib.image.dense_image_warp (called directly by the reference at lib/Teco.py:120,140,224,254)."""
    img, fl = _need_cuda(image), _need_cuda(flow)
    return K.warp_forward(img, fl, torch.empty_like(img))


def space_to_depth(x, block=4):
    """tf.space_to_depth (main.py:201): channel = (dy*4+dx)*C + c."""
    B, H, W, Cn = x.shape
    return x.reshape(B, H // block, block, W // block, block, Cn).permute(0, 1, 3, 2, 4, 5).reshape(
        B, H // block, W // block, block * block * Cn)


def random_flip_batch(input, decision):
    return torch.where((decision < 0.5).view(-1, 1, 1, 1), input.flip(2), input)


def random_flip(input, decision):
    return input.flip(1) if float(decision) < 0.5 else input


def print_configuration_op(FLAGS):
    print('[Configurations]:')
    for name, value in sorted(vars(FLAGS).items()):
        print('\t%s: %s' % (name, str(value)))
    print('End of configuration')


def copy_update_configuration(FLAGS, updateDict={}):
    from types import SimpleNamespace
    d = dict(vars(FLAGS))
    d.update(updateDict)
    return SimpleNamespace(**d)


def compute_psnr(ref, target):
    diff = target.float() - ref.float()
    mse = (diff * diff).mean()
    return 10.0 * torch.log10(255.0 * 255.0 / mse)


# ------------------------------------------------------------------------------------------------
# VGG-19 feature extractor (reference lib/ops.py:287-334)
# ------------------------------------------------------------------------------------------------
def vgg_arg_scope(weight_decay=0.0005):
    return None         # defined but never applied in the reference


def vgg_19(inputs, num_classes=1000, is_training=False, dropout_keep_prob=0.5, spatial_squeeze=True, scope='vgg_19',
           reuse=False, fc_conv_padding='VALID'):
    """Returns (net after pool5, end_points dict keyed '<scope>/convB/convB_j' and '<scope>/poolB')."""
    net = _need_cuda(inputs)
    end_points = OrderedDict()
    cfg = [(1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512)]
    with variable_scope(scope, reuse=reuse):
        for blk, reps, cout in cfg:
            with variable_scope("conv%d" % blk):
                for j in range(1, reps + 1):
                    # slim.repeat names the layers conv<b>/conv<b>_<j>/{weights,biases} (no extra 'Conv' level)
                    Cin = net.shape[-1]
                    with variable_scope("conv%d_%d" % (blk, j)):
                        w = get_variable("weights", (3, 3, Cin, cout))
                        b = get_variable("biases", (cout,), "zeros")
                    cp = pad8(Cin)
                    wt = torch.zeros(9, cout, cp, device=net.device)
                    wt[:, :, :Cin] = w.permute(0, 1, 3, 2).reshape(9, cout, Cin)
                    N, H, W, _ = net.shape
                    net = _conv_forward(_pad_channels(net, cp), wt, b, (N, H, W, cp, H, W, 3, 1, 1, 1, 0), cout, ACT_RELU)
                    end_points["%s/conv%d/conv%d_%d" % (scope, blk, blk, j)] = net
            net = maxpool(net)
            end_points["%s/pool%d" % (scope, blk)] = net
    return net, end_points


# ------------------------------------------------------------------------------------------------
# data helpers (reference lib/ops.py:339-367) -- the step before the path, also on the MFMA conv engine
# ------------------------------------------------------------------------------------------------
def gaussian_2dkernel(size=5, sig=1.):
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0       # == scipy.signal.gaussian(size, std=sig)
    g = np.exp(-0.5 * (x / sig) ** 2).reshape(size, 1)
    k = np.outer(g, g)
    return k / k.sum()


def tf_data_gaussDownby4(HRdata, sigma=1.5):
    """9x9 (sigma 1.5) Gaussian blur + stride-4 VALID down-sampling of RGB frames."""
    x = _need_cuda(HRdata)
    N, H, W, Cn = x.shape
    if Cn != 3:
        raise ValueError("tf_data_gaussDownby4 only works for RGB images")
    k_w = 1 + 2 * int(sigma * 3.0)
    gk = np.float32(gaussian_2dkernel(k_w, sigma)).reshape(-1)
    Ho, Wo = (H - k_w) // 4 + 1, (W - k_w) // 4 + 1
    # one depthwise HIP kernel (3 channels: memory-bound, no MFMA); the loader's fused form also crops + preprocesses the
    # target in the same launch (gauss_down_crop_preprocess below)
    return K.gauss_down4_preprocess(x.float().contiguous(), gk, torch.empty(N, Ho, Wo, 3, device=x.device))


def gauss_down_crop_preprocess(HRdata, sigma=1.5):
    """The training loader's GPU data step in ONE launch (reference lib/dataloader.py:306-332): HR crop [N,H,W,3] in [0,1]
    with a `border = int(1.5*3)` blur margin -> (preprocessLR(gauss_down4(HR)), preprocess(HR[border:-border]))."""
    x = _need_cuda(HRdata).float().contiguous()
    N, H, W, _ = x.shape
    k_w = 1 + 2 * int(sigma * 3.0)
    border = int(sigma * 3.0)
    gk = np.float32(gaussian_2dkernel(k_w, sigma)).reshape(-1)
    Ho, Wo = (H - k_w) // 4 + 1, (W - k_w) // 4 + 1
    lr = torch.empty(N, Ho, Wo, 3, device=x.device)
    tgt = torch.empty(N, 4 * Ho, 4 * Wo, 3, device=x.device)
    K.gauss_down4_preprocess(x, gk, lr, tgt, border)
    return lr, tgt


# ------------------------------------------------------------------------------------------------
# checkpoints / images (reference lib/ops.py:370-391, 521-523)
# ------------------------------------------------------------------------------------------------
def get_existing_from_ckpt(ckpt, var_list=None, rest_zero=False, print_level=1):
    """ckpt: a torch checkpoint holding {'variables': {tf_name: tensor}} (main.py save format) or the prefix of a
    TensorFlow tensor-bundle checkpoint (`<ckpt>.index`, as the reference reads through NewCheckpointReader).
    Returns a list of (variable_tensor, value) assignments like the reference's assign ops; shape mismatches
    raise ValueError (reference lib/ops.py:381-383); `rest_zero` zero-fills variables absent from the file."""
    from tecogan_amd.checkpoint import load_variables
    saved, _ = load_variables(ckpt)
    var_list = _VARS if var_list is None else var_list
    ops = []
    for name, var in var_list.items():
        if name in saved:
            if tuple(saved[name].shape) != tuple(var.shape):
                raise ValueError('Shape mismatch for var %s: ckpt %s vs graph %s' %
                                 (name, tuple(saved[name].shape), tuple(var.shape)))
            ops.append((var, saved[name]))
            if print_level > 1:
                print('loading %s' % name)
        elif rest_zero:
            ops.append((var, torch.zeros_like(var)))
            if print_level:
                print('Zero-filled: %s' % name)
        elif print_level:
            print('Not in checkpoint, kept: %s' % name)
    return ops


def save_img(out_path, img):
    """clip(img*255, 0, 255).astype(uint8) -- truncation, like the reference -- written as RGB by PIL.  Device tensors are
    converted on the GPU (tg_frame_to_u8) so only a quarter of the bytes crosses PCIe; the streaming inference loop uses
    the asynchronous tecogan_amd.output.FrameWriter instead."""
    from PIL import Image
    if isinstance(img, torch.Tensor) and img.is_cuda:
        u8 = K.frame_to_u8(img.detach().float().contiguous(), torch.empty(img.shape, dtype=torch.uint8, device=img.device))
        arr = u8.cpu().numpy()
    else:
        if isinstance(img, torch.Tensor):
            img = img.detach().float().cpu().numpy()
        arr = np.clip(img * 255.0, 0, 255).astype(np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    Image.fromarray(arr).save(out_path)


def gif_summary(*a, **k):
    return None          # TensorBoard gif summaries are out of scope (ffmpeg pipe)

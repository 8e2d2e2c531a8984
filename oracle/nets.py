"""Oracle networks: fnet, generator_F, discriminator_F, VGG-19 (torch CPU, NHWC).

TEST INFRASTRUCTURE ONLY.  Parameters are plain dicts keyed by the TF variable
names of the reference graph so they interchange with the
product's state and with a future TF-bundle reader.
"""
import math
from collections import OrderedDict

import torch

from . import ops as O

# --------------------------------------------------------------------------- #
# parameter specs (name -> shape), TF layouts
# --------------------------------------------------------------------------- #
FNET_BLOCKS = [("encoder_1", 6, 32), ("encoder_2", 32, 64), ("encoder_3", 64, 128),
               ("decoder_1", 128, 256), ("decoder_2", 256, 128), ("decoder_3", 128, 64)]


def fnet_spec():
    """lib/frvsr.py:4-41 variable list."""
    s = OrderedDict()
    p = "fnet/autoencode_unit/"
    for name, cin, cout in FNET_BLOCKS:
        s[p + name + "/conv_1/Conv/weights"] = (3, 3, cin, cout)
        s[p + name + "/conv_1/Conv/biases"] = (cout,)
        s[p + name + "/conv_2/Conv/weights"] = (3, 3, cout, cout)
        s[p + name + "/conv_2/Conv/biases"] = (cout,)
    s[p + "output_stage/conv1/Conv/weights"] = (3, 3, 64, 32)
    s[p + "output_stage/conv1/Conv/biases"] = (32,)
    s[p + "output_stage/conv2/Conv/weights"] = (3, 3, 32, 2)
    s[p + "output_stage/conv2/Conv/biases"] = (2,)
    return s


def generator_spec(num_resblock, cin=51, cout=3):
    """lib/frvsr.py:44-88 variable list."""
    s = OrderedDict()
    p = "generator/generator_unit/"
    s[p + "input_stage/conv/Conv/weights"] = (3, 3, cin, 64)
    s[p + "input_stage/conv/Conv/biases"] = (64,)
    for i in range(1, num_resblock + 1):
        for j in (1, 2):
            s[p + "resblock_%d/conv_%d/Conv/weights" % (i, j)] = (3, 3, 64, 64)
            s[p + "resblock_%d/conv_%d/Conv/biases" % (i, j)] = (64,)
    for j in (1, 2):
        # conv2d_transpose filter layout [kh,kw,Cout,Cin]
        s[p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/weights" % j] = (3, 3, 64, 64)
        s[p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/biases" % j] = (64,)
    s[p + "output_stage/conv/Conv/weights"] = (3, 3, 64, cout)
    s[p + "output_stage/conv/Conv/biases"] = (cout,)
    return s


DIS_BLOCKS = [("disblock_1", 64, 64), ("disblock_3", 64, 64), ("disblock_5", 64, 128), ("disblock_7", 128, 256)]


def discriminator_spec(cin=27):
    """lib/Teco.py:30-74 trainable variable list (moving stats are state, not listed)."""
    s = OrderedDict()
    p = "tdiscriminator/discriminator_unit/"
    s[p + "input_stage/conv/Conv/weights"] = (3, 3, cin, 64)
    s[p + "input_stage/conv/Conv/biases"] = (64,)
    for name, ci, co in DIS_BLOCKS:
        s[p + name + "/conv1/Conv/weights"] = (4, 4, ci, co)
        s[p + name + "/BatchNorm/beta"] = (co,)
    s[p + "dense_layer_2/dense/kernel"] = (256, 1)
    s[p + "dense_layer_2/dense/bias"] = (1,)
    return s


VGG_CFG = [(1, 2, 3, 64), (2, 2, 64, 128), (3, 4, 128, 256), (4, 4, 256, 512), (5, 4, 512, 512)]


def vgg_spec():
    """lib/ops.py:319-327 variable list."""
    s = OrderedDict()
    for blk, reps, cin, cout in VGG_CFG:
        for j in range(1, reps + 1):
            ci = cin if j == 1 else cout
            s["vgg_19/conv%d/conv%d_%d/weights" % (blk, blk, j)] = (3, 3, ci, cout)
            s["vgg_19/conv%d/conv%d_%d/biases" % (blk, blk, j)] = (cout,)
    return s


def init_params(spec, seed, dtype=torch.float32, vgg_he=False):
    """Seeded xavier-uniform (lib/ops.py:40,52) for weights/kernels, zeros for biases/beta.

    limit = sqrt(6/(fan_in+fan_out)), fan = k*k*C for conv, plain for dense.
    `vgg_he`: He-normal stand-in for the (absent) pretrained vgg_19.ckpt.
    Drawn in float64 from a torch.Generator so oracle and product can share it.
    """
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in spec.items():
        if len(shape) == 1:
            out[name] = torch.zeros(shape, dtype=dtype)
            continue
        if len(shape) == 4:
            rf = shape[0] * shape[1]
            fan_in, fan_out = rf * shape[2], rf * shape[3]
            if "Conv2d_transpose" in name:        # [kh,kw,Cout,Cin]
                fan_in, fan_out = rf * shape[3], rf * shape[2]
        else:
            fan_in, fan_out = shape
        if vgg_he:
            w = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fan_in)
        else:
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim
        out[name] = w.to(dtype)
    return out


# --------------------------------------------------------------------------- #
# networks
# --------------------------------------------------------------------------- #
def fnet(P, x):
    """lib/frvsr.py:4-41.  x [N,h,w,6] -> flow [N,h,w,2] (LR pixels, |.|<=24)."""
    p = "fnet/autoencode_unit/"

    def two(net, scope):
        net = O.conv2(net, P[p + scope + "/conv_1/Conv/weights"], P[p + scope + "/conv_1/Conv/biases"])
        net = O.lrelu(net, 0.2)
        net = O.conv2(net, P[p + scope + "/conv_2/Conv/weights"], P[p + scope + "/conv_2/Conv/biases"])
        return O.lrelu(net, 0.2)

    net = x
    for scope in ("encoder_1", "encoder_2", "encoder_3"):          # frvsr.py:5-13
        net = O.maxpool(two(net, scope))
    for scope in ("decoder_1", "decoder_2", "decoder_3"):          # frvsr.py:15-24
        net = O.upsample2_legacy(two(net, scope))
    net = O.conv2(net, P[p + "output_stage/conv1/Conv/weights"], P[p + "output_stage/conv1/Conv/biases"])
    net = O.lrelu(net, 0.2)
    net = O.conv2(net, P[p + "output_stage/conv2/Conv/weights"], P[p + "output_stage/conv2/Conv/biases"])
    return torch.tanh(net) * 24.0                                   # frvsr.py:39


def generator_F(P, x, num_resblock):
    """lib/frvsr.py:44-88.  x [N,h,w,51] -> HR [N,4h,4w,3] in [-1,1]."""
    p = "generator/generator_unit/"
    net = O.relu(O.conv2(x, P[p + "input_stage/conv/Conv/weights"], P[p + "input_stage/conv/Conv/biases"]))
    for i in range(1, num_resblock + 1):                            # frvsr.py:68-70
        s = p + "resblock_%d/" % i
        r = O.relu(O.conv2(net, P[s + "conv_1/Conv/weights"], P[s + "conv_1/Conv/biases"]))
        r = O.conv2(r, P[s + "conv_2/Conv/weights"], P[s + "conv_2/Conv/biases"])
        net = r + net
    for j in (1, 2):                                                # frvsr.py:72-77
        s = p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/" % j
        net = O.relu(O.conv2_tran(net, P[s + "weights"], P[s + "biases"], 2))
    net = O.conv2(net, P[p + "output_stage/conv/Conv/weights"], P[p + "output_stage/conv/Conv/biases"])
    net = net + O.bicubic_four(x[..., 0:3])                         # frvsr.py:81-86
    return O.preprocess(net)                                        # frvsr.py:87


def discriminator_F(P, x, bn_state=None):
    """lib/Teco.py:30-74.  x [tb,H,W,27] -> (prob [tb,H/16,W/16,1], [4 layer maps]).

    bn_state: optional dict updated in place with moving stats (decay 0.9, A.7).
    """
    p = "tdiscriminator/discriminator_unit/"
    net = O.lrelu(O.conv2(x, P[p + "input_stage/conv/Conv/weights"], P[p + "input_stage/conv/Conv/biases"]), 0.2)
    layers = []
    for name, _, _ in DIS_BLOCKS:
        net = O.conv2(net, P[p + name + "/conv1/Conv/weights"], None, 2)
        net, mean, var = O.batchnorm(net, P[p + name + "/BatchNorm/beta"])
        if bn_state is not None:
            mm, mv = p + name + "/BatchNorm/moving_mean", p + name + "/BatchNorm/moving_variance"
            n = net.shape[0] * net.shape[1] * net.shape[2]
            with torch.no_grad():   # [TF1] fused BN feeds the UNBIASED variance to the moving average
                bn_state[mm] = bn_state[mm] * 0.9 + mean * 0.1
                bn_state[mv] = bn_state[mv] * 0.9 + var * (n / max(n - 1, 1)) * 0.1
        net = O.lrelu(net, 0.2)
        layers.append(net)
    net = O.denselayer(net, P[p + "dense_layer_2/dense/kernel"], P[p + "dense_layer_2/dense/bias"])
    return torch.sigmoid(net), layers


VGG_TAPS = ("vgg_19/conv2/conv2_2", "vgg_19/conv3/conv3_4", "vgg_19/conv4/conv4_4", "vgg_19/conv5/conv5_4")


def vgg19_features(P, x, taps=VGG_TAPS, norm=True):
    """lib/Teco.py:5-24 + lib/ops.py:287-334: x in [-1,1] -> dict of (normalised) post-ReLU taps."""
    net = O.vgg_preprocess(x)
    out = {}
    for blk, reps, _, _ in VGG_CFG:
        for j in range(1, reps + 1):
            key = "vgg_19/conv%d/conv%d_%d" % (blk, blk, j)
            net = O.relu(O.conv2(net, P[key + "/weights"], P[key + "/biases"]))
            if key in taps:
                out[key] = O.vgg_norm(net) if norm else net
        if key == taps[-1]:
            break
        net = O.maxpool(net)
    return out

"""Drop-in for the reference's `lib/Teco.py`: VGG19_slim, discriminator_F, TecoGAN, FRVSR.

`TecoGAN(r_inputs, r_targets, FLAGS, GAN_Flag)` builds the hipGraph-captured training program
(`tecogan_amd.engine.TrainEngine`) and returns the reference's `Network` namedtuple.  Where the reference
puts graph tensors / ops into the tuple (to be fetched with `sess.run`), this returns zero-argument
callables with the same meaning: `net.train()` runs one optimisation step (== sess.run(Net.train)),
`net.gen_output()` returns s_gen_output of the last step, `net.update_list()` the raw loss values, ...
"""
import collections

import torch

from lib import ops as _ops
from lib.frvsr import *  # noqa: F401,F403
from lib.frvsr import _get_net
from tecogan_amd import kernels as K
from tecogan_amd.engine import TrainEngine
from tecogan_amd.nets import DIS_CPAD, Discriminator
from tecogan_amd.params import discriminator_spec

VGG_MEAN = [123.68, 116.78, 103.94]

Network = collections.namedtuple('Network', 'gen_output, train, learning_rate, update_list, '
                                            'update_list_name, update_list_avg, image_summary, global_step')


def VGG19_slim(input, reuse, deep_list=None, norm_flag=True):
    """reference lib/Teco.py:5-24: VGG preprocessing + (per-pixel L2-normalised) feature maps."""
    x = _ops._need_cuda(input)
    img = _ops.deprocess(x) * 255.0 - torch.tensor(VGG_MEAN, device=x.device)
    _, output = _ops.vgg_19(img, is_training=False, reuse=reuse)
    results = {}
    for key, feat in output.items():
        if deep_list is None or key in deep_list:
            if norm_flag:
                feat = feat / torch.sqrt((feat * feat).sum(dim=3, keepdim=True) + 1e-12)
            results[key] = feat
    return results


def discriminator_F(dis_inputs, FLAGS=None):
    """reference lib/Teco.py:30-74: returns (sigmoid map [tb,H/16,W/16,1], [4 layer feature maps])."""
    if FLAGS is None:
        raise ValueError('No FLAGS is provided for generator')
    x = _ops._need_cuda(dis_inputs)
    if x.shape[-1] != 27:
        raise ValueError("discriminator_F: the HIP path is built for the 27-channel spatio-temporal input")
    reuse = _ops._SCOPE[-1][1] if _ops._SCOPE else False
    ps, net = _get_net("discriminator_unit", "tdiscriminator", discriminator_spec(), Discriminator, reuse, x.device)
    xin = K.concat2_pad(x, None, torch.empty(*x.shape[:-1], DIS_CPAD, device=x.device))
    prob, layers, _ = net.forward(xin, keep=False)
    return prob, layers


class _Averager:
    """tf.train.ExponentialMovingAverage(0.99) over the loss list (reference lib/Teco.py:433-435), no debias."""

    def __init__(self):
        self.shadow = None

    def update(self, vals):
        if self.shadow is None:
            self.shadow = [0.0] * len(vals)
        self.shadow = [s - 0.01 * (s - v) for s, v in zip(self.shadow, vals)]
        return self.shadow


def TecoGAN(r_inputs, r_targets, FLAGS, GAN_Flag=True, act_dtype=None, process_group=None):
    """reference lib/Teco.py:77-517.  r_inputs [B,RNN_N,h,w,3] in [0,1]; r_targets [B,RNN_N,4h,4w,3] in [-1,1]
    (CUDA tensors; they are the step's input buffers: refill them in place, or pass a new batch to train())."""
    if act_dtype is None:
        act_dtype = torch.bfloat16 if getattr(FLAGS, "act_dtype", "bf16") == "bf16" else torch.float32
    dev = r_inputs.device
    eng = TrainEngine(FLAGS, dev, gan=GAN_Flag, act_dtype=act_dtype, seed=getattr(FLAGS, "rand_seed", 1) + 41,
                      process_group=process_group)
    eng.set_batch(r_inputs, r_targets)
    avg = _Averager()
    state = {"names": None, "vals": None, "avg": None}

    def train(inputs=None, targets=None):
        eng.step(inputs, targets)
        return None

    def refresh():
        L = eng.losses()
        names = [k for k in L if k not in ("t_balance", "t_balance_now")]
        vals = [L[k] for k in names]
        state["names"], state["vals"] = names, vals
        state["avg"] = avg.update(vals)
        return L

    def update_list():
        refresh()
        return state["vals"]

    def update_list_avg():
        L = refresh()
        extra = []
        if GAN_Flag:                                   # lib/Teco.py:451-452,495-496
            extra = [L["t_balance"], min(FLAGS.Dt_ratio_max, FLAGS.Dt_ratio_0 + FLAGS.Dt_ratio_add * eng.global_step()),
                     int(eng.sched[8].item()), eng.global_step() - int(eng.sched[8].item())]
        return state["avg"] + extra

    def update_list_name():
        if state["names"] is None:
            refresh()
        extra = ["t_balance", "Dst_ratio", "withD_counter", "w_o_D_counter"] if GAN_Flag else []
        return state["names"] + extra

    def gen_output():                                   # s_gen_output: [B*T,H,W,3], (b, t) order like the reference
        g = eng.gen
        return g.transpose(0, 1).reshape(-1, *g.shape[2:])

    net = Network(gen_output=gen_output, train=train,
                  learning_rate=lambda: float(eng.hyper[-1, 5].item()) or FLAGS.learning_rate,
                  update_list=update_list, update_list_name=update_list_name, update_list_avg=update_list_avg,
                  image_summary=None, global_step=eng.global_step)
    net.train.engine = eng                              # escape hatch for checkpointing / tests
    return net


def FRVSR(r_inputs, r_targets, FLAGS, **kw):
    """reference lib/Teco.py:521-522."""
    return TecoGAN(r_inputs, r_targets, FLAGS, False, **kw)

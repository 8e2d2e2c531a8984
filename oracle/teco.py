"""Oracle training / inference steps: restatement of lib/Teco.py:77-522 and main.py:186-260.

TEST INFRASTRUCTURE ONLY.  torch-CPU autograd plays the role of tf.gradients; the three
TF-Adams, the EMA balance and the tf.cond D-gate follow SURVEY.md Appendix A.10/A.11.
"""
import math
from collections import OrderedDict
from types import SimpleNamespace

import torch

from . import nets as NN
from . import ops as O


def default_flags(**kw):
    """main.py:30-103 defaults, then the runGan.py case-3 overrides unless told otherwise."""
    f = dict(RNN_N=10, batch_size=4, crop_size=32, num_resblock=16, pingpang=True, pp_scaling=0.5,
             vgg_scaling=0.2, warp_scaling=1.0, EPS=1e-12, learning_rate=5e-5, decay_step=500000,
             decay_rate=1.0, stair=False, beta=0.9, adameps=1e-8, ratio=0.01, Dt_mergeDs=True,
             Dt_ratio_0=1.0, Dt_ratio_add=0.0, Dt_ratio_max=1.0, Dbalance=0.4, crop_dt=0.75,
             D_LAYERLOSS=True)
    f.update(kw)
    return SimpleNamespace(**f)


def frvsr_flags(**kw):
    """runGan.py:250-272 (case 4): no D, no ping-pong, 10 res blocks, VGG off."""
    base = dict(num_resblock=10, pingpang=False, ratio=-0.01, vgg_scaling=-0.002, learning_rate=5e-5, stair=True)
    base.update(kw)
    return default_flags(**base)


class State:
    """Weights + optimiser/EMA state of one training run (all torch CPU tensors)."""

    def __init__(self, flags, seed=42, gan=True, dtype=torch.float32):
        self.flags = flags
        self.gan = gan
        self.P = OrderedDict()
        self.P.update(NN.init_params(NN.generator_spec(flags.num_resblock), seed, dtype))
        self.P.update(NN.init_params(NN.fnet_spec(), seed + 1, dtype))
        if gan:
            # Dt_mergeDs=False: temporal-only D on the 9 warped channels (lib/Teco.py:246-250,269-272)
            self.P.update(NN.init_params(NN.discriminator_spec(27 if flags.Dt_mergeDs else 9), seed + 2, dtype))
        self.vgg = NN.init_params(NN.vgg_spec(), seed + 3, dtype, vgg_he=True) if flags.vgg_scaling > 0 else None
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.t = {"generator": 0, "fnet": 0, "tdiscriminator": 0}    # per-Adam step counts
        self.global_step = 0
        self.tb = 0.0                                                # EMA of t_balance (shadow from 0)
        self.bn = {}
        if gan:
            p = "tdiscriminator/discriminator_unit/"
            for name, _, co in NN.DIS_BLOCKS:
                self.bn[p + name + "/BatchNorm/moving_mean"] = torch.zeros(co, dtype=dtype)
                self.bn[p + name + "/BatchNorm/moving_variance"] = torch.ones(co, dtype=dtype)


def forward_losses(P, vggP, r_inputs, r_targets, F, gan, bn_state=None, global_step=0):
    """lib/Teco.py:77-417: everything up to the losses.  Returns a dict of tensors (graph attached)."""
    B = r_inputs.shape[0]
    if F.pingpang:                                               # Teco.py:80-85
        r_inputs = torch.cat((r_inputs, r_inputs[:, :-1].flip(1)), 1)
        r_targets = torch.cat((r_targets, r_targets[:, :-1].flip(1)), 1)
    T = r_inputs.shape[1]
    cs = F.crop_size
    H = cs * 4
    # --- FNet on all consecutive pairs (Teco.py:102-117)
    pre, cur = r_inputs[:, :-1], r_inputs[:, 1:]
    fnet_in = torch.cat((pre, cur), -1).reshape(B * (T - 1), cs, cs, 6)
    flow_lr = NN.fnet(P, fnet_in)
    gen_flow = O.upscale_four(flow_lr * 4.0).reshape(B, T - 1, H, H, 2)
    input_frames = cur.reshape(B * (T - 1), cs, cs, 3)
    s_input_warp = O.dense_image_warp(pre.reshape(B * (T - 1), cs, cs, 3), flow_lr)    # Teco.py:120-122
    # --- recurrent generator (Teco.py:125-155)
    x0 = torch.cat((r_inputs[:, 0], torch.zeros(B, cs, cs, 48, dtype=r_inputs.dtype)), -1)
    gen_pre = NN.generator_F(P, x0, F.num_resblock)
    outs, warps = [gen_pre], []
    for i in range(T - 1):
        w = O.dense_image_warp(gen_pre, gen_flow[:, i])
        warps.append(w)
        x = torch.cat((r_inputs[:, i + 1], O.space_to_depth4(O.preprocessLR(O.deprocess(w)))), -1)
        gen_pre = NN.generator_F(P, x, F.num_resblock)
        outs.append(gen_pre)
    gen_outputs = torch.stack(outs, 1)                           # [B,T,H,H,3]
    s_gen = gen_outputs.reshape(B * T, H, H, 3)
    s_tar = r_targets.reshape(B * T, H, H, 3)
    R = OrderedDict(gen_outputs=gen_outputs, flow_lr=flow_lr, gen_warppre=torch.stack(warps, 1) if warps else None)
    names, vals = [], []

    if F.vgg_scaling > 0:                                        # Teco.py:174-178
        gen_vgg = NN.vgg19_features(vggP, s_gen)
        with torch.no_grad():
            tar_vgg = NN.vgg19_features(vggP, s_tar)

    if gan:                                                      # Teco.py:180-272
        t_size = 3 * (T // 3)
        t_gen = gen_outputs[:, :t_size].reshape(B * t_size, H, H, 3)
        t_tar = r_targets[:, :t_size].reshape(B * t_size, H, H, 3)
        tb = B * t_size // 3
        if not F.pingpang:                                       # Teco.py:190-204
            back_in = torch.cat((r_inputs[:, 2:t_size:3], r_inputs[:, 1:t_size:3]), -1).reshape(tb, cs, cs, 6)
            flow_back = O.upscale_four(NN.fnet(P, back_in) * 4.0).reshape(B, t_size // 3, H, H, 2)
            v_pre, v_nxt = gen_flow[:, 0:t_size:3], flow_back
        else:                                                    # Teco.py:206-209
            v_pre = gen_flow[:, 0:t_size:3]
            idx = list(range(T - 1))[-2:-1 - t_size:-3]
            v_nxt = gen_flow[:, idx]
        T_vel = torch.stack((v_pre, torch.zeros_like(v_pre), v_nxt), 2).reshape(B * t_size, H, H, 2).detach()
        off = 0
        if F.crop_dt < 1.0:                                      # Teco.py:216-220
            csd = int(cs * 4 * F.crop_dt)
            off = (cs * 4 - csd) // 2
        t_input = O.pack_triplets(r_inputs[:, :t_size].reshape(B * t_size, cs, cs, 3), tb)
        input_hi = O.resize_bilinear_legacy(t_input, H, H)       # Teco.py:240-244

        def d_input(frames):                                     # Teco.py:224-245 / 254-269
            warped = O.pack_triplets(O.dense_image_warp(frames, T_vel), tb)
            if not F.Dt_mergeDs:                                 # Teco.py:231-232,249-250: cropped, NOT padded back
                return warped[:, off:H - off, off:H - off] if off else warped
            warped = O.crop_pad_dt(warped, off)
            return torch.cat((O.pack_triplets(frames, tb), warped, input_hi), -1)

        real_out, real_layers = NN.discriminator_F(P, d_input(t_tar), bn_state)
        fake_out, fake_layers = NN.discriminator_F(P, d_input(t_gen), bn_state)
        R.update(d_real=real_out, d_fake=fake_out)
        if F.D_LAYERLOSS:                                        # Teco.py:275-313
            norm = [12.0, 14.0, 24.0, 100.0]
            sum_layer = 0
            for i in range(4):
                ll = (real_layers[i] - fake_layers[i]).abs().sum(3).mean()
                names.append("D_layer_%d_loss" % i)
                vals.append(ll)
                sum_layer = sum_layer + 0.02 * ll / norm[i]
            names.append("D_layer_loss_sum")
            vals.append(sum_layer)

    content = ((s_gen - s_tar) ** 2).sum(3).mean()               # Teco.py:320-325
    names.append("l2_content_loss"); vals.append(content)
    gen_loss = content
    warp_loss = ((input_frames - s_input_warp) ** 2).sum(3).mean()   # Teco.py:329-333
    names.append("l2_warp_loss"); vals.append(warp_loss)
    if F.vgg_scaling > 0:                                        # Teco.py:339-359
        vgg_loss = 0
        for i, key in enumerate(NN.VGG_TAPS):
            cur_d = 1.0 - (gen_vgg[key] * tar_vgg[key]).sum(3).mean()
            names.append("vgg_loss_%d" % (i + 2)); vals.append(cur_d)
            vgg_loss = vgg_loss + cur_d
        gen_loss = gen_loss + F.vgg_scaling * vgg_loss
        names.append("vgg_all"); vals.append(vgg_loss)
    if F.pingpang:                                               # Teco.py:362-372
        first = gen_outputs[:, 0:F.RNN_N - 1]
        last_rev = gen_outputs[:, list(range(T))[-1:-F.RNN_N:-1]]
        pploss = (first - last_rev).abs().mean()
        if F.pp_scaling > 0:
            gen_loss = gen_loss + pploss * F.pp_scaling
        names.append("PingPang"); vals.append(pploss)
    if gan:                                                      # Teco.py:374-417
        t_adv = (-torch.log(fake_out + F.EPS)).mean()
        dt_ratio = min(F.Dt_ratio_max, F.Dt_ratio_0 + F.Dt_ratio_add * float(global_step))   # Teco.py:379-380
        gen_loss = gen_loss + F.ratio * t_adv * dt_ratio
        names.append("t_adversarial_loss"); vals.append(t_adv)
        if F.D_LAYERLOSS:
            gen_loss = gen_loss + sum_layer * dt_ratio
        d_fake_l = torch.log(1 - fake_out + F.EPS)
        d_real_l = torch.log(real_out + F.EPS)
        t_discrim = (-(d_fake_l + d_real_l)).mean()
        t_balance = d_real_l.mean() + t_adv
        names += ["t_discrim_loss", "t_discrim_real_output", "t_discrim_fake_output"]
        vals += [t_discrim, real_out.mean(), fake_out.mean()]
        R.update(discrim_loss=t_discrim, t_balance=t_balance)
    names.append("All_loss_Gen"); vals.append(gen_loss)
    R.update(gen_loss=gen_loss, warp_loss=warp_loss, fnet_loss=F.warp_scaling * warp_loss + gen_loss,
             names=names, vals=vals)
    return R


def _split(P):
    g = [k for k in P if k.startswith("generator/")]
    f = [k for k in P if k.startswith("fnet/")]
    d = [k for k in P if k.startswith("tdiscriminator/")]
    return g, f, d


def train_step(S, r_inputs, r_targets):
    """One `sess.run(Net.train)` (main.py:377-387) of TecoGAN()/FRVSR() (lib/Teco.py:419-517).

    Semantics fixed as SURVEY section 5 states: all gradients are taken from the PRE-update weights,
    then D (if the gate is open), G and fnet are applied.  Returns the dict of forward results plus
    `grads` (name -> tensor) and `with_D` (bool).
    """
    F = S.flags
    P = OrderedDict((k, v.detach().clone().requires_grad_()) for k, v in S.P.items())
    R = forward_losses(P, S.vgg, r_inputs, r_targets, F, S.gan, S.bn if S.gan else None, S.global_step)
    gk, fk, dk = _split(P)
    grads = {}
    gg = torch.autograd.grad(R["gen_loss"], [P[k] for k in gk], retain_graph=True)
    fg = torch.autograd.grad(R["fnet_loss"], [P[k] for k in fk], retain_graph=S.gan)
    grads.update(zip(gk, gg))
    grads.update(zip(fk, fg))
    with_d = False
    if S.gan:
        dg = torch.autograd.grad(R["discrim_loss"], [P[k] for k in dk])
        grads.update(zip(dk, dg))
        # gate on the OLD average, then update it (Teco.py:415-417,464,477,493-494)
        with_d = S.tb < F.Dbalance
        S.tb = O.ema_tf(S.tb, float(R["t_balance"]))
    lr = O.exponential_decay(F.learning_rate, S.global_step, F.decay_step, F.decay_rate, F.stair)

    def apply(keys, scope, lr_scope):
        S.t[scope] += 1
        for k in keys:
            O.adam_tf_step(S.P[k], grads[k], S.m[k], S.v[k], S.t[scope], lr_scope, F.beta, 0.999, F.adameps)

    with torch.no_grad():
        if with_d:
            apply(dk, "tdiscriminator", lr if F.Dt_mergeDs else lr * 0.3)      # Teco.py:422-425
        apply(gk, "generator", lr)
        apply(fk, "fnet", lr)
    S.global_step += 1
    R["grads"] = grads
    R["with_D"] = with_d
    R["lr"] = lr
    return R


# ------------------------------------------------------------------------------------------------
class InferenceState:
    """main.py:195-199: pre_inputs, pre_gen ([0,1]), pre_warp."""

    def __init__(self, h, w, dtype=torch.float32):
        self.pre_inputs = torch.zeros(1, h, w, 3, dtype=dtype)
        self.pre_gen = torch.zeros(1, 4 * h, 4 * w, 3, dtype=dtype)
        self.pre_warp = torch.zeros(1, 4 * h, 4 * w, 3, dtype=dtype)
        self.first = True


def inference_step(P, st, frame, num_resblock):
    """One iteration of the loop main.py:253-260: optional `before_ops`, then `outputs`."""
    h, w = frame.shape[1], frame.shape[2]
    with torch.no_grad():
        if not st.first:                                          # main.py:209-216
            oh, ow = h - h // 8 * 8, w - w // 8 * 8
            # fnet itself shrinks sizes that are not multiples of 8 (3 VALID pools, 3 x2 upsamples)
            flow = NN.fnet(P, torch.cat((st.pre_inputs, frame), -1))
            assert flow.shape[1] == h - oh and flow.shape[2] == w - ow
            flow = _pad_symmetric(flow, oh, ow)                       # main.py:188-190,212
            st.pre_warp = O.dense_image_warp(st.pre_gen, O.upscale_four(flow * 4.0))
        x = torch.cat((frame, O.space_to_depth4(st.pre_warp)), -1)        # main.py:201-202
        out = O.deprocess(NN.generator_F(P, x, num_resblock))           # main.py:204-207
        st.pre_inputs, st.pre_gen, st.first = frame, out, False
    return out


def _pad_symmetric(x, ph, pw):
    """tf.pad(x, [[0,0],[0,ph],[0,pw],[0,0]], 'SYMMETRIC') (main.py:190,212)."""
    if ph:
        x = torch.cat((x, x[:, x.shape[1] - ph:].flip(1)), 1)
    if pw:
        x = torch.cat((x, x[:, :, x.shape[2] - pw:].flip(2)), 2)
    return x


def save_img_u8(img):
    """lib/ops.py:521-523: clip(img*255, 0, 255).astype(uint8) (truncation), RGB kept here."""
    return (img * 255.0).clamp(0, 255).to(torch.uint8)

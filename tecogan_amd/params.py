"""Flat parameter store for the FRVSR/TecoGAN networks (MI355X layout).

All trainable parameters of a run live in ONE fp32 device buffer (`flat`), ordered by optimiser scope
(generator | fnet | tdiscriminator, reference lib/Teco.py:421,441-442), with matching flat gradient /
Adam-m / Adam-v buffers.  The gradient buffer is what the conv weight-gradient kernels accumulate
into (fp32 atomics), what the fused TF-Adam kernel consumes, and what RCCL all-reduces -- no per-tensor
copies anywhere.  Individual parameters are views keyed by the reference's TF variable names
 so checkpoints interchange.

The MFMA convolution engine reads weights as [tap][out][in] panels; `repack()` produces, in two
launches for the whole store, the transposed (`wT`) and natural (`wN`) compute copies in the
activation dtype with first-layer input channels zero-padded to a multiple of 8.
"""
import math
import os
from collections import OrderedDict

import torch

from . import kernels as K

SCOPES = ("generator", "fnet", "tdiscriminator")
K4S2_FRAG = True      # fragment-order copies of the discriminator's 4x4 stride-2 weights (csrc/conv4x4s2.hip; conv_igemm.hip without: 2.20 -> 2.01 ms of
                      # main-stream discriminator segments, profiles/r05f_seg_timeline.txt); a plain attribute, set by A/B tools

FNET_BLOCKS = [("encoder_1", 6, 32), ("encoder_2", 32, 64), ("encoder_3", 64, 128),
               ("decoder_1", 128, 256), ("decoder_2", 256, 128), ("decoder_3", 128, 64)]
DIS_BLOCKS = [("disblock_1", 64, 64), ("disblock_3", 64, 64), ("disblock_5", 64, 128), ("disblock_7", 128, 256)]
VGG_CFG = [(1, 2, 3, 64), (2, 2, 64, 128), (3, 4, 128, 256), (4, 4, 256, 512), (5, 4, 512, 512)]


def pad8(c):
    return (c + 7) // 8 * 8


def generator_spec(num_resblock, cin=51, cout=3):
    """Variables of generator_F (reference lib/frvsr.py:44-88)."""
    s, p = OrderedDict(), "generator/generator_unit/"
    s[p + "input_stage/conv/Conv/weights"] = (3, 3, cin, 64)
    s[p + "input_stage/conv/Conv/biases"] = (64,)
    for i in range(1, num_resblock + 1):
        for j in (1, 2):
            s[p + "resblock_%d/conv_%d/Conv/weights" % (i, j)] = (3, 3, 64, 64)
            s[p + "resblock_%d/conv_%d/Conv/biases" % (i, j)] = (64,)
    for j in (1, 2):
        s[p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/weights" % j] = (3, 3, 64, 64)
        s[p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/biases" % j] = (64,)
    s[p + "output_stage/conv/Conv/weights"] = (3, 3, 64, cout)
    s[p + "output_stage/conv/Conv/biases"] = (cout,)
    return s


def fnet_spec():
    """Variables of fnet (reference lib/frvsr.py:4-41)."""
    s, p = OrderedDict(), "fnet/autoencode_unit/"
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


def discriminator_spec(cin=27):
    """Trainable variables of discriminator_F (reference lib/Teco.py:30-74)."""
    s, p = OrderedDict(), "tdiscriminator/discriminator_unit/"
    s[p + "input_stage/conv/Conv/weights"] = (3, 3, cin, 64)
    s[p + "input_stage/conv/Conv/biases"] = (64,)
    for name, ci, co in DIS_BLOCKS:
        s[p + name + "/conv1/Conv/weights"] = (4, 4, ci, co)
        s[p + name + "/BatchNorm/beta"] = (co,)
    s[p + "dense_layer_2/dense/kernel"] = (256, 1)
    s[p + "dense_layer_2/dense/bias"] = (1,)
    return s


def vgg_spec():
    """Variables of vgg_19 up to conv5_4 (reference lib/ops.py:319-327)."""
    s = OrderedDict()
    for blk, reps, cin, cout in VGG_CFG:
        for j in range(1, reps + 1):
            s["vgg_19/conv%d/conv%d_%d/weights" % (blk, blk, j)] = (3, 3, cin if j == 1 else cout, cout)
            s["vgg_19/conv%d/conv%d_%d/biases" % (blk, blk, j)] = (cout,)
    return s


def init_values(spec, seed, he_normal=False):
    """Seeded xavier-uniform (reference lib/ops.py:40,52) / zeros; float64 draws from a torch.Generator
    in spec order, cast to fp32 -- the same stream the test oracle uses, so weights match bit for bit."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in spec.items():
        if len(shape) == 1:
            out[name] = torch.zeros(shape)
            continue
        if len(shape) == 4:
            rf = shape[0] * shape[1]
            fan_in, fan_out = rf * shape[2], rf * shape[3]
            if "Conv2d_transpose" in name:
                fan_in, fan_out = rf * shape[3], rf * shape[2]
        else:
            fan_in, fan_out = shape
        if he_normal:
            w = torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fan_in)
        else:
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim
        out[name] = w.float()
    return out


def damp_values(values, conv2=0.25, out=0.1):
    """Scaled copy of an xavier-initialised generator: res-block `conv_2` weights x `conv2`, output conv weights x `out`.
    With plain xavier weights the 4x recurrence (HR output -> warp -> space-to-depth -> next input) is expansive: the
    frame maximum doubles every frame (2 -> 1e5 over 18 frames) and fp32 rounding noise grows with it (an fp32 and an fp64
    run of the same CPU oracle differ by 1.6e-2 of the frame maximum at frame 18, tests/oracle_conditioning.py).  Damped
    this way the generator behaves like a trained one -- output = bicubic(LR) + small residual, frames stay in the image
    range -- which is the regime where a 1e-3 per-pixel comparison means something.  Used by the BASELINE-size parity
    tests and by bench.py's bf16-vs-fp32 error record; timing runs keep the reference's xavier init (lib/ops.py:40,52)."""
    out_v = OrderedDict()
    for k, v in values.items():
        if k.startswith("generator/") and "/resblock_" in k and k.endswith("conv_2/Conv/weights"):
            v = v * conv2
        elif k == "generator/generator_unit/output_stage/conv/Conv/weights":
            v = v * out
        out_v[k] = v
    return out_v


class ParamStore:
    """One flat fp32 buffer + compute copies.  `specs`: OrderedDict scope -> OrderedDict(name -> shape)."""

    def __init__(self, specs, device, act_dtype=torch.float32, trainable=True, bpad=(), wide_frag=False):
        """bpad: names of weights whose OUTPUT channels are also padded to a multiple of 8 in the natural copy
        (layers whose output-gradient is kept channel-padded, e.g. the 3-channel generator output conv).
        wide_frag: also keep fragment-order copies of the wide 3x3 layers for csrc/conv3x3_wr.hip (frozen stores: the
        copies are refreshed by repack() like the others, which a frozen store runs once per load())."""
        self.device, self.act_dtype, self.trainable = device, act_dtype, trainable
        self.wide_frag = bool(wide_frag) and act_dtype == torch.bfloat16
        self.wide = {}                                             # name -> [forward copy or None, input-gradient copy or None]
        self._k4 = None                                            # (flat buffer, table, count) of the 4x4 stride-2 copies
        self.entries = OrderedDict()
        self.scope_range = OrderedDict()
        off = poff = 0
        rows = []
        for scope, spec in specs.items():
            start = off
            for name, shape in spec.items():
                n = 1
                for d in shape:
                    n *= d
                e = dict(offset=off, shape=tuple(shape), numel=n, scope=scope, packed=None)
                if len(shape) in (2, 4):
                    shp = shape if len(shape) == 4 else (1, 1) + tuple(shape)
                    taps, A, Bd = shp[0] * shp[1], shp[2], shp[3]
                    Ap = pad8(A)
                    Bp = pad8(Bd) if name in bpad else Bd
                    e.update(packed=poff, taps=taps, A=A, B=Bd, Apad=Ap, Bpad=Bp, k=shp[0])
                    rows.append([off, poff, taps, A, Bd, Ap, Bp])
                    poff += (taps * Ap * Bp + 7) // 8 * 8          # keep every packed tensor 16-B aligned
                self.entries[name] = e
                off += n
            off = (off + 3) // 4 * 4                               # 16-B aligned scope boundaries
            self.scope_range[scope] = (start, off)
        self.numel, self.packed_numel = off, poff
        self.flat = torch.zeros(off, device=device)
        if trainable:
            self.grad = torch.zeros(off, device=device)
            self.m = torch.zeros(off, device=device)
            self.v = torch.zeros(off, device=device)
        self.wT = torch.zeros(max(poff, 8), device=device, dtype=act_dtype)
        self.wN = torch.zeros(max(poff, 8), device=device, dtype=act_dtype)
        self.table = torch.tensor(rows, dtype=torch.int64, device=device).contiguous()
        self.ntab = len(rows)
        # fragment-order copies of the generator's residual-block convs and transposed convs (64 -> 64, 3x3) for the
        # latency-regime kernels (csrc/resblock_lat.hip, hr_bwd_lat.hip; bf16 compute copies only): the [tap][out][in]-style
        # operand (dst_t: row = the tensor's LAST axis) in wTf, the [tap][in][out]-style one in wNf
        self.frag = OrderedDict()
        if act_dtype == torch.bfloat16:
            for name, e in self.entries.items():
                if ("/resblock_" in name or "/conv_tran" in name) and e.get("taps") == 9 and e["A"] == 64 and e["B"] == 64:
                    self.frag[name] = len(self.frag) * 36864
                elif name.startswith("generator/") and "/input_stage/" in name and e.get("taps") == 9 and e["A"] <= 64 and e["B"] == 64:
                    # the generator's input conv (51 -> 64): its copy has the input channels zero-padded to 64 (csrc/resblock_chain.hip
                    # runs it in front of the trunk's first block)
                    self.frag[name] = len(self.frag) * 36864
        nf = len(self.frag)
        self.wTf = torch.zeros(max(nf * 36864, 8), device=device, dtype=torch.bfloat16)
        self.wNf = torch.zeros(max(nf * 36864, 8), device=device, dtype=torch.bfloat16)
        self.frag_table = torch.tensor([[self.entries[n]["offset"], o, self.entries[n]["A"]] for n, o in self.frag.items()] or [[0, 0, 64]],
                                       dtype=torch.int64, device=device).contiguous()

    # ---- views ------------------------------------------------------------------------------
    def view(self, name, buf=None):
        e = self.entries[name]
        b = self.flat if buf is None else buf
        return b[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])

    def gview(self, name):
        return self.view(name, self.grad)

    def packed(self, name, transposed):
        """Compute copy of a weight: [tap][B][Apad] if transposed else [tap][Apad][Bpad] (flat view)."""
        e = self.entries[name]
        n = e["taps"] * e["Apad"] * (e["B"] if transposed else e["Bpad"])
        return (self.wT if transposed else self.wN)[e["packed"]:e["packed"] + n]

    def packed_wide(self, name, transposed):
        """Fragment-order copy of a wide 3x3 layer for tg_conv3x3_wide_frag (forward operand if transposed, else the
        input-gradient operand with mirrored taps), or None."""
        w = self.wide.get(name)
        return None if w is None else w[0 if transposed else 1]

    def packed_frag(self, name, transposed):
        """Fragment-order copy (csrc/resblock_lat.hip) of a residual-block conv, or None when the store keeps none."""
        o = self.frag.get(name)
        if o is None:
            return None
        return (self.wTf if transposed else self.wNf)[o:o + 36864]

    def scope_slice(self, scope, buf):
        a, b = self.scope_range[scope]
        return buf[a:b]

    # ---- state ------------------------------------------------------------------------------
    def load(self, values):
        """values: name -> CPU/GPU tensor with the TF shape."""
        for name, t in values.items():
            if name not in self.entries:
                raise KeyError("unknown variable " + name)
            if tuple(t.shape) != self.entries[name]["shape"]:
                raise ValueError("shape mismatch for %s: %s vs %s" % (name, tuple(t.shape), self.entries[name]["shape"]))
            self.view(name).copy_(t.to(self.device, torch.float32))
        self.repack()

    def state_dict(self):
        return OrderedDict((n, self.view(n).detach().cpu().clone()) for n in self.entries)

    def repack(self):
        """Refresh both compute copies from the fp32 master (one launch for the whole store)."""
        if self.ntab:
            K.pack_weights_both(self.flat, self.wT, self.wN, self.table, self.ntab)
        if self.frag:
            K.pack_weights_frag(self.flat, self.wTf, self.wNf, self.frag_table, len(self.frag))
        if self.act_dtype == torch.bfloat16 and K4S2_FRAG:
            # the discriminator's 4x4 stride-2 convs (csrc/conv4x4s2.hip): fragment-order copies of both operands, refreshed with
            # the other compute copies (trainable weights: every step) -- all of them in ONE launch, out of wT / wN
            if self._k4 is None:
                rows, off = [], 0
                for name, e in self.entries.items():
                    if e.get("taps") != 16 or e["Apad"] != e["A"] or e["Bpad"] != e["B"] or e["A"] % 64 or e["B"] % 64:
                        continue
                    A, B, n = e["A"], e["B"], 16 * e["A"] * e["B"]
                    rows.append([e["packed"], off, 16, B, A])                       # forward operand: wT [16][B][A]
                    rows.append([e["packed"] | (1 << 62), off + n, 16, A, B])       # input gradient: wN [16][A][B]
                    self.wide[name] = [off, off + n, n]
                    off += 2 * n
                buf = torch.empty(max(off, 8), device=self.device, dtype=torch.bfloat16)
                for name, w in list(self.wide.items()):
                    if len(w) == 3:
                        self.wide[name] = [buf[w[0]:w[0] + w[2]], buf[w[1]:w[1] + w[2]]]
                self._k4 = (buf, torch.tensor(rows or [[0, 0, 16, 16, 32]], dtype=torch.int64, device=self.device).contiguous(), len(rows))
            if self._k4[2]:
                K.pack_taps_frag_multi(self.wT, self.wN, self._k4[0], self._k4[1], self._k4[2])
        if self.wide_frag:
            for name, e in self.entries.items():
                if e.get("taps") != 9 or e["Apad"] != e["A"] or e["Bpad"] != e["B"]:
                    continue
                A, B = e["A"], e["B"]
                w = self.wide.setdefault(name, [None, None])
                if A % 32 == 0 and A > 64 and B % 64 == 0:         # forward: Cin = A, Cout = B
                    if w[0] is None:
                        w[0] = torch.empty(9 * A * B, device=self.device, dtype=torch.bfloat16)
                    K.pack_wide_frag(self.packed(name, True), w[0], B, A, False)
                if B % 32 == 0 and B > 64 and A % 64 == 0:         # input gradient: Cin = B, Cout = A, taps mirrored
                    if w[1] is None:
                        w[1] = torch.empty(9 * A * B, device=self.device, dtype=torch.bfloat16)
                    K.pack_wide_frag(self.packed(name, False), w[1], A, B, True)

    def zero_grad(self):
        self.grad.zero_()

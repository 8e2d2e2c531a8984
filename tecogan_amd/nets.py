"""FNet, generator_F, discriminator_F and VGG-19 as explicit kernel schedules (forward + hand-written
backward) over the C-ABI kernels -- no autograd graph, no tracing: every launch below is an enqueue on
the current HIP stream, so a whole training step is one capturable hipGraph.

Fusion plan (what the reference runs as separate TF ops, lib/frvsr.py / lib/Teco.py / lib/ops.py):
  forward : conv + bias + {ReLU | LeakyReLU | tanh*24 | sigmoid} + residual add    -> one MFMA kernel
  backward: bwd_data + residual-gradient add + act'(saved output) of the PRODUCER   -> one MFMA kernel
            maxpool/upsample backward also apply the producer's LeakyReLU derivative
  weights : gradients are accumulated straight into the flat fp32 gradient buffer (ParamStore.grad).
Activations are NHWC in `ps.act_dtype` (fp32 parity mode / bf16 throughput mode); network outputs
that feed losses or the recurrence (HR frame, flow, D probability) are fp32.
"""
import os

import torch

from . import kernels as K
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH
from . import params
from .params import DIS_BLOCKS, FNET_BLOCKS, VGG_CFG, pad8

_F32 = torch.float32


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# --------------------------------------------------------------------------------------------------
# layer primitives
# --------------------------------------------------------------------------------------------------
# Kernel choices settled by same-box A/Bs (plain attributes: tools/mb_vgg.py --no-wide and the like set them):
# wide frozen 3x3 layers (VGG-19 conv2_2 ... conv5_4) through csrc/conv3x3_wr.hip when the store keeps fragment-order copies
# (conv3x3_dma.hip without: VGG pass of 48 images 1650 -> 1517 us, profiles/r05c_mb_vgg.txt)
WIDE_FRAG = True
# the discriminator's batch-norm statistics from the epilogue of its 4x4 stride-2 convs (csrc/conv4x4s2.hip) instead of two
# reduction launches per layer and pass (fwd_loss 0.615 -> 0.581 ms, profiles/r05h_seg_timeline.txt)
BN_STATS_IN_CONV = True


def conv_fwd(ps, wname, bname, x, stride=1, act=ACT_NONE, alpha=0.0, res=None, out_dtype=None, out=None, flags=0):
    """slim.conv2d SAME (reference lib/ops.py:47-56) + fused epilogue.  x [N,H,W,Cin_pad]."""
    e = ps.entries[wname]
    N, H, W, Cp = x.shape
    assert Cp == e["Apad"], (wname, Cp, e["Apad"])
    k = e["k"]
    Ho, pt = K.same_pad(H, k, stride)
    Wo, pl = K.same_pad(W, k, stride)
    if out is None:
        out = _empty((N, Ho, Wo, e["B"]), out_dtype or ps.act_dtype, x)
    d = K.conv_desc(N, H, W, Cp, Ho, Wo, e["B"], k, k, stride, pt, pl, 0, K.dt(x), K.dt(out), act, alpha, flags=flags)
    # (fragment-order copies: the 4x4 ones are refreshed by ParamStore.repack only while params.K4S2_FRAG holds -- looked up under
    #  the same switch, or a store built with it and flipped later would run on stale weights)
    wf = ps.packed_wide(wname, True) if ((WIDE_FRAG and k == 3) or (params.K4S2_FRAG and k == 4)) else None
    if wf is not None and k == 4 and K.conv4x4s2_frag_ok(d, ps.view(bname) if bname else None, None):
        K.conv4x4s2_frag(d, x, wf, ps.view(bname) if bname else None, res, None, out)
    elif wf is not None and k == 3 and K.conv3x3_wide_frag_ok(d, res):
        K.conv3x3_wide_frag(d, x, wf, ps.view(bname) if bname else None, res, None, out)
    else:
        K.conv_forward(d, x, ps.packed(wname, True), ps.view(bname) if bname else None, res, None, out)
    return out


def conv_bwd_data(ps, wname, dy, in_hw, stride=1, res=None, aux=None, mask_act=ACT_NONE, mask_alpha=0.0, out=None,
                  flags=0):
    """Input gradient of conv_fwd: transposed mode over the natural (HWIO) copy.  -> [N,H,W,Cin_pad]."""
    e = ps.entries[wname]
    N, Ho, Wo, Co = dy.shape
    H, W = in_hw
    k = e["k"]
    _, pt = K.same_pad(H, k, stride)
    _, pl = K.same_pad(W, k, stride)
    dx = _empty((N, H, W, e["Apad"]), ps.act_dtype, dy) if out is None else out
    d = K.conv_desc(N, Ho, Wo, Co, H, W, e["Apad"], k, k, stride, pt, pl, 1, K.dt(dy), K.dt(dx), 0, 0.0,
                    mask_act, mask_alpha, flags=flags)
    wf = ps.packed_wide(wname, False) if ((WIDE_FRAG and k == 3) or (params.K4S2_FRAG and k == 4)) else None
    if wf is not None and k == 4 and K.conv4x4s2_frag_ok(d, None, aux):
        K.conv4x4s2_frag(d, dy, wf, None, res, aux, dx)
    elif wf is not None and k == 3 and K.conv3x3_wide_frag_ok(d, res):
        K.conv3x3_wide_frag(d, dy, wf, None, res, aux, dx)
    else:
        K.conv_forward(d, dy, ps.packed(wname, False), None, res, aux, dx)
    return dx


def conv_wgrad(ps, wname, bname, x, dy, stride=1, flags=0):
    """dW (HWIO) and dbias accumulated into the flat gradient buffer."""
    e = ps.entries[wname]
    N, H, W, Cp = x.shape
    _, Ho, Wo, Co = dy.shape
    k = e["k"]
    _, pt = K.same_pad(H, k, stride)
    _, pl = K.same_pad(W, k, stride)
    d = K.conv_desc(N, H, W, e["A"], Ho, Wo, e["B"], k, k, stride, pt, pl, 0, 0, 0, flags=flags)     # logical channel counts
    K.conv_wgrad(d, x, dy, ps.gview(wname), ps.gview(bname) if bname else None, ldx=Cp, ldy=Co)


def conv_wgrad_args(ps, wname, bname, x, dy, stride=1, flags=0):
    """The arguments conv_wgrad would launch with, for K.conv_wgrad_multi: (desc, x, dy, dW view, dbias view, ldx, ldy)."""
    e = ps.entries[wname]
    N, H, W, Cp = x.shape
    _, Ho, Wo, Co = dy.shape
    k = e["k"]
    _, pt = K.same_pad(H, k, stride)
    _, pl = K.same_pad(W, k, stride)
    d = K.conv_desc(N, H, W, e["A"], Ho, Wo, e["B"], k, k, stride, pt, pl, 0, 0, 0, flags=flags)
    return d, x, dy, ps.gview(wname), ps.gview(bname) if bname else None, Cp, Co


def deconv_fwd(ps, wname, bname, x, act=ACT_NONE, alpha=0.0, out=None, flags=0):
    """slim.conv2d_transpose k3 s2 SAME (reference lib/ops.py:35-44, [TF1] A.2): transposed mode, pad 0."""
    e = ps.entries[wname]                      # TF layout [kh,kw,Cout,Cin] -> A = Cout, B = Cin
    N, H, W, Ci = x.shape
    k = e["k"]
    if out is None:
        out = _empty((N, 2 * H, 2 * W, e["A"]), ps.act_dtype, x)
    d = K.conv_desc(N, H, W, Ci, 2 * H, 2 * W, e["A"], k, k, 2, 0, 0, 1, K.dt(x), K.dt(out), act, alpha, flags=flags)
    K.conv_forward(d, x, ps.packed(wname, False), ps.view(bname), None, None, out)
    return out


def deconv_bwd_data(ps, wname, dy, aux=None, mask_act=ACT_NONE, mask_alpha=0.0, out=None, flags=0):
    e = ps.entries[wname]
    N, H2, W2, Co = dy.shape
    k = e["k"]
    dx = _empty((N, H2 // 2, W2 // 2, e["B"]), ps.act_dtype, dy) if out is None else out
    d = K.conv_desc(N, H2, W2, Co, H2 // 2, W2 // 2, e["B"], k, k, 2, 0, 0, 0, K.dt(dy), K.dt(dx), 0, 0.0,
                    mask_act, mask_alpha, flags=flags)
    K.conv_forward(d, dy, ps.packed(wname, True), None, None, aux, dx)
    return dx


def deconv_wgrad(ps, wname, bname, x, dy, flags=0):
    """dW in TF [kh,kw,Cout,Cin] layout: X := dy (gathered, stride 2), Y := x; dbias = colsum(dy)."""
    e = ps.entries[wname]
    N, H2, W2, Co = dy.shape
    k = e["k"]
    d = K.conv_desc(N, H2, W2, Co, H2 // 2, W2 // 2, e["B"], k, k, 2, 0, 0, 0, 0, 0, flags=flags)
    K.conv_wgrad(d, dy, x, ps.gview(wname), None)
    K.colsum(dy, dy.numel() // Co, Co, ps.gview(bname))


# --------------------------------------------------------------------------------------------------
# generator_F -- reference lib/frvsr.py:44-88
# --------------------------------------------------------------------------------------------------
GEN_CPAD = pad8(51)


class Generator:
    """generator_F.  Two ways to run it:
      * `forward(x_in)`                     : stateless (inference / API calls), nothing kept;
      * `begin_sequence` + `forward_t` / `backward_t` + `wgrad_sequence` : the training recurrence.  Every layer's
        activations and output-gradients of all T frames live in frame-major `[T*B, ...]` buffers, the per-frame
        backward only runs the (sequential) bwd_data chain, and the weight gradients -- the weights are shared by
        all frames -- are computed ONCE per layer over the whole sequence afterwards: T x fewer launches and
        T x fewer atomics than per-frame wgrad, and off the critical path of the recurrence."""
    P = "generator/generator_unit/"

    def __init__(self, ps, num_resblock):
        self.ps, self.nres = ps, num_resblock
        self.seq = None
        # Kernel-selection attributes (plain attributes, no environment switches: tests and tools/ flip them for A/Bs).
        # grouped_wgrad : ONE grouped weight-gradient launch for all res-block convs (FRVSR step 4.25 -> 3.91 ms, profiles/r01o_*)
        # hr_tail       : fused HR tail of the stateless (inference) forward for bf16 tensors -- the phase-form kernel of the
        #                 training recurrence without its t2 store (csrc/hr_fwd_lat.hip; 1080p tail 227 -> 164 us against the
        #                 round-3 kernel it replaced, profiles/r04m_ab.txt)
        # ws_frag       : inference res-block convs with the weights in fragment order (csrc/conv3x3_ws.hip, profiles/r04q_ab.txt)
        # resblock_lat / hr_fwd_lat / hr_bwd_lat : the latency-regime kernels of the training recurrence (one launch per residual
        #                 block, per transposed conv / HR tail, per BPTT HR tail); off = the generic tg_conv_forward launches
        #                 (bit-identical blocks: tests/test_train_gpu.py)
        self.grouped_wgrad = True
        self.hr_tail = True
        self.ws_frag = True
        self.fused_block = True        # throughput regime (inference): a residual block as ONE launch (csrc/resblock_thr.hip)
        # resblock_plane : throughput regime, the WHOLE trunk of a frame as one persistent launch with the activations resident in
        #                 LDS (csrc/resblock_plane.hip; bit-identical to nb x tg_resblock); needs one 16x32-pixel tile per compute
        #                 unit at most and at least half the chip's worth of tiles (1080p output: 255 tiles); else fused_block
        self.resblock_plane = True
        self.plane_variant = 0
        self.plane_input_conv = True   # ... with the input-stage conv in front of the first block, in the same launch
        self._plane_scratch = {}       # (N, H, W) -> exchange scratch, allocated (zeroed) on first use: in the eager warm-up run
        self.input_in_group = True     # the input conv's weight gradient as a narrower last group of the trunk's grouped launch
        self.resblock_lat = self.hr_fwd_lat = self.hr_bwd_lat = True
        # resblock_chain : the whole residual trunk of a frame (and its input-gradient chain) as ONE persistent launch with neighbour
        #                 hand-offs instead of kernel boundaries (csrc/resblock_chain.hip; bit-identical to the per-block launches);
        #                 needs one 4x4 tile per compute unit at most (the training crops: [4,32,32] = 256 tiles)
        self.resblock_chain = True
        self.chain_variant = 14 << 1            # prefetch distance of the weight stream (tools/mb_chain.py)
        self.chain_input_conv = True            # ... with the input-stage conv in front of the first block, in the same launch
        self.resblock_max_tiles = 1024          # 4x4-pixel tiles up to which one workgroup per tile is the latency-optimal shape
        # scheduling hint for the recurrence's own launches (forward_t / backward_t): K.CONV_COEXIST when throughput work
        # of another stream shares the chip, so that the chain's bigger launches (HR deconv, output conv) also pick tile
        # shapes that fit NEXT to a resident VGG workgroup instead of waiting for a CU to drain
        self.chain_flags = 0

    # ---- stateless forward -----------------------------------------------------------------------
    def forward(self, x_in, keep=False, out=None, state=None):
        """x_in [B,h,w,56] (LR frame | s2d(warped prev HR) | 0-pad) -> HR frame [B,4h,4w,3] fp32 in [-1,1].
        state: optional fp32 [B,4h,4w,3] that receives deprocess(frame) in the same pass (the inference loop's recurrent
        state, main.py:207); with a state and out=False the [-1,1] frame itself is not written at all."""
        assert not keep, "training uses begin_sequence/forward_t"
        ps, p = self.ps, self.P
        win = p + "input_stage/conv/Conv/weights"
        # (the trunk's launch can take the input-stage conv in front of its first block: decided before that conv is launched)
        plane_in = (self.resblock_plane and self.plane_input_conv and self.ws_frag and ps.frag and x_in.dtype == torch.bfloat16
                    and 1 <= self.nres <= 16 and K.resblock_plane_ok(*x_in.shape[:3]) and ps.packed_frag(win, True) is not None
                    and x_in.shape[-1] % 8 == 0 and x_in.shape[-1] <= 64)
        a = None if plane_in else conv_fwd(ps, win, p + "input_stage/conv/Conv/biases", x_in, 1, ACT_RELU)
        # (one launch per residual block -- r kept in LDS, both convs on MFMA -- was built and measured in round 3: parity
        #  green, 35.2 us per block against 35.3 us for these two launches: bound by 2-way-conflicted LDS fragment reads under
        #  the gfx950 ds_read_b128 lane grouping; numbers and cycle stamps in profiles/r03p_resblock_ws.txt, kernel deleted)
        wsf = plane_in or (self.ws_frag and ps.frag and a.dtype == torch.bfloat16 and K.conv3x3_c64_frag_ok(*a.shape[:3]))
        bufs = [torch.empty_like(a), torch.empty_like(a)] if (wsf and self.fused_block and not plane_in) else None
        plane = plane_in or (wsf and self.resblock_plane and 1 <= self.nres <= 16 and K.resblock_plane_ok(*a.shape[:3]))
        if plane:
            key = tuple(x_in.shape[:3])
            if key not in self._plane_scratch:
                assert not torch.cuda.is_current_stream_capturing(), "the exchange scratch is zeroed ONCE: allocate it in the eager warm-up"
                self._plane_scratch[key] = K.resblock_plane_scratch(*key, x_in.device)
            names = [p + "resblock_%d/" % i for i in range(1, self.nres + 1)]
            pre = (x_in, ps.packed_frag(win, True), ps.view(p + "input_stage/conv/Conv/biases")) if plane_in else None
            trunk = a if a is not None else torch.empty(*key, 64, device=x_in.device, dtype=x_in.dtype)
            a = K.resblock_plane(a, [ps.packed_frag(s + "conv_1/Conv/weights", True) for s in names],
                                 [ps.view(s + "conv_1/Conv/biases") for s in names],
                                 [ps.packed_frag(s + "conv_2/Conv/weights", True) for s in names],
                                 [ps.view(s + "conv_2/Conv/biases") for s in names], trunk, self._plane_scratch[key], self.plane_variant,
                                 pre=pre)
        for i in range(1, (0 if plane else self.nres) + 1):
            s = p + "resblock_%d/" % i
            if bufs is not None:
                # the whole block in one launch, intermediate in LDS (csrc/resblock_thr.hip; not in place: ping-pong buffers)
                a = K.resblock_c64_thr(a, ps.packed_frag(s + "conv_1/Conv/weights", True), ps.view(s + "conv_1/Conv/biases"),
                                       ps.packed_frag(s + "conv_2/Conv/weights", True), ps.view(s + "conv_2/Conv/biases"), bufs[i & 1])
                continue
            if wsf:
                # same kernel, weights in fragment order: whole-line weight loads, two workgroups per CU (csrc/conv3x3_ws.hip)
                r = K.conv3x3_c64_frag(a, ps.packed_frag(s + "conv_1/Conv/weights", True), ps.view(s + "conv_1/Conv/biases"), None,
                                       torch.empty_like(a), ACT_RELU)
                a = K.conv3x3_c64_frag(r, ps.packed_frag(s + "conv_2/Conv/weights", True), ps.view(s + "conv_2/Conv/biases"), a,
                                       torch.empty_like(a), ACT_NONE)
                continue
            r = conv_fwd(ps, s + "conv_1/Conv/weights", s + "conv_1/Conv/biases", a, 1, ACT_RELU)
            a = conv_fwd(ps, s + "conv_2/Conv/weights", s + "conv_2/Conv/biases", r, 1, ACT_NONE, 0.0, res=a)
        s = p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/"
        t1 = deconv_fwd(ps, s % 1 + "weights", s % 1 + "biases", a, ACT_RELU)
        if self.hr_tail and t1.dtype == torch.bfloat16 and ps.frag:
            # fused HR tail: the phase-form kernel of the training recurrence, persistent at this size and without its t2 store
            # (csrc/hr_fwd_lat.hip) -- the 64-channel HR tensor t2 is never written
            wo, bo = p + "output_stage/conv/Conv/weights", p + "output_stage/conv/Conv/biases"
            N, h2, w2, _ = t1.shape
            o = None if out is False else (torch.empty(N, 2 * h2, 2 * w2, 3, device=t1.device) if out is None else out)
            assert o is not None or state is not None
            r = K.hr_tail_train(t1, ps.packed_frag(s % 2 + "weights", False), ps.view(s % 2 + "biases"), ps.packed(wo, True),
                                ps.view(bo), x_in, None, o, state)
            return r, None
        t2 = deconv_fwd(ps, s % 2 + "weights", s % 2 + "biases", t1, ACT_RELU)
        c = conv_fwd(ps, p + "output_stage/conv/Conv/weights", p + "output_stage/conv/Conv/biases", t2, 1,
                     out_dtype=_F32)
        if out is False:
            assert state is not None
            return K.bicubic_add_preprocess(c, x_in, None, state), None
        out = K.bicubic_add_preprocess(c, x_in, torch.empty_like(c) if out is None else out, state)  # (c+bicubic(LR))*2-1
        return out, None

    # ---- training recurrence -----------------------------------------------------------------------
    def begin_sequence(self, T, B, h, w, dev):
        """Allocate the frame-major activation / gradient buffers of one training step."""
        dt, n = self.ps.act_dtype, self.nres

        def buf(hh, ww, c, dtype=None):
            return torch.empty(T, B, hh, ww, c, device=dev, dtype=dtype or dt)

        q = dict(T=T, B=B, h=h, w=w)
        q["x_in"] = buf(h, w, GEN_CPAD)
        q["a"] = [buf(h, w, 64) for _ in range(n + 1)]          # a[0] = relu(input conv), a[i] = block i output
        q["r"] = [None] + [buf(h, w, 64) for _ in range(n)]     # r[i] = relu(conv_1 of block i)
        q["t1"], q["t2"] = buf(2 * h, 2 * w, 64), buf(4 * h, 4 * w, 64)
        q["c"] = torch.empty(B, 4 * h, 4 * w, 3, device=dev)    # scratch (per frame)
        # gradients w.r.t. each conv's pre-activation output
        q["g_in"] = buf(h, w, 64)
        q["g_c1"] = [None] + [buf(h, w, 64) for _ in range(n)]
        q["g_c2"] = [None] + [buf(h, w, 64) for _ in range(n)]
        q["g_t1"], q["g_t2"] = buf(2 * h, 2 * w, 64), buf(4 * h, 4 * w, 64)
        q["g_out"] = buf(4 * h, 4 * w, 8)                       # 3 real channels, zero-padded: 16-B rows for the MFMA paths
        q["dx_in"] = torch.empty(B, h, w, GEN_CPAD, device=dev, dtype=dt)
        q["chain"] = {}                                         # (mode, t) -> K.ChainArgs, built on first use
        q["chain_scratch"] = None
        self.seq = q
        return q

    def _chain(self, mode, t):
        """The trunk of frame t as one persistent launch: pointer arrays built once per (direction, frame)."""
        ps, p, q, n = self.ps, self.P, self.seq, self.nres
        ca = q["chain"].get((mode, t))
        if ca is None:
            if q["chain_scratch"] is None:
                q["chain_scratch"] = K.resblock_chain_scratch(q["B"], q["h"], q["w"], q["a"][0].device)
            cas = []
            order = list(range(1, n + 1)) if mode == 0 else list(range(n, 0, -1))
            for c0 in range(0, n, 16):                          # at most 16 blocks per launch
                blk = order[c0:c0 + 16]
                nm = [p + "resblock_%d/" % i for i in blk]
                if mode == 0:
                    x = q["a"][blk[0] - 1][t]
                    wi = p + "input_stage/conv/Conv/weights"
                    pre = None
                    if c0 == 0 and self.chain_input_conv and ps.packed_frag(wi, True) is not None:
                        # the input-stage conv in front of the first block, in the same launch (a[0] is written by it)
                        pre = (q["x_in"][t], ps.packed_frag(wi, True), ps.view(p + "input_stage/conv/Conv/biases"), x)
                    cas.append(K.ChainArgs(0, None if pre else x, [ps.packed_frag(s + "conv_1/Conv/weights", True) for s in nm],
                                           [ps.view(s + "conv_1/Conv/biases") for s in nm],
                                           [ps.packed_frag(s + "conv_2/Conv/weights", True) for s in nm],
                                           [ps.view(s + "conv_2/Conv/biases") for s in nm], None, None,
                                           [q["r"][i][t] for i in blk], [q["a"][i][t] for i in blk], q["chain_scratch"],
                                           self.chain_variant, pre))
                else:
                    x = q["g_c2"][blk[0]][t]
                    cas.append(K.ChainArgs(1, x, [ps.packed_frag(s + "conv_2/Conv/weights", False) for s in nm], None,
                                           [ps.packed_frag(s + "conv_1/Conv/weights", False) for s in nm], None,
                                           [q["r"][i][t] for i in blk], q["a"][0][t] if blk[-1] == 1 else None,
                                           [q["g_c1"][i][t] for i in blk],
                                           [q["g_c2"][i - 1][t] if i > 1 else q["g_in"][t] for i in blk], q["chain_scratch"],
                                           self.chain_variant))
            ca = q["chain"][(mode, t)] = cas
        for c in ca:
            c.launch()

    def handoff_give_ups(self):
        """Workgroups of the one-launch trunks (csrc/resblock_chain.hip, resblock_plane.hip) that gave up waiting for a neighbour
        since the scratch was allocated (sticky counters; reading them synchronises).  Non-zero means a launch could not get all
        its workgroups resident within the spin bound and its result was garbage: callers check at natural sync points and raise."""
        n = 0
        if self.seq is not None and self.seq.get("chain_scratch") is not None:
            n += int(self.seq["chain_scratch"][2])
        for sc in self._plane_scratch.values():
            n += int(sc[2])
        return n

    def _chained(self):
        q = self.seq
        return (self._fused_blocks() and self.resblock_chain and self.nres >= 2 and self.ps.frag
                and K.resblock_chain_ok(q["B"], q["h"], q["w"]))

    def _fused_blocks(self):
        """True when the recurrence runs its residual blocks as one launch each: bf16 frames whose 4x4-pixel tiles are few
        enough for one workgroup per tile to be the LATENCY-optimal shape (the training crops; not the 1080p stream)."""
        q = self.seq
        tiles = q["B"] * ((q["h"] + 3) // 4) * ((q["w"] + 3) // 4)
        return self.resblock_lat and self.ps.act_dtype == torch.bfloat16 and tiles <= self.resblock_max_tiles

    def forward_t(self, t, out):
        """Frame t: reads seq['x_in'][t] (filled by the warp kernel), writes the HR frame into `out`."""
        ps, p, q = self.ps, self.P, self.seq
        x_in = q["x_in"][t]
        cf = self.chain_flags
        fused = self._fused_blocks()
        chained = self._chained()
        in_chain = chained and self.chain_input_conv and ps.packed_frag(p + "input_stage/conv/Conv/weights", True) is not None
        if not in_chain:
            a = conv_fwd(ps, p + "input_stage/conv/Conv/weights", p + "input_stage/conv/Conv/biases", x_in, 1, ACT_RELU,
                         out=q["a"][0][t], flags=cf)
        if chained:
            self._chain(0, t)
            a = q["a"][self.nres][t]
        for i in range(1, 0 if chained else self.nres + 1):
            s = p + "resblock_%d/" % i
            if fused:
                a = K.resblock(0, a, ps.packed_frag(s + "conv_1/Conv/weights", True), ps.view(s + "conv_1/Conv/biases"),
                               ps.packed_frag(s + "conv_2/Conv/weights", True), ps.view(s + "conv_2/Conv/biases"), None, None,
                               q["r"][i][t], q["a"][i][t], w_frag=True)
                continue
            r = conv_fwd(ps, s + "conv_1/Conv/weights", s + "conv_1/Conv/biases", a, 1, ACT_RELU, out=q["r"][i][t], flags=cf)
            a = conv_fwd(ps, s + "conv_2/Conv/weights", s + "conv_2/Conv/biases", r, 1, ACT_NONE, 0.0, res=a,
                         out=q["a"][i][t], flags=cf)
        s = p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/"
        if fused and self.hr_fwd_lat:
            # the two transposed convs as latency-regime launches; the second one fused with the output conv and the bicubic skip
            t1 = K.deconv_lat_forward(a, ps.packed_frag(s % 1 + "weights", False), ps.view(s % 1 + "biases"), q["t1"][t])
            return K.hr_tail_train(t1, ps.packed_frag(s % 2 + "weights", False), ps.view(s % 2 + "biases"),
                                   ps.packed(p + "output_stage/conv/Conv/weights", True),
                                   ps.view(p + "output_stage/conv/Conv/biases"), x_in, q["t2"][t], out)
        t1 = deconv_fwd(ps, s % 1 + "weights", s % 1 + "biases", a, ACT_RELU, out=q["t1"][t], flags=cf)
        t2 = deconv_fwd(ps, s % 2 + "weights", s % 2 + "biases", t1, ACT_RELU, out=q["t2"][t], flags=cf)
        c = conv_fwd(ps, p + "output_stage/conv/Conv/weights", p + "output_stage/conv/Conv/biases", t2, 1, out=q["c"],
                     flags=cf)
        return K.bicubic_add_preprocess(c, x_in, out)

    def backward_t(self, t, d_out, need_dx=True):
        """bwd_data chain of frame t (no weight gradients here).  Returns d x_in [B,h,w,56] or None."""
        ps, p, q, n = self.ps, self.P, self.seq, self.nres
        h, w = q["h"], q["w"]
        cf = self.chain_flags
        s = p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/"
        if self._fused_blocks() and self.hr_bwd_lat:
            # frame gradient -> g_out, g_t2 (both kept for the weight gradients) and g_t1 in ONE launch (csrc/hr_bwd_lat.hip)
            g = K.hr_tail_backward(d_out, 2.0, ps.packed(p + "output_stage/conv/Conv/weights", False), q["t2"][t],
                                   ps.packed_frag(s % 2 + "weights", True), q["t1"][t], q["g_out"][t], q["g_t2"][t], q["g_t1"][t])
        else:
            dc = K.concat2_pad(d_out, None, q["g_out"][t], scale=2.0)                          # d/dc of (.)*2-1
            g = conv_bwd_data(ps, p + "output_stage/conv/Conv/weights", dc, (4 * h, 4 * w), 1, aux=q["t2"][t],
                              mask_act=ACT_RELU, out=q["g_t2"][t], flags=cf)
            g = deconv_bwd_data(ps, s % 2 + "weights", g, aux=q["t1"][t], mask_act=ACT_RELU, out=q["g_t1"][t], flags=cf)
        if self._fused_blocks() and self.hr_bwd_lat:
            g = K.deconv_lat_backward(g, ps.packed_frag(s % 1 + "weights", True), None, q["g_c2"][n][t] if n else q["g_in"][t])
        else:
            g = deconv_bwd_data(ps, s % 1 + "weights", g, out=q["g_c2"][n][t] if n else q["g_in"][t], flags=cf)
        if n == 0:
            g = K.act_backward(g, q["a"][0][t], g, ACT_RELU)
        fused = self._fused_blocks()
        chained = self._chained()
        if chained:
            self._chain(1, t)
            g = q["g_in"][t]
        for i in range(n if not chained else 0, 0, -1):
            sc = p + "resblock_%d/" % i
            if fused:
                # d r = bwd(conv_2)(g) * relu'(r) -> g_c1 (conv_1's weight gradient reads it); d a_{i-1} = bwd(conv_1)(d r) + g
                g = K.resblock(1, g, ps.packed_frag(sc + "conv_2/Conv/weights", False), None,
                               ps.packed_frag(sc + "conv_1/Conv/weights", False), None, q["r"][i][t],
                               q["a"][0][t] if i == 1 else None, q["g_c1"][i][t],
                               q["g_c2"][i - 1][t] if i > 1 else q["g_in"][t], w_frag=True)
                continue
            dr = conv_bwd_data(ps, sc + "conv_2/Conv/weights", g, (h, w), 1, aux=q["r"][i][t], mask_act=ACT_RELU,
                               out=q["g_c1"][i][t], flags=cf)
            # d a_{i-1} = bwd(conv_1)(dr) + skip gradient; block 1's input is itself a ReLU output (masked here)
            g = conv_bwd_data(ps, sc + "conv_1/Conv/weights", dr, (h, w), 1, res=g,
                              aux=q["a"][0][t] if i == 1 else None, mask_act=ACT_RELU if i == 1 else ACT_NONE,
                              out=q["g_c2"][i - 1][t] if i > 1 else q["g_in"][t], flags=cf)
        if not need_dx:
            return None
        return conv_bwd_data(ps, p + "input_stage/conv/Conv/weights", g, (h, w), 1, out=q["dx_in"], flags=cf)

    def wgrad_sequence(self, t0=0, t1=None, flags=0):
        """Weight / bias gradients of frames [t0, t1): one launch per layer over the (t1-t0)*B frames.
        flags: K.CONV_COEXIST when the launches run beside the BPTT chain (capped residency)."""
        ps, p, q, n = self.ps, self.P, self.seq, self.nres
        t1 = q["T"] if t1 is None else t1
        if t1 <= t0:
            return

        def flat(x):
            x = x[t0:t1]
            return x.reshape(-1, *x.shape[2:])

        wi, bi = p + "input_stage/conv/Conv/weights", p + "input_stage/conv/Conv/biases"
        with_input = self.grouped_wgrad and n >= 1 and self.input_in_group and 2 * n + 1 <= 40
        if not with_input:
            conv_wgrad(ps, wi, bi, flat(q["x_in"]), flat(q["g_in"]), flags=flags)
        if self.grouped_wgrad and n >= 1:
            # the 2n res-block convs have one geometry and their gradients are all due here: ONE grouped launch
            # (tg_conv_wgrad_grouped) instead of 2n -- the per-launch fixed cost is paid once
            names, xs, dys = [], [], []
            for i in range(1, n + 1):
                sc = p + "resblock_%d/" % i
                names += [sc + "conv_1/Conv/", sc + "conv_2/Conv/"]
                xs += [flat(q["a"][i - 1]), flat(q["r"][i])]
                dys += [flat(q["g_c1"][i]), flat(q["g_c2"][i])]
            e = ps.entries[names[0] + "weights"]
            N, H, W, Cp = xs[0].shape
            _, pt = K.same_pad(H, e["k"], 1)
            _, pl = K.same_pad(W, e["k"], 1)
            d = K.conv_desc(N, H, W, e["A"], H, W, e["B"], e["k"], e["k"], 1, pt, pl, 0, 0, 0, flags=flags)
            if with_input:
                # ... and the input conv (51 channels in a 56-channel pixel, same images, 64 outputs) rides in the same launch as a
                # narrower last group (round 5: its own launch took 110 us for 5 GFLOP on the row kernel)
                xi = flat(q["x_in"])
                K.conv_wgrad_grouped_plus(d, xs, dys, [ps.gview(nm + "weights") for nm in names], [ps.gview(nm + "biases") for nm in names],
                                          (xi, xi.shape[-1], ps.entries[wi]["A"], flat(q["g_in"]), ps.gview(wi), ps.gview(bi)),
                                          ldx=Cp, ldy=dys[0].shape[-1])
            for g0 in range(0, 0 if with_input else len(names), 40):   # TG_WGRAD_MAX_GROUPS per call
                sl = slice(g0, g0 + 40)
                K.conv_wgrad_grouped(d, xs[sl], dys[sl], [ps.gview(nm + "weights") for nm in names[sl]],
                                     [ps.gview(nm + "biases") for nm in names[sl]], ldx=Cp, ldy=dys[0].shape[-1])
        else:
            for i in range(1, n + 1):
                sc = p + "resblock_%d/" % i
                conv_wgrad(ps, sc + "conv_1/Conv/weights", sc + "conv_1/Conv/biases", flat(q["a"][i - 1]), flat(q["g_c1"][i]),
                           flags=flags)
                conv_wgrad(ps, sc + "conv_2/Conv/weights", sc + "conv_2/Conv/biases", flat(q["r"][i]), flat(q["g_c2"][i]),
                           flags=flags)
        s = p + "conv_tran2highres/conv_tran%d/Conv2d_transpose/"
        deconv_wgrad(ps, s % 1 + "weights", s % 1 + "biases", flat(q["a"][n]), flat(q["g_t1"]), flags=flags)
        deconv_wgrad(ps, s % 2 + "weights", s % 2 + "biases", flat(q["t1"]), flat(q["g_t2"]), flags=flags)
        conv_wgrad(ps, p + "output_stage/conv/Conv/weights", p + "output_stage/conv/Conv/biases", flat(q["t2"]),
                   flat(q["g_out"]), flags=flags)


# --------------------------------------------------------------------------------------------------
# fnet -- reference lib/frvsr.py:4-41
# --------------------------------------------------------------------------------------------------
FNET_CPAD = pad8(6)


class FNet:
    P = "fnet/autoencode_unit/"

    def __init__(self, ps):
        self.ps = ps
        self.wgrad_multi = True        # the 14 weight gradients as ONE multi-geometry launch (651 -> 490 us, profiles/r03u_mb_fnet.txt)

    def forward(self, x, keep=True):
        """x [N,h,w,8] (prev LR | cur LR | 0-pad) -> flow [N,h',w',2] fp32 (h' = h - h%8)."""
        ps, p = self.ps, self.P
        saved = []
        net = x
        for name, _, _ in FNET_BLOCKS:
            s = p + name
            c1 = conv_fwd(ps, s + "/conv_1/Conv/weights", s + "/conv_1/Conv/biases", net, 1, ACT_LRELU, 0.2)
            c2 = conv_fwd(ps, s + "/conv_2/Conv/weights", s + "/conv_2/Conv/biases", c1, 1, ACT_LRELU, 0.2)
            N, H, W, Cc = c2.shape
            if name.startswith("encoder"):
                nxt = K.maxpool2_forward(c2, _empty((N, H // 2, W // 2, Cc), c2.dtype, c2))
            else:
                nxt = K.upsample2_forward(c2, _empty((N, 2 * H, 2 * W, Cc), c2.dtype, c2))
            saved.append((net, c1, c2))
            net = nxt
        s = p + "output_stage/"
        o1 = conv_fwd(ps, s + "conv1/Conv/weights", s + "conv1/Conv/biases", net, 1, ACT_LRELU, 0.2)
        flow = conv_fwd(ps, s + "conv2/Conv/weights", s + "conv2/Conv/biases", o1, 1, ACT_TANH, 24.0, out_dtype=_F32)
        return flow, ((saved, net, o1, flow) if keep else None)

    def backward(self, saved_all, d_flow, flags=0):
        """Backward pass: the input-gradient chain first, then the 14 weight gradients as ONE multi-geometry launch
        (tg_conv_wgrad_multi; `wgrad_multi = False`: one launch per layer, interleaved with the chain as in round 2): as
        separate launches they are 14 x 17-35 us of launch, prologue and atomics tails for 37 GFLOP of work."""
        ps, p = self.ps, self.P
        saved, net_last, o1, flow = saved_all
        todo = [] if self.wgrad_multi else None

        def wg(wname, bname, x, dy):
            if todo is None:
                conv_wgrad(ps, wname, bname, x, dy, flags=flags)
            else:
                todo.append(conv_wgrad_args(ps, wname, bname, x, dy, flags=flags))

        s = p + "output_stage/"
        if ps.entries[s + "conv2/Conv/weights"]["Bpad"] == 8:
            # the 2-channel flow gradient zero-padded to 8 channels (16-byte pixels), as the generator's 3-channel output
            # gradient is: its weight gradient then runs on the MFMA row kernel (83 -> ~20 us at 72 pairs, it used to fall to
            # the generic kernel) and its input gradient on the 8-channel conv kernel
            g2 = K.act_backward(d_flow, flow, torch.empty_like(flow), ACT_TANH, 24.0)
            g = K.concat2_pad(g2, None, _empty(flow.shape[:3] + (8,), ps.act_dtype, flow))
        else:
            g = K.act_backward(d_flow, flow, _empty(flow.shape, ps.act_dtype, flow), ACT_TANH, 24.0)
        wg(s + "conv2/Conv/weights", s + "conv2/Conv/biases", o1, g)
        g = conv_bwd_data(ps, s + "conv2/Conv/weights", g, o1.shape[1:3], 1, aux=o1, mask_act=ACT_LRELU, mask_alpha=0.2,
                          flags=flags)
        wg(s + "conv1/Conv/weights", s + "conv1/Conv/biases", net_last, g)
        g = conv_bwd_data(ps, s + "conv1/Conv/weights", g, net_last.shape[1:3], 1, flags=flags)          # d (resampled map)
        for bi in range(len(FNET_BLOCKS) - 1, -1, -1):
            name = FNET_BLOCKS[bi][0]
            sc = p + name
            x_in, c1, c2 = saved[bi]
            if name.startswith("encoder"):
                g = K.maxpool2_backward(c2, g, torch.empty_like(c2), ACT_LRELU, 0.2)
            else:
                g = K.upsample2_backward(g, torch.empty_like(c2), c2, ACT_LRELU, 0.2)
            wg(sc + "/conv_2/Conv/weights", sc + "/conv_2/Conv/biases", c1, g)
            g = conv_bwd_data(ps, sc + "/conv_2/Conv/weights", g, c1.shape[1:3], 1, aux=c1, mask_act=ACT_LRELU,
                              mask_alpha=0.2, flags=flags)
            wg(sc + "/conv_1/Conv/weights", sc + "/conv_1/Conv/biases", x_in, g)
            if bi > 0:
                g = conv_bwd_data(ps, sc + "/conv_1/Conv/weights", g, x_in.shape[1:3], 1, flags=flags)
        if todo:
            ds, xs, dys, dws, dbs, lxs, lys = zip(*todo)
            K.conv_wgrad_multi(list(ds), list(xs), list(dys), list(dws), list(dbs), list(lxs), list(lys))
        return None


# --------------------------------------------------------------------------------------------------
# discriminator_F -- reference lib/Teco.py:30-74
# --------------------------------------------------------------------------------------------------
DIS_CPAD = pad8(27)


class Discriminator:
    P = "tdiscriminator/discriminator_unit/"

    def __init__(self, ps):
        self.ps = ps
        dev = ps.device
        # [mean, var] moving statistics per block ([TF1] A.7; updated, never read by the path)
        self.moving = [torch.stack((torch.zeros(co), torch.ones(co))).to(dev).contiguous() for _, _, co in DIS_BLOCKS]
        self.scratch, self._cursor = None, 0

    def set_scratch(self, buf):
        """Give the BN statistics / backward sums a caller-owned fp32 pool that the caller zeroes ONCE per step
        (one fill node instead of one hipMemset per batch-norm call: 20 per TecoGAN step)."""
        self.scratch, self._cursor = buf, 0

    def _ws(self, co, like):
        """[2][co] fp32 accumulator: a slice of the pre-zeroed pool, or a fresh tensor the C side zeroes itself."""
        if self.scratch is None:
            return _empty((2, co), _F32, like), False
        a = self._cursor
        if a + 2 * co > self.scratch.numel():                  # pool sized for another schedule: a self-zeroed tensor instead
            return _empty((2, co), _F32, like), False
        self._cursor += 2 * co
        return self.scratch[a:a + 2 * co].view(2, co), True

    def alloc_pair(self, tb, H, W, like):
        """Activation buffers for TWO passes over tb samples each (the real and the fake triplets of one training step,
        lib/Teco.py:252-272) as halves of `[2 tb, ...]` tensors, so that the discriminator's own backward pass runs ONCE over
        the 2 tb samples (`backward_pair`): one input-gradient and one weight-gradient launch per layer instead of two -- the
        weights are shared and dW is a sum over the samples anyway; only the batch statistics stay per pass."""
        ps, dt = self.ps, self.ps.act_dtype
        cp = ps.entries[self.P + "input_stage/conv/Conv/weights"]["Apad"]
        pair = dict(tb=tb, x=_empty((2 * tb, H, W, cp), dt, like), a=_empty((2 * tb, H, W, 64), dt, like), c=[], y=[], dcv=[])
        h, w = H, W
        for _, _, co in DIS_BLOCKS:
            h, w = K.same_pad(h, 4, 2)[0], K.same_pad(w, 4, 2)[0]
            pair["c"].append(_empty((2 * tb, h, w, co), dt, like))
            pair["y"].append(_empty((2 * tb, h, w, co), dt, like))
            pair["dcv"].append(_empty((2 * tb, h, w, co), dt, like))
        pair["prob"] = _empty((2 * tb, h, w, 1), _F32, like)
        return pair

    @staticmethod
    def pair_half(pair, k):
        """The buffers of pass k (0 / 1) of a pair, in the form `forward(into=...)` takes."""
        tb = pair["tb"]
        sl = slice(k * tb, (k + 1) * tb)
        return dict(x=pair["x"][sl], a=pair["a"][sl], c=[t[sl] for t in pair["c"]], y=[t[sl] for t in pair["y"]],
                    prob=pair["prob"][sl])

    def forward(self, x, keep=True, update_moving=True, flags=0, into=None):
        """x [tb,H,W,32|16] -> (prob [tb,H/16,W/16,1] fp32, [4 layer maps], saved).  into: buffers from pair_half()."""
        ps, p = self.ps, self.P
        a = conv_fwd(ps, p + "input_stage/conv/Conv/weights", p + "input_stage/conv/Conv/biases", x, 1, ACT_LRELU, 0.2,
                     flags=flags, out=into["a"] if into else None)
        saved, layers, net = [], [], a
        for bi, (name, _, co) in enumerate(DIS_BLOCKS):
            wname = p + name + "/conv1/Conv/weights"
            wf = ps.packed_wide(wname, True) if (BN_STATS_IN_CONV and params.K4S2_FRAG) else None     # (refreshed only under K4S2_FRAG)
            N, H, W, Cp = net.shape
            d = K.conv_desc(N, H, W, Cp, H // 2, W // 2, co, 4, 4, 2, 1, 1, 0, K.dt(net), K.dt(net), flags=flags)
            if wf is not None and H % 2 == 0 and W % 2 == 0 and K.conv4x4s2_frag_ok(d):
                # the conv's epilogue leaves the batch statistics in `stats` (K.BN_STAT_REPLICAS partial sets, summed into the
                # first by the batch norm's one finishing launch): no reduction launches in the batch norm
                rep, pz = self._ws(co * K.BN_STAT_REPLICAS, net)
                if not pz:
                    rep.zero_()
                stats = rep.view(-1)[:2 * co].view(2, co)
                c = into["c"][bi] if into else _empty((N, H // 2, W // 2, co), ps.act_dtype, net)
                K.conv4x4s2_frag(d, net, wf, None, None, None, c, bn_stats=rep)
                pz = 2
            else:
                stats, pz = self._ws(co, net)
                c = conv_fwd(ps, wname, None, net, 2, flags=flags, out=into["c"][bi] if into else None)
            y = into["y"][bi] if into else torch.empty_like(c)
            K.bn_lrelu_forward(c, y, ps.view(p + name + "/BatchNorm/beta"), 1e-3, 0.2, stats,
                               self.moving[bi] if update_moving else None, prezeroed=pz)
            saved.append((net, c, y, stats))
            layers.append(y)
            net = y
        prob = conv_fwd(ps, p + "dense_layer_2/dense/kernel", p + "dense_layer_2/dense/bias", net, 1, ACT_SIGMOID,
                        out_dtype=_F32, out=into["prob"] if into else None)
        return prob, layers, ((x, a, saved, prob) if keep else None)

    def backward(self, saved_all, d_prob, d_layers=None, wgrad=True, need_dx=False, flags=0):
        """d_prob fp32 [tb,h,w,1]; d_layers: optional list of 4 gradients (act dtype) w.r.t. the layer maps.
        wgrad=False leaves the discriminator's own gradients untouched (generator-side pass)."""
        ps, p = self.ps, self.P
        x, a, saved, prob = saved_all
        g = K.act_backward(d_prob, prob, _empty(prob.shape, ps.act_dtype, prob), ACT_SIGMOID)
        wn, bn = p + "dense_layer_2/dense/kernel", p + "dense_layer_2/dense/bias"
        y_last = saved[-1][2]
        if wgrad:
            conv_wgrad(ps, wn, bn, y_last, g, flags=flags)
        g = conv_bwd_data(ps, wn, g, y_last.shape[1:3], 1, res=d_layers[3] if d_layers else None, flags=flags)
        for bi in range(len(DIS_BLOCKS) - 1, -1, -1):
            name, _, co = DIS_BLOCKS[bi]
            net_in, c, y, stats = saved[bi]
            ws, pz = self._ws(co, c)
            dbeta = ps.gview(p + name + "/BatchNorm/beta") if wgrad else None
            dcv = K.bn_lrelu_backward(c, y, g, torch.empty_like(c), stats, 1e-3, 0.2, dbeta, ws, prezeroed=pz)
            if wgrad:
                conv_wgrad(ps, p + name + "/conv1/Conv/weights", None, net_in, dcv, 2, flags=flags)
            if bi > 0:
                g = conv_bwd_data(ps, p + name + "/conv1/Conv/weights", dcv, net_in.shape[1:3], 2,
                                  res=d_layers[bi - 1] if d_layers else None, flags=flags)
            else:   # net_in = a = lrelu(input conv)
                g = conv_bwd_data(ps, p + name + "/conv1/Conv/weights", dcv, net_in.shape[1:3], 2, aux=a,
                                  mask_act=ACT_LRELU, mask_alpha=0.2, flags=flags)
        wn, bn = p + "input_stage/conv/Conv/weights", p + "input_stage/conv/Conv/biases"
        if wgrad:
            conv_wgrad(ps, wn, bn, x, g, flags=flags)
        return conv_bwd_data(ps, wn, g, x.shape[1:3], 1, flags=flags) if need_dx else None

    def backward_pair(self, pair, saved_pair, d_prob_pair, flags=0):
        """The discriminator's OWN gradients (t_discrim_loss, lib/Teco.py:393-417,425-428) from both passes of a pair in one
        sweep over the 2 tb samples.  saved_pair = (saved of pass 0, saved of pass 1) -- their tensors are the halves of `pair`;
        d_prob_pair fp32 [2 tb,h,w,1].  Per layer: batch-norm backward per pass (its sums are over ONE pass's samples, as the
        forward statistics were), then one input-gradient launch over all 2 tb samples; the weight gradients of all layers go
        out together at the end (tg_conv_wgrad_multi: they only need the saved activations and the dcv buffers)."""
        ps, p, tb = self.ps, self.P, pair["tb"]
        todo = []
        g = K.act_backward(d_prob_pair, pair["prob"], _empty(pair["prob"].shape, ps.act_dtype, pair["prob"]), ACT_SIGMOID)
        wn, bn = p + "dense_layer_2/dense/kernel", p + "dense_layer_2/dense/bias"
        y_last = pair["y"][-1]
        todo.append(conv_wgrad_args(ps, wn, bn, y_last, g, flags=flags))
        g = conv_bwd_data(ps, wn, g, y_last.shape[1:3], 1, flags=flags)
        for bi in range(len(DIS_BLOCKS) - 1, -1, -1):
            name, _, co = DIS_BLOCKS[bi]
            dcv = pair["dcv"][bi]
            for k in (0, 1):
                sl = slice(k * tb, (k + 1) * tb)
                _, c, y, stats = saved_pair[k][2][bi]
                ws, pz = self._ws(co, c)
                K.bn_lrelu_backward(c, y, g[sl], dcv[sl], stats, 1e-3, 0.2, ps.gview(p + name + "/BatchNorm/beta"), ws, prezeroed=pz)
            net_in = pair["y"][bi - 1] if bi > 0 else pair["a"]
            todo.append(conv_wgrad_args(ps, p + name + "/conv1/Conv/weights", None, net_in, dcv, 2, flags=flags))
            if bi > 0:
                g = conv_bwd_data(ps, p + name + "/conv1/Conv/weights", dcv, net_in.shape[1:3], 2, flags=flags)
            else:   # net_in = a = lrelu(input conv)
                g = conv_bwd_data(ps, p + name + "/conv1/Conv/weights", dcv, net_in.shape[1:3], 2, aux=pair["a"],
                                  mask_act=ACT_LRELU, mask_alpha=0.2, flags=flags)
        todo.append(conv_wgrad_args(ps, p + "input_stage/conv/Conv/weights", p + "input_stage/conv/Conv/biases", pair["x"], g, flags=flags))
        ds, xs, dys, dws, dbs, lxs, lys = zip(*todo)
        K.conv_wgrad_multi(list(ds), list(xs), list(dys), list(dws), list(dbs), list(lxs), list(lys))


# --------------------------------------------------------------------------------------------------
# VGG-19 feature taps -- reference lib/Teco.py:5-24, lib/ops.py:287-334 (weights frozen, main.py:322-324)
# --------------------------------------------------------------------------------------------------
VGG_TAPS = ("vgg_19/conv2/conv2_2", "vgg_19/conv3/conv3_4", "vgg_19/conv4/conv4_4", "vgg_19/conv5/conv5_4")
VGG_CPAD = pad8(3)


class VGG19:
    def __init__(self, ps):
        self.ps = ps

    def forward(self, x, keep=True, flags=0):
        """x [N,H,W,8]: VGG-preprocessed image (lib/Teco.py:9-10) zero-padded to 8 channels.
        Returns the four post-ReLU taps (un-normalised) and the activations for backward."""
        ps = self.ps
        net, acts, taps = x, [], {}
        for blk, reps, _, _ in VGG_CFG:
            for j in range(1, reps + 1):
                key = "vgg_19/conv%d/conv%d_%d" % (blk, blk, j)
                inp = net
                net = conv_fwd(ps, key + "/weights", key + "/biases", inp, 1, ACT_RELU, flags=flags)
                acts.append((key, inp, net))
                if key in VGG_TAPS:
                    taps[key] = net
            if key == VGG_TAPS[-1]:
                break
            N, H, W, Cc = net.shape
            pooled = K.maxpool2_forward(net, _empty((N, H // 2, W // 2, Cc), net.dtype, net))
            acts.append(("pool%d" % blk, net, pooled))
            net = pooled
        return taps, (acts if keep else None)

    def backward(self, acts, d_taps, flags=0):
        """d_taps: key -> gradient w.r.t. the (post-ReLU) tap.  Returns d x [N,H,W,8] (dX only: the VGG
        weights are frozen).  `g` is always the gradient w.r.t. a conv's PRE-activation: the ReLU
        derivative of the producing layer is fused into the consumer's bwd_data epilogue (conv -> conv)
        or into the max-pool backward (conv -> pool)."""
        ps = self.ps
        g = None
        for idx in range(len(acts) - 1, -1, -1):
            key, inp, out = acts[idx]
            if key.startswith("pool"):
                # the pooled tensor `inp` is a tap when the conv before it is one: its loss gradient joins the routed
                # pool gradient inside the same kernel, (route(g) + d_tap) * relu'(inp)
                tap = d_taps.get(acts[idx - 1][0]) if idx > 0 else None
                g = K.maxpool2_backward(inp, g, torch.empty_like(inp), ACT_RELU, 0.0, add=tap)
                continue
            if key in d_taps and g is None:        # the last tap (conv5_4) has no pool behind it
                g = K.act_backward(d_taps[key], out, torch.empty_like(out), ACT_RELU)
            producer_is_conv = idx > 0 and not acts[idx - 1][0].startswith("pool")
            g = conv_bwd_data(ps, key + "/weights", g, inp.shape[1:3], 1, aux=inp if producer_is_conv else None,
                              mask_act=ACT_RELU if producer_is_conv else ACT_NONE, flags=flags)
        return g

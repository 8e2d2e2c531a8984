"""Tensor-level (no autograd) wrappers of the C ABI: torch is only the owner of device memory
and of the HIP stream.  Every function enqueues on torch's current stream (hipGraph capturable)."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import CONV_COEXIST, ConvDesc, TG_BF16, TG_F32, check, lib  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt(t):
    if t.dtype == torch.float32:
        return TG_F32
    if t.dtype == torch.bfloat16:
        return TG_BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.TecoHipError("tensor is not on the GPU: tecogan_amd has no CPU path")
    if not t.is_contiguous():
        raise L.TecoHipError("tensor must be contiguous (NHWC)")
    return C.c_void_p(t.data_ptr())


def same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_desc(N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad_t, pad_l, mode, in_dt, out_dt,
              act=0, act_alpha=0.0, mask_act=0, mask_alpha=0.0, flags=0):
    return ConvDesc(N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad_t, pad_l, mode, in_dt, out_dt,
                    act, act_alpha, mask_act, mask_alpha, flags)


def conv_forward(desc, x, w, bias, res, aux, out):
    check(lib().tg_conv_forward(C.byref(desc), _p(x), _p(w), _p(bias), _p(res), _p(aux), _p(out), _stream()),
          "tg_conv_forward")
    return out


def conv_wgrad(desc, x, y, dw, dbias, ldx=0, ldy=0):
    check(lib().tg_conv_wgrad(C.byref(desc), _p(x), dt(x), ldx, _p(y), dt(y), ldy, _p(dw), _p(dbias), _stream()),
          "tg_conv_wgrad")


def conv_wgrad_grouped(desc, xs, ys, dws, dbiases, ldx=0, ldy=0):
    """Weight gradients of len(xs) layers of identical geometry in one call (one launch for bf16 3x3 stride-1 layers)."""
    G = len(xs)
    assert G >= 1 and len(ys) == G and len(dws) == G and (dbiases is None or len(dbiases) == G)

    def table(ts):
        return (C.c_void_p * G)(*[_p(t) if t is not None else None for t in ts])
    tx, ty, tw = table(xs), table(ys), table(dws)
    tb = table(dbiases) if dbiases is not None else None
    check(lib().tg_conv_wgrad_grouped(C.byref(desc), G, tx, dt(xs[0]), ldx, ty, dt(ys[0]), ldy, tw, tb, _stream()),
          "tg_conv_wgrad_grouped")


def conv_wgrad_grouped_plus(desc, xs, ys, dws, dbiases, extra, ldx=0, ldy=0):
    """conv_wgrad_grouped plus one more layer of the same spatial geometry / output width with fewer input channels, in the same
    launch where the transpose-read kernel applies.  extra = (x, ldx, cin, dy, dW, dbias)."""
    G = len(xs)
    assert G >= 1 and len(ys) == G and len(dws) == G and (dbiases is None or len(dbiases) == G)

    def table(ts):
        return (C.c_void_p * G)(*[_p(t) if t is not None else None for t in ts])
    tx, ty, tw = table(xs), table(ys), table(dws)
    tb = table(dbiases) if dbiases is not None else None
    xe, ldxe, cine, ye, dwe, dbe = extra
    check(lib().tg_conv_wgrad_grouped_plus(C.byref(desc), G, tx, dt(xs[0]), ldx, ty, dt(ys[0]), ldy, tw, tb, _p(xe), ldxe, cine, _p(ye),
                                           _p(dwe), _p(dbe), _stream()), "tg_conv_wgrad_grouped_plus")


def conv_wgrad_multi(descs, xs, ys, dws, dbiases, ldxs, ldys):
    """Weight gradients of len(xs) layers of DIFFERENT geometry in one call (one launch for bf16 3x3 stride-1 layers)."""
    G = len(xs)
    assert G >= 1 and len(descs) == G and len(ys) == G and len(dws) == G and len(dbiases) == G
    arr = (ConvDesc * G)(*descs)

    def table(ts):
        return (C.c_void_p * G)(*[_p(t) if t is not None else None for t in ts])
    lx, ly = (C.c_int * G)(*ldxs), (C.c_int * G)(*ldys)
    check(lib().tg_conv_wgrad_multi(arr, G, table(xs), dt(xs[0]), lx, table(ys), dt(ys[0]), ly, table(dws), table(dbiases),
                                    _stream()), "tg_conv_wgrad_multi")


def colsum(x, rows, Cn, out):
    check(lib().tg_colsum(_p(x), dt(x), rows, Cn, _p(out), _stream()), "tg_colsum")


def pack_weights(src_base, dst_base, tab, count, transpose):
    check(lib().tg_pack_weights(_p(src_base), _p(dst_base), dt(dst_base), _p(tab), count, int(transpose), _stream()),
          "tg_pack_weights")


def pack_weights_both(src_base, dst_t, dst_n, tab, count):
    check(lib().tg_pack_weights_both(_p(src_base), _p(dst_t), _p(dst_n), dt(dst_t), _p(tab), count, _stream()), "tg_pack_weights_both")


def warp_s2d_forward(pre, flow_lr, lr, out, scale, shift, warped=None):
    B, h, w, _ = lr.shape
    hf, wf = (flow_lr.shape[1], flow_lr.shape[2]) if flow_lr is not None else (h, w)
    check(lib().tg_warp_s2d_forward(_p(pre), _p(flow_lr), _p(lr), _p(out), dt(out), B, h, w, hf, wf, out.shape[3],
                                    scale, shift, _p(warped), _stream()), "tg_warp_s2d_forward")
    return out


def warp_s2d_backward(d_out, pre, flow_lr, d_pre, d_flow_lr, scale):
    B, h, w, Cpad = d_out.shape
    check(lib().tg_warp_s2d_backward(_p(d_out), dt(d_out), _p(pre), _p(flow_lr), _p(d_pre), _p(d_flow_lr), B, h, w,
                                     Cpad, scale, _stream()), "tg_warp_s2d_backward")


def warp_forward(img, flow, out):
    B, H, W, Cn = img.shape
    check(lib().tg_warp_forward(_p(img), _p(flow), _p(out), B, H, W, Cn, _stream()), "tg_warp_forward")
    return out


def warp_backward(d_out, img, flow, d_img, d_flow):
    B, H, W, Cn = img.shape
    check(lib().tg_warp_backward(_p(d_out), _p(img), _p(flow), _p(d_img), _p(d_flow), B, H, W, Cn, _stream()),
          "tg_warp_backward")


def upscale4_forward(x, out, gain=1.0):
    B, h, w, Cn = x.shape
    check(lib().tg_upscale4_forward(_p(x), _p(out), B, h, w, Cn, gain, _stream()), "tg_upscale4_forward")
    return out


def upscale4_backward(d_out, d_in, gain=1.0):
    B, h, w, Cn = d_in.shape
    check(lib().tg_upscale4_backward(_p(d_out), _p(d_in), B, h, w, Cn, gain, _stream()), "tg_upscale4_backward")
    return d_in


def maxpool2_forward(x, out):
    N, H, W, Cn = x.shape
    check(lib().tg_maxpool2_forward(_p(x), _p(out), dt(x), N, H, W, Cn, _stream()), "tg_maxpool2_forward")
    return out


def maxpool2_backward(x, d_out, d_in, act=0, alpha=0.0, add=None):
    N, H, W, Cn = x.shape
    check(lib().tg_maxpool2_backward(_p(x), _p(d_out), _p(d_in), dt(x), N, H, W, Cn, act, alpha, _p(add), _stream()),
          "tg_maxpool2_backward")
    return d_in


def upsample2_forward(x, out):
    N, H, W, Cn = x.shape
    check(lib().tg_upsample2_forward(_p(x), _p(out), dt(x), N, H, W, Cn, _stream()), "tg_upsample2_forward")
    return out


def upsample2_backward(d_out, d_in, y=None, act=0, alpha=0.0):
    N, H, W, Cn = d_in.shape
    check(lib().tg_upsample2_backward(_p(d_out), _p(d_in), dt(d_in), N, H, W, Cn, _p(y), act, alpha, _stream()),
          "tg_upsample2_backward")
    return d_in


def bicubic_add_preprocess(conv_out, gen_in, out, state=None):
    """out = (conv_out + bicubic_four(LR)) * 2 - 1; state (optional) = (out + 1) / 2; `out` may be None with a state."""
    B, h, w, Cpad = gen_in.shape
    check(lib().tg_bicubic_add_preprocess(_p(conv_out), _p(gen_in), dt(gen_in), Cpad, _p(out), _p(state), B, h, w,
                                          _stream()), "tg_bicubic_add_preprocess")
    return out if out is not None else state


def resblock(mode, x, w1, b1, w2, b2, aux1, aux2, mid, out, w_frag=False):
    """One residual block (mode 0) / its input-gradient chain (mode 1) as one launch (csrc/resblock_lat.hip; bf16, C = 64).
    w_frag: w1 / w2 are fragment-order copies (pack_weights_frag / frag_order)."""
    N, H, W, Cn = x.shape
    check(lib().tg_resblock(mode, _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(aux1), _p(aux2), _p(mid), _p(out), N, H, W, Cn,
                            dt(x), int(bool(w_frag)), _stream()), "tg_resblock")
    return out


def graph_node_count(raw_graph):
    """(nodes, kernel nodes) of a captured graph: raw_graph = torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()."""
    n, k = C.c_int(0), C.c_int(0)
    check(lib().tg_graph_node_count(C.c_void_p(int(raw_graph)), C.byref(n), C.byref(k)), "tg_graph_node_count")
    return n.value, k.value


def resblock_chain_ok(N, H, W):
    """tg_resblock_chain needs every 4x4-pixel tile's workgroup resident at once: at most one tile per compute unit."""
    return N * ((H + 3) // 4) * ((W + 3) // 4) <= torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count


def resblock_chain_scratch(N, H, W, device):
    """The exchange scratch of tg_resblock_chain (control words + granule ring), zeroed once; one per stream."""
    n = C.c_int64(0)
    check(lib().tg_resblock_chain_scratch_bytes(N, H, W, C.byref(n)), "tg_resblock_chain_scratch_bytes")
    return torch.zeros(n.value // 4, dtype=torch.int32, device=device)


class ChainArgs:
    """The per-block pointer arrays of one tg_resblock_chain call, built once (the tensors they point to are kept alive here)."""

    def __init__(self, mode, x, w1, b1, w2, b2, aux1, aux2_last, mid, out, scratch, variant=0, pre=None):
        """pre (forward only): (generator input [N,H,W,Cpad], input conv's fragment-order weights, bias, a0 out [N,H,W,64]) -- the
        input-stage conv runs in the same launch in front of the first block; `x` may then be None."""
        nb = len(w1)
        assert nb == len(w2) == len(out) and 1 <= nb <= 16
        self.keep = (x, w1, b1, w2, b2, aux1, aux2_last, mid, out, scratch, pre)
        self.pre = pre
        if x is None:
            x = pre[3]
        arr = lambda ts: None if ts is None else (C.c_void_p * nb)(*[None if t is None else _p(t) for t in ts])   # noqa: E731
        self.a = (arr(w1), arr(b1), arr(w2), arr(b2), arr(aux1), arr(mid), arr(out))
        self.mode, self.nb, self.variant = mode, nb, variant
        self.x, self.aux2, self.scratch = x, aux2_last, scratch
        N, H, W, Cn = x.shape
        for t in [x] + list(out) + [t for t in (mid or []) if t is not None] + [t for t in (aux1 or []) if t is not None]:
            assert tuple(t.shape) == (N, H, W, Cn) and t.dtype == x.dtype

    def launch(self):
        N, H, W, Cn = self.x.shape
        w1, b1, w2, b2, aux1, mid, out = self.a
        px, pw, pb, po = self.pre if self.pre is not None else (None, None, None, None)
        check(lib().tg_resblock_chain(self.mode, _p(self.x), self.nb, w1, b1, w2, b2, aux1, _p(self.aux2), mid, out, _p(self.scratch),
                                      _p(px), px.shape[-1] if px is not None else 0, _p(pw), _p(pb), _p(po),
                                      N, H, W, Cn, dt(self.x), self.variant, _stream()), "tg_resblock_chain")


def resblock_chain(mode, x, w1, b1, w2, b2, aux1, aux2_last, mid, out, scratch, variant=0, pre=None):
    """nb residual blocks (mode 0) / their input-gradient chain (mode 1) as ONE persistent launch (csrc/resblock_chain.hip):
    lists of per-block tensors in processing order, fragment-order weights.  Returns out[-1]."""
    ChainArgs(mode, x, w1, b1, w2, b2, aux1, aux2_last, mid, out, scratch, variant, pre).launch()
    return out[-1]


def resblock_plane_ok(N, H, W):
    """tg_resblock_plane: one 16x32-pixel tile per compute unit at most (every workgroup resident), and enough tiles to fill the
    chip (below that the per-block launches use more of it)."""
    nt = N * ((H + 15) // 16) * ((W + 31) // 32)
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return cus // 2 <= nt <= cus


def resblock_plane_scratch(N, H, W, device):
    """The exchange scratch of tg_resblock_plane (control words + granule ring), zeroed once; one per stream."""
    n = C.c_int64(0)
    check(lib().tg_resblock_plane_scratch_bytes(N, H, W, C.byref(n)), "tg_resblock_plane_scratch_bytes")
    return torch.zeros(n.value // 4, dtype=torch.int32, device=device)


class PlaneArgs:
    """The per-block pointer arrays of one tg_resblock_plane call, built once (the tensors they point to are kept alive here)."""

    def __init__(self, x, w1, b1, w2, b2, out, scratch, variant=0, pre=None):
        """pre: (generator input [N,H,W,Cpad], the input conv's fragment-order weights, bias) -- the input-stage conv runs in the same
        launch in front of the first block; `x` may then be None."""
        nb = len(w1)
        assert nb == len(w2) and 1 <= nb <= 16 and out.shape[-1] == 64
        assert (x is None and pre is not None and tuple(pre[0].shape[:3]) == tuple(out.shape[:3])) or (tuple(out.shape) == tuple(x.shape) and out.dtype == x.dtype)
        self.keep = (x, w1, b1, w2, b2, out, scratch, pre)
        arr = lambda ts: None if ts is None else (C.c_void_p * nb)(*[None if t is None else _p(t) for t in ts])   # noqa: E731
        self.a = (arr(w1), arr(b1), arr(w2), arr(b2))
        self.nb, self.variant, self.x, self.out, self.scratch, self.pre = nb, variant, x, out, scratch, pre

    def launch(self):
        N, H, W, Cn = self.out.shape
        w1, b1, w2, b2 = self.a
        px, pw, pb = self.pre if self.pre is not None else (None, None, None)
        check(lib().tg_resblock_plane(_p(self.x), self.nb, w1, b1, w2, b2, _p(self.out), _p(self.scratch), _p(px),
                                      px.shape[-1] if px is not None else 0, _p(pw), _p(pb), N, H, W, Cn, dt(self.out),
                                      self.variant, _stream()), "tg_resblock_plane")
        return self.out


def resblock_plane(x, w1, b1, w2, b2, out, scratch, variant=0, pre=None):
    """nb residual blocks of the stateless forward as ONE persistent launch (csrc/resblock_plane.hip); fragment-order weights."""
    return PlaneArgs(x, w1, b1, w2, b2, out, scratch, variant, pre).launch()


def conv3x3_c64_frag_ok(N, H, W):
    """The throughput regime of tg_conv3x3_c64_frag: at least 256 tiles of 8x16 pixels."""
    return N * ((H + 7) // 8) * ((W + 15) // 16) >= 256


def conv3x3_c64_frag(x, w_frag, bias, res, out, act=0, alpha=0.0):
    """out = act(conv3x3(x, W) + b) [+ res], 64 -> 64 bf16, W in fragment order (csrc/conv3x3_ws.hip)."""
    N, H, W, C = x.shape
    assert C == 64 and x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and tuple(out.shape) == tuple(x.shape)
    check(lib().tg_conv3x3_c64_frag(_p(x), _p(w_frag), _p(bias), _p(res), _p(out), N, H, W, act, alpha, _stream()),
          "tg_conv3x3_c64_frag")
    return out


def pack_wide_frag(w, w_frag, Cout, Cin, flip):
    """Fragment-order copy of a [9][Cout][Cin] bf16 conv operand (csrc/conv3x3_wr.hip); flip mirrors the taps (input gradient)."""
    assert w.dtype == torch.bfloat16 and w_frag.dtype == torch.bfloat16 and w.numel() == 9 * Cout * Cin == w_frag.numel()
    check(lib().tg_pack_wide_frag(_p(w), _p(w_frag), Cout, Cin, int(flip), _stream()), "tg_pack_wide_frag")
    return w_frag


def conv3x3_wide_frag_ok(desc, res=None):
    """What tg_conv3x3_wide_frag covers (the mirror of its argument checks, so that anything else falls back to tg_conv_forward
    instead of raising): wide 3x3 stride-1 SAME bf16 layers with a none / ReLU / LeakyReLU epilogue on images larger than 8x8, or of
    exactly 8x8 pixels (packed tiles: no residual operand)."""
    if not (desc.KH == 3 and desc.KW == 3 and desc.stride == 1 and desc.Cin % 32 == 0 and desc.Cin > 64 and desc.Cout % 64 == 0
            and desc.in_dtype == 1 and desc.out_dtype == 1 and desc.act < L.ACT_TANH):
        return False
    if desc.Hin == 8 and desc.Win == 8:
        return res is None
    return desc.Hin > 8 and desc.Win > 8


def conv3x3_wide_frag(desc, x, w_frag, bias, res, aux, out, tile_rows=0, ksplit=0):
    """tg_conv_forward's result for a wide 3x3 layer, weights streamed into registers from the fragment-order copy."""
    check(lib().tg_conv3x3_wide_frag(C.byref(desc), _p(x), _p(w_frag), _p(bias), _p(res), _p(aux), _p(out), tile_rows, ksplit,
                                     _stream()), "tg_conv3x3_wide_frag")
    return out


BN_STAT_REPLICAS = 16      # TG_BN_STAT_REPLICAS of include/tecogan_hip.h


def pack_taps_frag(w, w_frag, taps, Cout, Cin):
    """Fragment-order copy of a [taps][Cout][Cin] bf16 conv operand (csrc/conv4x4s2.hip)."""
    assert w.dtype == torch.bfloat16 and w_frag.dtype == torch.bfloat16 and w.numel() == taps * Cout * Cin == w_frag.numel()
    check(lib().tg_pack_taps_frag(_p(w), _p(w_frag), taps, Cout, Cin, _stream()), "tg_pack_taps_frag")
    return w_frag


def pack_taps_frag_multi(src_t, src_n, dst, tab, count):
    """Fragment-order copies of `count` operands taken from the two flat compute-copy buffers, one launch (see the header)."""
    check(lib().tg_pack_taps_frag_multi(_p(src_t), _p(src_n), _p(dst), _p(tab), count, _stream()), "tg_pack_taps_frag_multi")


def conv4x4s2_frag_ok(desc, bias=None, aux=None):
    """What tg_conv4x4s2_frag covers (the mirror of its argument checks): the discriminator's 4x4 stride-2 bf16 convs (even sizes;
    epilogue bias, none / ReLU / LeakyReLU, residual -- no mask operand) and their input gradients (residual and activation mask
    only: no bias, no activation)."""
    if not (desc.KH == 4 and desc.KW == 4 and desc.stride == 2 and desc.pad_t == 1 and desc.pad_l == 1 and desc.Cin % 32 == 0
            and desc.Cout % 64 == 0 and desc.in_dtype == 1 and desc.out_dtype == 1):
        return False
    if desc.mode == 0:
        return (aux is None and desc.act < L.ACT_TANH
                and desc.Hin % 2 == 0 and desc.Win % 2 == 0 and desc.Hout * 2 == desc.Hin and desc.Wout * 2 == desc.Win)
    return bias is None and desc.act == L.ACT_NONE and desc.Hout == 2 * desc.Hin and desc.Wout == 2 * desc.Win


def conv4x4s2_frag(desc, x, w_frag, bias, res, aux, out, bn_stats=None):
    """tg_conv_forward's result for a 4x4 stride-2 layer (either direction), weights streamed into registers.  bn_stats (forward,
    [2][Cout] fp32, zero on entry) += the per-channel mean and second moment of the result: bn_lrelu_forward(..., prezeroed=2)."""
    check(lib().tg_conv4x4s2_frag(C.byref(desc), _p(x), _p(w_frag), _p(bias), _p(res), _p(aux), _p(out), _p(bn_stats), _stream()),
          "tg_conv4x4s2_frag")
    return out


def resblock_c64_thr(x, w1_frag, b1, w2_frag, b2, out):
    """out = x + conv3x3(relu(conv3x3(x, W1) + b1), W2) + b2 in one launch, throughput regime (csrc/resblock_thr.hip; bf16, 64 ch)."""
    N, H, W, C = x.shape
    assert C == 64 and x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and tuple(out.shape) == tuple(x.shape)
    check(lib().tg_resblock_c64_thr(_p(x), _p(w1_frag), _p(b1), _p(w2_frag), _p(b2), _p(out), N, H, W, _stream()), "tg_resblock_c64_thr")
    return out


def hr_tail_backward(d_frame, scale, w_out, t2, w_tr_frag, t1, g_out, g_t2, g_t1):
    """Frame gradient -> g_out, g_t2, g_t1 in one launch (csrc/hr_bwd_lat.hip; bf16): see tg_hr_tail_backward."""
    N, H2, W2, C = t1.shape
    assert C == 64 and t1.dtype == torch.bfloat16 and d_frame.dtype == torch.float32
    check(lib().tg_hr_tail_backward(_p(d_frame), float(scale), _p(w_out), _p(t2), _p(w_tr_frag), _p(t1), _p(g_out), _p(g_t2),
                                    _p(g_t1), N, H2, W2, _stream()), "tg_hr_tail_backward")
    return g_t1


def deconv_lat_forward(x, w_frag, bias, out):
    """relu(conv2d_transpose k3 s2 (x) + b) in the latency regime (csrc/hr_fwd_lat.hip; bf16, 64 channels)."""
    N, H1, W1, C = x.shape
    assert C == 64 and x.dtype == torch.bfloat16
    check(lib().tg_deconv_lat_forward(_p(x), _p(w_frag), _p(bias), _p(out), N, H1, W1, _stream()), "tg_deconv_lat_forward")
    return out


def deconv_lat_backward(dy, w_frag, aux, dx):
    """Input gradient of the k3 s2 transposed conv in the latency regime (csrc/hr_bwd_lat.hip): dy [N,2H,2W,64] -> dx [N,H,W,64]."""
    N, H, W, C = dx.shape
    assert C == 64 and dy.dtype == torch.bfloat16 and tuple(dy.shape) == (N, 2 * H, 2 * W, 64)
    check(lib().tg_deconv_lat_backward(_p(dy), _p(w_frag), _p(aux), _p(dx), N, H, W, _stream()), "tg_deconv_lat_backward")
    return dx


def hr_tail_train(t1, w2_frag, b2, w3, b3, gen_in, t2, frame, state=None):
    """Second transposed conv (t2 stored unless None) + output conv + bicubic skip + value range(s) in one launch: the training
    recurrence's tail and, with thousands of tiles, the inference frame's (persistent launch)."""
    N, H1, W1, C = t1.shape
    assert C == 64 and t1.dtype == torch.bfloat16 and gen_in.dtype == torch.bfloat16
    assert all(o is None or o.dtype == torch.float32 for o in (frame, state))
    check(lib().tg_hr_tail_train(_p(t1), _p(w2_frag), _p(b2), _p(w3), _p(b3), _p(gen_in), gen_in.shape[-1], _p(t2), _p(frame),
                                 _p(state), N, H1, W1, _stream()), "tg_hr_tail_train")
    return frame if frame is not None else state


def pack_weights_frag(src_base, dst_t, dst_n, tab, count):
    check(lib().tg_pack_weights_frag(_p(src_base), _p(dst_t), _p(dst_n), _p(tab), count, _stream()), "tg_pack_weights_frag")


def frag_order(w_rows):
    """[9][64 rows][64 k] (the operand layout of tg_conv_forward) -> the fragment order of tg_resblock(w_frag=1):
    [2 tap + kk][wave][lane = 16 fg + frow][j] = W[tap][16 wave + frow][32 kk + 8 fg + j].  (torch view / permute; tests and
    one-off callers -- the training engine's copies come from tg_pack_weights_frag.)"""
    return w_rows.reshape(9, 4, 16, 2, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)


def act_backward(d_out, y, d_in, act=0, alpha=0.0, scale=1.0):
    check(lib().tg_act_backward(_p(d_out), _p(y), _p(d_in), dt(d_out), dt(d_in), d_out.numel(), act, alpha, scale,
                                _stream()), "tg_act_backward")
    return d_in


def concat2_pad(a, b, out, scale=1.0):
    Ca, Cb, Cpad = a.shape[-1], (b.shape[-1] if b is not None else 0), out.shape[-1]
    check(lib().tg_concat2_pad(_p(a), Ca, _p(b), Cb, _p(out), dt(out), Cpad, out.numel() // Cpad, scale, _stream()),
          "tg_concat2_pad")
    return out


def lincomb(a, b, out, alpha, beta=0.0, accumulate=False):
    check(lib().tg_lincomb(_p(a), _p(b), _p(out), a.numel(), alpha, beta, int(accumulate), _stream()), "tg_lincomb")
    return out


def schedule_step(state, hyper, nopt, gated_opt, t_balance, beta1, beta2, eps):
    check(lib().tg_schedule_step(_p(state), _p(hyper), nopt, gated_opt, _p(t_balance), beta1, beta2, eps, _stream()),
          "tg_schedule_step")


def bn_lrelu_forward(x, y, beta, eps, alpha, stats, moving, prezeroed=False):
    Cn = x.shape[-1]
    check(lib().tg_bn_lrelu_forward(_p(x), _p(y), dt(x), x.numel() // Cn, Cn, _p(beta), eps, alpha, _p(stats),
                                    _p(moving), int(prezeroed), _stream()), "tg_bn_lrelu_forward")
    return y


def bn_lrelu_backward(x, y, d_y, d_x, stats, eps, alpha, d_beta, ws, prezeroed=False):
    Cn = x.shape[-1]
    check(lib().tg_bn_lrelu_backward(_p(x), _p(y), _p(d_y), _p(d_x), dt(x), x.numel() // Cn, Cn, _p(stats), eps,
                                     alpha, _p(d_beta), _p(ws), int(prezeroed), _stream()), "tg_bn_lrelu_backward")
    return d_x


def adam_tf(p, g, m, v, hyper, grad_scale=1.0):
    check(lib().tg_adam_tf(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), grad_scale, _stream()), "tg_adam_tf")


def sum_sq_diff(a, b, scale, out):
    check(lib().tg_sum_sq_diff(_p(a), _p(b), dt(a), a.numel(), scale, _p(out), _stream()), "tg_sum_sq_diff")


def sum_abs_diff(a, b, scale, out):
    check(lib().tg_sum_abs_diff(_p(a), _p(b), dt(a), a.numel(), scale, _p(out), _stream()), "tg_sum_abs_diff")


# ---- TecoGAN losses / discriminator input -------------------------------------------------------
def pingpong(gen, d_gen, T, npair, loss_scale, grad_scale, loss):
    check(lib().tg_pingpong(_p(gen), _p(d_gen), T, npair, gen.numel() // T, loss_scale, grad_scale, _p(loss),
                            _stream()), "tg_pingpong")


def vgg_preprocess_forward(x, out):
    Cpad = out.shape[-1]
    check(lib().tg_vgg_preprocess_forward(_p(x), _p(out), dt(out), out.numel() // Cpad, Cpad, _stream()),
          "tg_vgg_preprocess_forward")
    return out


def vgg_preprocess_backward(d_out, d_x):
    Cpad = d_out.shape[-1]
    check(lib().tg_vgg_preprocess_backward(_p(d_out), dt(d_out), _p(d_x), d_out.numel() // Cpad, Cpad, _stream()),
          "tg_vgg_preprocess_backward")


def cosine_loss(g, t, cos_scale, grad_scale, cos_sum, d_g):
    Cn = g.shape[-1]
    check(lib().tg_cosine_loss(_p(g), _p(t), dt(g), g.numel() // Cn, Cn, cos_scale, grad_scale, _p(cos_sum), _p(d_g),
                               _stream()), "tg_cosine_loss")


def l1_loss(r, f, loss_scale, grad_scale, loss, d_f, grad_scale_dev=None):
    check(lib().tg_l1_loss(_p(r), _p(f), dt(r), r.numel(), loss_scale, grad_scale, _p(grad_scale_dev), _p(loss), _p(d_f),
                           _stream()), "tg_l1_loss")


def gan_losses(real, fake, eps, adv_weight, out, d_real_D, d_fake_D, d_fake_G, adv_scale_dev=None):
    check(lib().tg_gan_losses(_p(real), _p(fake), real.numel(), eps, adv_weight, _p(adv_scale_dev), _p(out), _p(d_real_D),
                              _p(d_fake_D), _p(d_fake_G), _stream()), "tg_gan_losses")


def dt_ratio(state, r0, add, rmax, out):
    """out[0] = min(rmax, r0 + add * state[0]) from the device-side global step (lib/Teco.py:379-380)."""
    check(lib().tg_dt_ratio(_p(state), r0, add, rmax, _p(out), _stream()), "tg_dt_ratio")


def _int_array(v):
    return (C.c_int * len(v))(*v)


def pack_d_input_forward(frames, lr, flow_pre, flow_nxt, idx_pre, idx_nxt, out, B, h, w, off, merge):
    nt = len(idx_pre)
    check(lib().tg_pack_d_input_forward(_p(frames), _p(lr), _p(flow_pre), _p(flow_nxt), _int_array(idx_pre),
                                        _int_array(idx_nxt), _p(out), dt(out), B, h, w, nt, off, int(merge),
                                        out.shape[-1], _stream()), "tg_pack_d_input_forward")
    return out


def pack_d_input_backward(d_out, frames, flow_pre, flow_nxt, idx_pre, idx_nxt, d_frames, B, h, w, off, merge):
    nt = len(idx_pre)
    check(lib().tg_pack_d_input_backward(_p(d_out), dt(d_out), _p(frames), _p(flow_pre), _p(flow_nxt),
                                         _int_array(idx_pre), _int_array(idx_nxt), _p(d_frames), B, h, w, nt, off,
                                         int(merge), d_out.shape[-1], _stream()), "tg_pack_d_input_backward")


def seq_gather(src, dst, idx):
    """src [B,T0,...] -> dst [T,B,...] with dst[t] = src[:, idx[t]] (ping-pong order + frame-major layout)."""
    B, T0 = src.shape[0], src.shape[1]
    T = len(idx)
    check(lib().tg_seq_gather(_p(src), _p(dst), B, T0, T, src[0, 0].numel(), _int_array(idx), _stream()), "tg_seq_gather")
    return dst


def affine(x, out, scale, shift):
    check(lib().tg_affine(_p(x), _p(out), x.numel(), scale, shift, _stream()), "tg_affine")
    return out


def gauss_down4_preprocess(hr, weights2d, lr, target=None, border=0):
    """hr [N,H,W,3] fp32 -> lr (Gaussian k x k, stride 4, VALID); target (optional) = 2*hr[crop]-1 in the same launch."""
    N, H, W, _ = hr.shape
    k = int(round(len(weights2d) ** 0.5))
    wts = (C.c_float * (k * k))(*[float(v) for v in weights2d])
    check(lib().tg_gauss_down4_preprocess(_p(hr), _p(lr), _p(target), N, H, W, k, wts, border, _stream()),
          "tg_gauss_down4_preprocess")
    return lr


def frame_to_u8(frame, out, bgr=False):
    """frame [...,3] fp32 in [0,1] -> out uint8, truncating like save_img (lib/ops.py:521-523)."""
    check(lib().tg_frame_to_u8(_p(frame), _p(out), frame.numel() // 3, int(bgr), _stream()), "tg_frame_to_u8")
    return out


# ---- built-in launch profiler (csrc/runtime.hip) -----------------------------------------------------
def prof_enable(on=True):
    check(lib().tg_prof_enable(int(bool(on))), "tg_prof_enable")


def prof_stamp(dst):
    """dst: one int64 element (device) <- the device wall clock (100 MHz) when the current stream gets here; capturable."""
    check(lib().tg_prof_stamp(_p(dst), _stream()), "tg_prof_stamp")


def prof_collect(max_entries=256):
    """Synchronise, aggregate and clear the launch records: list of dicts sorted by total time (descending)."""
    buf = (L.ProfEntry * max_entries)()
    n = C.c_int(0)
    check(lib().tg_prof_collect(buf, max_entries, C.byref(n)), "tg_prof_collect")
    out = [dict(name=buf[i].name.decode(), calls=int(buf[i].calls), total_us=float(buf[i].total_us),
                flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(min(n.value, max_entries))]
    out.sort(key=lambda e: -e["total_us"])
    return out

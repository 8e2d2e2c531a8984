"""Kernel-level parity: every HIP entry point of the C ABI against the CPU oracle (fp32: 1e-4 abs/rel
unless stated; index shuffles bit-exact).  These call through libtecogan_hip.so via ctypes."""
import os

import pytest
import torch

import oracle.ops as O
from tecogan_amd import kernels as K
from tecogan_amd._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, TG_BF16, TG_F32)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(a, b, tol=1e-4, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(1.0, ref), "%s: max err %g (ref max %g)" % (what, err, ref)


def tight(a, b, what="", rel=4e-3, abs_=2e-4):
    """The bound of a bf16 kernel that multiplies exactly, accumulates in fp32 and rounds ONCE: half a bf16 step of the result
    (2^-9 relative, doubled for the epilogue's own operations) plus the fp32 summation-order noise, PER ELEMENT against the oracle
    on the bf16-rounded operands (no tolerance relative to the tensor maximum on a timed-path kernel)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    err = (a - b).abs()
    bound = rel * b.abs() + abs_
    bad = err > bound
    assert not bad.any(), "%s: %d of %d elements beyond %g |ref| + %g; worst err %g at ref %g (%.1f x its bound)" % (
        what, int(bad.sum()), bad.numel(), rel, abs_, float(err[bad].max()), float(b[bad][err[bad].argmax()]),
        float((err / bound).max()))


def tdt(dtype):
    return TG_F32 if dtype == torch.float32 else TG_BF16


def run_conv_fwd(x, w_hwio, b, stride, act=ACT_NONE, alpha=0.0, res=None, dtype=torch.float32, out_dtype=None):
    """conv2 forward through the gather engine; w operand = [tap][Cout][Cin]."""
    out_dtype = out_dtype or dtype
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w_hwio.shape
    Ho, pt = K.same_pad(H, kh, stride)
    Wo, pl = K.same_pad(W, kw, stride)
    wt = w_hwio.permute(0, 1, 3, 2).reshape(kh * kw, Cout, Cin).contiguous().to(DEV, dtype)
    out = torch.empty(N, Ho, Wo, Cout, device=DEV, dtype=out_dtype)
    d = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pt, pl, 0, tdt(dtype), tdt(out_dtype), act, alpha)
    K.conv_forward(d, x.to(DEV, dtype).contiguous(), wt, None if b is None else b.to(DEV), 
                   None if res is None else res.to(DEV, out_dtype).contiguous(), None, out)
    return out


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s
    (2, 8, 8, 64, 64, 3, 1),
    (1, 5, 7, 8, 32, 3, 1),
    (2, 9, 13, 56, 64, 3, 1),
    (1, 16, 16, 64, 3, 3, 1),
    (3, 6, 6, 32, 2, 3, 1),
    (2, 16, 16, 64, 64, 4, 2),
    (1, 9, 11, 32, 128, 4, 2),
    (1, 4, 4, 256, 1, 1, 1),
    (1, 12, 12, 128, 256, 3, 1),
    (1, 33, 35, 64, 64, 3, 1),
    (1, 6, 6, 51, 64, 3, 1),     # scalar (non-vector) channel path
    # halo-tile 3x3 kernel (conv3x3.hip): persistent tile loop, partial edge tiles, every tile height
    (1, 270, 250, 64, 64, 3, 1),   # TH=16, 272 tiles > 256 workgroups, weights stationary
    (1, 130, 130, 128, 64, 3, 1),  # TH=4, multi-chunk (weights re-staged per chunk), persistent
    (2, 64, 64, 64, 128, 3, 1),    # two output-channel tiles
    (4, 32, 32, 64, 64, 3, 1),     # TH=2: the generator's training shape
    (3, 128, 96, 32, 32, 3, 1),    # BN=32
    (1, 128, 128, 64, 3, 3, 1),    # BN=16 (generator output conv)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_fp32(case):
    N, H, W, Cin, Cout, k, s = case
    x, w, b = rnd(N, H, W, Cin, seed=1), rnd(k, k, Cin, Cout, seed=2, scale=0.2), rnd(Cout, seed=3)
    ref = O.conv2(x, w, b, s)
    got = run_conv_fwd(x, w, b, s)
    close(got, ref, 1e-4, "conv fwd %s" % (case,))


def test_conv_forward_epilogues():
    x, w, b = rnd(2, 8, 8, 64, seed=1), rnd(3, 3, 64, 64, seed=2, scale=0.2), rnd(64, seed=3)
    res = rnd(2, 8, 8, 64, seed=4)
    pre = O.conv2(x, w, b, 1)
    close(run_conv_fwd(x, w, b, 1, ACT_RELU), torch.relu(pre), 1e-4, "relu")
    close(run_conv_fwd(x, w, b, 1, ACT_LRELU, 0.2), O.lrelu(pre, 0.2), 1e-4, "lrelu")
    close(run_conv_fwd(x, w, b, 1, ACT_TANH, 24.0), torch.tanh(pre) * 24.0, 1e-4, "tanh")
    close(run_conv_fwd(x, w, b, 1, ACT_SIGMOID), torch.sigmoid(pre), 1e-4, "sigmoid")
    close(run_conv_fwd(x, w, b, 1, ACT_NONE, 0.0, res), pre + res, 1e-4, "residual")


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_bf16(case):
    N, H, W, Cin, Cout, k, s = case
    x, w, b = rnd(N, H, W, Cin, seed=1), rnd(k, k, Cin, Cout, seed=2, scale=0.2), rnd(Cout, seed=3)
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    ref = O.conv2(xb, wb, b, s)
    got = run_conv_fwd(x, w, b, s, dtype=torch.bfloat16, out_dtype=torch.float32)
    close(got, ref, 2e-4, "conv bf16 fwd (f32 out) %s" % (case,))
    got = run_conv_fwd(x, w, b, s, dtype=torch.bfloat16)
    tight(got, ref, "conv bf16 fwd %s" % (case,), abs_=2e-4 * max(1.0, float(ref.abs().max()) / 4.0))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bwd_data_and_wgrad_fp32(case):
    N, H, W, Cin, Cout, k, s = case
    x = rnd(N, H, W, Cin, seed=1).requires_grad_()
    w = rnd(k, k, Cin, Cout, seed=2, scale=0.2).requires_grad_()
    b = rnd(Cout, seed=3).requires_grad_()
    y = O.conv2(x, w, b, s)
    gy = rnd(*y.shape, seed=5)
    y.backward(gy)
    Ho, pt = K.same_pad(H, k, s)
    Wo, pl = K.same_pad(W, k, s)
    # bwd_data = transposed mode with the HWIO weights as stored: operand [tap][n=Cin][k=Cout]
    d = K.conv_desc(N, Ho, Wo, Cout, H, W, Cin, k, k, s, pt, pl, 1, TG_F32, TG_F32)
    dx = torch.empty(N, H, W, Cin, device=DEV)
    K.conv_forward(d, gy.to(DEV), w.detach().reshape(k * k, Cin, Cout).contiguous().to(DEV), None, None, None, dx)
    close(dx, x.grad, 1e-4, "bwd_data %s" % (case,))
    dg = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, TG_F32, TG_F32)
    dw = torch.zeros(k, k, Cin, Cout, device=DEV)
    db = torch.zeros(Cout, device=DEV)
    K.conv_wgrad(dg, x.detach().to(DEV), gy.to(DEV), dw, db)
    close(dw, w.grad, 2e-4, "wgrad %s" % (case,))
    close(db, b.grad, 2e-4, "bgrad %s" % (case,))


def test_conv_bwd_mask_and_residual_epilogue():
    """dx = (bwd_data(dy) + res) * relu'(aux): the fused form used between res-block layers."""
    N, H, W, Cc = 2, 8, 8, 64
    w = rnd(3, 3, Cc, Cc, seed=2, scale=0.2)
    gy, res, aux = rnd(N, H, W, Cc, seed=5), rnd(N, H, W, Cc, seed=6), rnd(N, H, W, Cc, seed=7)
    x = torch.zeros(N, H, W, Cc, requires_grad=True)
    O.conv2(x, w, None, 1).backward(gy)
    for act, alpha, fac in ((ACT_RELU, 0.0, (aux > 0).float()), (ACT_LRELU, 0.2, torch.where(aux > 0, 1.0, 0.2))):
        d = K.conv_desc(N, H, W, Cc, H, W, Cc, 3, 3, 1, 1, 1, 1, TG_F32, TG_F32, 0, 0.0, act, alpha)
        dx = torch.empty(N, H, W, Cc, device=DEV)
        K.conv_forward(d, gy.to(DEV), w.reshape(9, Cc, Cc).contiguous().to(DEV), None, res.to(DEV), aux.to(DEV), dx)
        close(dx, (x.grad + res) * fac, 1e-4, "mask epilogue act=%d" % act)


@pytest.mark.parametrize("case", [(2, 30, 45, 64), (1, 128, 128, 64), (3, 9, 70, 128), (2, 17, 33, 32)])
def test_conv3x3_8channel_tap_packed_kernel(case):
    """bf16 8-channel inputs take the tap-packed kernel (csrc/conv3x3.hip conv3x3_c8_kernel): forward with bias +
    LeakyReLU, and the input-gradient form (mirrored taps, + residual, * relu' mask) of the generator's output conv."""
    N, H, W, Cout = case
    x, w, b = rnd(N, H, W, 8, seed=1), rnd(3, 3, 8, Cout, seed=2, scale=0.3), rnd(Cout, seed=3)
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    ref = O.lrelu(O.conv2(xb, wb, b, 1), 0.2)
    got = run_conv_fwd(x, w, b, 1, ACT_LRELU, 0.2, dtype=torch.bfloat16)
    tight(got, ref, "c8 fwd %s" % (case,))
    # input-gradient form: y = conv(z, w2) with z [N,H,W,Cout] and 8 output channels; dz = bwd(gy) [+ res] * relu'(aux)
    w2 = rnd(3, 3, Cout, 8, seed=4, scale=0.3).bfloat16().float()
    gy = rnd(N, H, W, 8, seed=5).bfloat16().float()
    res, aux = rnd(N, H, W, Cout, seed=6).bfloat16().float(), rnd(N, H, W, Cout, seed=7).bfloat16().float()
    z = torch.zeros(N, H, W, Cout, requires_grad=True)
    O.conv2(z, w2, None, 1).backward(gy)
    d = K.conv_desc(N, H, W, 8, H, W, Cout, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    dz = torch.empty(N, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    K.conv_forward(d, gy.to(DEV, torch.bfloat16), w2.reshape(9, Cout, 8).contiguous().to(DEV, torch.bfloat16), None,
                   res.to(DEV, torch.bfloat16), aux.to(DEV, torch.bfloat16), dz)
    tight(dz, (z.grad + res) * (aux > 0).float(), "c8 bwd-form %s" % (case,))


@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 64), (1, 5, 7, 64, 64), (1, 16, 16, 32, 64)])
def test_deconv_fwd_bwd_fp32(shape):
    N, H, W, Cin, Cout = shape
    x = rnd(N, H, W, Cin, seed=1).requires_grad_()
    w = rnd(3, 3, Cout, Cin, seed=2, scale=0.2).requires_grad_()      # TF conv2d_transpose layout
    b = rnd(Cout, seed=3).requires_grad_()
    y = O.conv2_tran(x, w, b, 2)
    gy = rnd(*y.shape, seed=5)
    y.backward(gy)
    # forward: transposed mode, pad 0, weights as stored ([tap][Cout][Cin])
    d = K.conv_desc(N, H, W, Cin, 2 * H, 2 * W, Cout, 3, 3, 2, 0, 0, 1, TG_F32, TG_F32, ACT_NONE)
    out = torch.empty(N, 2 * H, 2 * W, Cout, device=DEV)
    K.conv_forward(d, x.detach().to(DEV), w.detach().reshape(9, Cout, Cin).contiguous().to(DEV), b.detach().to(DEV),
                   None, None, out)
    close(out, y, 1e-4, "deconv fwd")
    # bwd_data: gather s2 pad 0 over dy with operand [tap][n=Cin][k=Cout]
    dg = K.conv_desc(N, 2 * H, 2 * W, Cout, H, W, Cin, 3, 3, 2, 0, 0, 0, TG_F32, TG_F32)
    dx = torch.empty(N, H, W, Cin, device=DEV)
    K.conv_forward(dg, gy.to(DEV), w.detach().permute(0, 1, 3, 2).reshape(9, Cin, Cout).contiguous().to(DEV), None,
                   None, None, dx)
    close(dx, x.grad, 1e-4, "deconv bwd_data")
    # wgrad: X = dy (gathered), Y = x -> [tap][Cout][Cin]; bias grad = colsum(dy)
    dw = torch.zeros(3, 3, Cout, Cin, device=DEV)
    K.conv_wgrad(dg, gy.to(DEV), x.detach().to(DEV), dw, None)
    close(dw, w.grad, 2e-4, "deconv wgrad")
    db = torch.zeros(Cout, device=DEV)
    K.colsum(gy.to(DEV), gy.numel() // Cout, Cout, db)
    close(db, b.grad, 2e-4, "deconv bgrad")


def test_wgrad_bf16_inputs():
    N, H, W, Cin, Cout = 2, 8, 8, 64, 64
    x, gy = rnd(N, H, W, Cin, seed=1).bfloat16(), rnd(N, H, W, Cout, seed=5).bfloat16()
    xr = x.float().requires_grad_()
    w = torch.zeros(3, 3, Cin, Cout, requires_grad=True)
    O.conv2(xr, w, None, 1).backward(gy.float())
    d = K.conv_desc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    dw = torch.zeros(3, 3, Cin, Cout, device=DEV)
    K.conv_wgrad(d, x.to(DEV), gy.to(DEV), dw, None)
    close(dw, w.grad, 2e-4, "wgrad bf16 operands")


# --------------------------------------------------------------------------------------------
def oracle_gen_input(pre, flow_lr, lr, scale, shift, cpad):
    B, h, w, _ = lr.shape
    if pre is None:
        s2d = torch.zeros(B, h, w, 48)
    else:
        hf, wf = flow_lr.shape[1:3]
        fl = flow_lr
        if hf < h:     # main.py:212 tf.pad(..., "SYMMETRIC"): mirror including the edge row
            fl = torch.cat((fl, fl[:, hf - (h - hf):].flip(1)), 1)
        if wf < w:
            fl = torch.cat((fl, fl[:, :, wf - (w - wf):].flip(2)), 2)
        warped = O.dense_image_warp(pre, O.upscale_four(fl * 4.0))
        s2d = O.space_to_depth4(warped * scale + shift)
    return torch.cat((lr, s2d, torch.zeros(B, h, w, cpad - 51)), -1)


@pytest.mark.parametrize("B,h,w", [(2, 8, 8), (1, 5, 9), (4, 32, 32)])
def test_warp_s2d_forward_backward(B, h, w):
    pre = rnd(B, 4 * h, 4 * w, 3, seed=1).requires_grad_()
    flow = rnd(B, h, w, 2, seed=2, scale=3.0).requires_grad_()
    lr = rnd(B, h, w, 3, seed=3)
    ref = oracle_gen_input(pre, flow, lr, 0.5, 0.5, 56)
    out = torch.empty(B, h, w, 56, device=DEV)
    K.warp_s2d_forward(pre.detach().to(DEV), flow.detach().to(DEV), lr.to(DEV), out, 0.5, 0.5)
    close(out, ref, 2e-5, "warp_s2d fwd")
    g = rnd(B, h, w, 56, seed=4)
    ref.backward(g)
    d_pre = torch.zeros(B, 4 * h, 4 * w, 3, device=DEV)
    d_flow = torch.zeros(B, h, w, 2, device=DEV)
    K.warp_s2d_backward(g.to(DEV), pre.detach().to(DEV), flow.detach().to(DEV), d_pre, d_flow, 0.5)
    close(d_pre, pre.grad, 1e-4, "warp_s2d d_pre")
    close(d_flow, flow.grad, 5e-4, "warp_s2d d_flow")


@pytest.mark.parametrize("B,h,w,kind", [(2, 8, 8, "const"), (1, 5, 9, "ramp"), (4, 32, 32, "ramp"), (1, 7, 18, "tiny"),
                                        (2, 6, 10, "steps")])
def test_warp_s2d_backward_merged_scatter_smooth_flows(B, h, w, kind):
    """The scatter of warp_s2d_backward hands tap contributions to neighbouring lanes wherever the footprints line up (smooth
    flows: 3 instead of 12 atomics per HR pixel).  Constant / slowly varying / tiny / piecewise-constant flows exercise the
    hand-over in both directions, across LR-pixel and wave boundaries, and next to places where it must NOT happen."""
    pre = rnd(B, 4 * h, 4 * w, 3, seed=1).requires_grad_()
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    if kind == "const":
        fl = torch.stack((torch.full((h, w), 0.3275), torch.full((h, w), -0.6125)), -1)
    elif kind == "ramp":
        fl = torch.stack((0.21 + 0.013 * yy - 0.007 * xx, -0.4 + 0.011 * xx + 0.005 * yy), -1)
    elif kind == "steps":
        fl = torch.stack((torch.where(xx < w // 2, 0.2625, -0.7375), torch.where(yy < h // 2, 0.5125, 1.2625)), -1)
    else:                       # (exactly integer displacements would put every query ON a cell border, where one ulp of the
        fl = torch.full((h, w, 2), 0.0125)      # blended flow decides the cell: not a property of the scatter under test)
    flow = fl.expand(B, h, w, 2).clone().requires_grad_()
    lr = rnd(B, h, w, 3, seed=3)
    ref = oracle_gen_input(pre, flow, lr, 0.5, 0.5, 56)
    g = rnd(B, h, w, 56, seed=4)
    ref.backward(g)
    for gd in (g, g.bfloat16()):
        d_pre = torch.zeros(B, 4 * h, 4 * w, 3, device=DEV)
        d_flow = torch.zeros(B, h, w, 2, device=DEV)
        K.warp_s2d_backward(gd.to(DEV), pre.detach().to(DEV), flow.detach().to(DEV), d_pre, d_flow, 0.5)
        tol = 1e-4 if gd.dtype == torch.float32 else 8e-3
        close(d_pre, pre.grad, tol, "warp_s2d d_pre (%s)" % kind)
        close(d_flow, flow.grad, 5 * tol, "warp_s2d d_flow (%s)" % kind)


def test_warp_s2d_zero_flow_is_exact_shuffle():
    """Bit-exact: zero flow -> the s2d channels are exactly the reshuffled input."""
    B, h, w = 2, 8, 8
    pre, lr = rnd(B, 4 * h, 4 * w, 3, seed=1), rnd(B, h, w, 3, seed=3)
    out = torch.empty(B, h, w, 56, device=DEV)
    K.warp_s2d_forward(pre.to(DEV), torch.zeros(B, h, w, 2, device=DEV), lr.to(DEV), out, 1.0, 0.0)
    assert torch.equal(out[..., 3:51].cpu(), O.space_to_depth4(pre))
    assert torch.equal(out[..., :3].cpu(), lr)
    assert torch.equal(out[..., 51:].cpu(), torch.zeros(B, h, w, 5))


def test_warp_s2d_first_frame_and_symmetric_pad():
    B, h, w = 1, 9, 12
    lr = rnd(B, h, w, 3, seed=3)
    out = torch.empty(B, h, w, 56, device=DEV)
    K.warp_s2d_forward(None, None, lr.to(DEV), out, 0.5, 0.5)
    close(out, oracle_gen_input(None, None, lr, 0.5, 0.5, 56), 0, "first frame")
    pre = rnd(B, 4 * h, 4 * w, 3, seed=1)
    flow = rnd(B, 8, 8, 2, seed=2, scale=2.0)     # h%8=1, w%8=4 rows/cols mirrored
    K.warp_s2d_forward(pre.to(DEV), flow.to(DEV), lr.to(DEV), out, 1.0, 0.0)
    close(out, oracle_gen_input(pre, flow, lr, 1.0, 0.0, 56), 2e-5, "symmetric-padded flow")


def test_warp_plain_forward_backward():
    B, H, W, Cc = 2, 16, 12, 3
    img = rnd(B, H, W, Cc, seed=1).requires_grad_()
    flow = rnd(B, H, W, 2, seed=2, scale=4.0).requires_grad_()
    ref = O.dense_image_warp(img, flow)
    out = torch.empty(B, H, W, Cc, device=DEV)
    K.warp_forward(img.detach().to(DEV), flow.detach().to(DEV), out)
    close(out, ref, 2e-5, "warp fwd")
    g = rnd(B, H, W, Cc, seed=3)
    ref.backward(g)
    d_img = torch.zeros(B, H, W, Cc, device=DEV)
    d_flow = torch.empty(B, H, W, 2, device=DEV)
    K.warp_backward(g.to(DEV), img.detach().to(DEV), flow.detach().to(DEV), d_img, d_flow)
    close(d_img, img.grad, 1e-4, "warp d_img")
    close(d_flow, flow.grad, 1e-4, "warp d_flow")


def test_warp_integer_flow_and_clamp():
    """KAT: integer flow shifts; out-of-range queries clamp to the edge."""
    B, H, W = 1, 8, 8
    img = rnd(B, H, W, 3, seed=1)
    flow = torch.zeros(B, H, W, 2)
    flow[..., 0], flow[..., 1] = 2.0, -1.0
    out = torch.empty(B, H, W, 3, device=DEV)
    K.warp_forward(img.to(DEV), flow.to(DEV), out)
    out = out.cpu()
    assert torch.equal(out[0, 2:, :7], img[0, :6, 1:])
    assert torch.equal(out[0, 0, :7], img[0, 0, 1:])     # clamped rows


def test_upscale4_forward_backward():
    x = rnd(2, 6, 5, 2, seed=1).requires_grad_()
    ref = O.upscale_four(x * 4.0)
    out = torch.empty(2, 24, 20, 2, device=DEV)
    K.upscale4_forward(x.detach().to(DEV), out, 4.0)
    close(out, ref, 1e-5, "upscale4 fwd")
    g = rnd(2, 24, 20, 2, seed=2)
    ref.backward(g)
    d_in = torch.empty(2, 6, 5, 2, device=DEV)
    K.upscale4_backward(g.to(DEV), d_in, 4.0)
    close(d_in, x.grad, 1e-4, "upscale4 bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 9, 7, 64)])
def test_maxpool_and_upsample2(shape, dtype):
    x0 = rnd(*shape, seed=1).to(dtype).float()
    x = x0.clone().requires_grad_()
    N, H, W, Cc = shape
    ref = O.maxpool(x)
    out = torch.empty(N, H // 2, W // 2, Cc, device=DEV, dtype=dtype)
    K.maxpool2_forward(x0.to(DEV, dtype), out)
    assert torch.equal(out.float().cpu(), ref.detach())
    g = rnd(*ref.shape, seed=2).to(dtype).float()
    ref.backward(g)
    d_in = torch.empty(*shape, device=DEV, dtype=dtype)
    K.maxpool2_backward(x0.to(DEV, dtype), g.to(DEV, dtype), d_in)
    assert torch.equal(d_in.float().cpu(), x.grad)
    x = x0.clone().requires_grad_()
    ref = O.upsample2_legacy(x)
    out = torch.empty(N, 2 * H, 2 * W, Cc, device=DEV, dtype=dtype)
    K.upsample2_forward(x0.to(DEV, dtype), out)
    close(out, ref, 1e-6 if dtype == torch.float32 else 1e-2, "upsample2 fwd")
    g = rnd(*ref.shape, seed=3).to(dtype).float()
    ref.backward(g)
    d_in = torch.empty(*shape, device=DEV, dtype=dtype)
    K.upsample2_backward(g.to(DEV, dtype), d_in)
    close(d_in, x.grad, 1e-5 if dtype == torch.float32 else 2e-2, "upsample2 bwd")


def test_bicubic_add_preprocess():
    B, h, w = 2, 7, 9
    gen_in = rnd(B, h, w, 56, seed=1)
    conv_out = rnd(B, 4 * h, 4 * w, 3, seed=2)
    ref = O.preprocess(conv_out + O.bicubic_four(gen_in[..., :3]))
    out = torch.empty(B, 4 * h, 4 * w, 3, device=DEV)
    K.bicubic_add_preprocess(conv_out.to(DEV), gen_in.to(DEV), out)
    close(out, ref, 2e-6, "bicubic epilogue")


def test_bn_lrelu_forward_backward():
    x = (rnd(6, 8, 8, 64, seed=1) * 2 + 0.7).requires_grad_()
    beta = rnd(64, seed=2).requires_grad_()
    y, mean, var = O.batchnorm(x, beta)
    y = O.lrelu(y, 0.2)
    xd = x.detach().to(DEV)
    out = torch.empty_like(xd)
    stats = torch.empty(2, 64, device=DEV)
    moving = torch.stack((torch.zeros(64), torch.ones(64))).to(DEV)
    K.bn_lrelu_forward(xd, out, beta.detach().to(DEV), 1e-3, 0.2, stats, moving)
    close(out, y, 1e-5, "bn fwd")
    close(stats[0], mean, 1e-5, "bn mean")
    close(stats[1], var, 1e-5, "bn var")
    n = 6 * 8 * 8
    close(moving[0], mean * 0.1, 1e-5, "moving mean")
    close(moving[1], 0.9 + var * (n / (n - 1)) * 0.1, 1e-5, "moving var")
    g = rnd(*y.shape, seed=3)
    y.backward(g)
    dx = torch.empty_like(xd)
    dbeta = torch.zeros(64, device=DEV)
    ws = torch.empty(2, 64, device=DEV)
    K.bn_lrelu_backward(xd, out, g.to(DEV), dx, stats, 1e-3, 0.2, dbeta, ws)
    close(dx, x.grad, 1e-4, "bn dx")
    close(dbeta, beta.grad, 1e-4, "bn dbeta")


@pytest.mark.parametrize("C", [64, 128, 256, 40])
def test_bn_lrelu_bf16(C):
    """bf16 activations: C in {64,128,256} takes the 16-byte kernels, 40 the scalar ones; reference = fp32 math on the
    same bf16-rounded operands (slim.batch_norm training mode + lrelu, reference lib/ops.py:84-90)."""
    x = (rnd(3, 7, 9, C, seed=1) * 2 + 0.7).bfloat16().float().requires_grad_()
    beta = rnd(C, seed=2).requires_grad_()
    y, mean, var = O.batchnorm(x, beta)
    y = O.lrelu(y, 0.2)
    xd = x.detach().to(DEV, torch.bfloat16)
    out = torch.empty_like(xd)
    stats = torch.empty(2, C, device=DEV)
    K.bn_lrelu_forward(xd, out, beta.detach().to(DEV), 1e-3, 0.2, stats, None)
    close(stats[0], mean, 1e-4, "bn mean bf16")
    close(stats[1], var, 1e-4, "bn var bf16")
    tight(out, y, "bn fwd bf16", abs_=1e-3)       # (+ the bf16 rounding of the stored statistics' consumer: mean / rstd in fp32)
    g = rnd(*y.shape, seed=3).bfloat16().float()
    # backward reference on the bf16-rounded forward output (its sign selects the lrelu branch in the kernel)
    yb = out.float().cpu()
    xr = x.detach().clone().requires_grad_()
    br = beta.detach().clone().requires_grad_()
    yr, _, _ = O.batchnorm(xr, br)
    (yr * torch.where(yb > 0, 1.0, 0.2) * g).sum().backward()
    dx = torch.empty_like(xd)
    dbeta = torch.zeros(C, device=DEV)
    ws = torch.empty(2, C, device=DEV)
    K.bn_lrelu_backward(xd, out, g.to(DEV, torch.bfloat16), dx, stats, 1e-3, 0.2, dbeta, ws)
    close(dx, xr.grad, 2e-2, "bn dx bf16")
    close(dbeta, br.grad, 1e-3, "bn dbeta bf16")


def test_adam_and_gate():
    p, g = rnd(1000, seed=1), rnd(1000, seed=2)
    m, v = torch.zeros(1000), torch.zeros(1000)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    import math
    for t in (1, 2, 3):
        O.adam_tf_step(pr, g, mr, vr, t, 5e-5)
        lr_t = 5e-5 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        hyper = torch.tensor([lr_t, 0.9, 0.999, 1e-8, 1.0], device=DEV)
        K.adam_tf(pd, g.to(DEV), md, vd, hyper)
    close(pd, pr, 1e-6, "adam p")
    close(vd, vr, 1e-6, "adam v")
    before = pd.clone()
    K.adam_tf(pd, g.to(DEV), md, vd, torch.tensor([1.0, 0.9, 0.999, 1e-8, 0.0], device=DEV))
    assert torch.equal(pd, before)


def test_act_backward_and_reductions():
    y, g = rnd(3, 5, 7, 2, seed=1) * 20, rnd(3, 5, 7, 2, seed=2)
    d = torch.empty(3, 5, 7, 2, device=DEV)
    K.act_backward(g.to(DEV), y.to(DEV), d, ACT_TANH, 24.0, 1.0)
    close(d, g * (24.0 - y * y / 24.0), 1e-5, "tanh bwd")
    a, b = rnd(4, 33, 17, 3, seed=3), rnd(4, 33, 17, 3, seed=4)
    out = torch.zeros(2, device=DEV)
    K.sum_sq_diff(a.to(DEV), b.to(DEV), 0.5, out[0:1])
    K.sum_abs_diff(a.to(DEV), b.to(DEV), 2.0, out[1:2])
    close(out[0], ((a - b) ** 2).sum() * 0.5, 1e-5, "sum sq")
    close(out[1], (a - b).abs().sum() * 2.0, 1e-5, "sum abs")


@pytest.mark.parametrize("C", [64, 128, 256, 512, 48])
def test_cosine_loss_bf16_vectorised_and_scalar(C):
    """VGG cosine loss (reference lib/Teco.py:15-23): C in {64..512} takes the 16-byte kernel, 48 the scalar one."""
    npix = 37
    g = (rnd(npix, C, seed=1) + 0.3).bfloat16().float().requires_grad_()
    t = (rnd(npix, C, seed=2) + 0.3).bfloat16().float()
    gh = g / torch.sqrt((g * g).sum(-1, keepdim=True) + 1e-12)
    th = t / torch.sqrt((t * t).sum(-1, keepdim=True) + 1e-12)
    cos = (gh * th).sum(-1).sum()
    (cos * 0.7).backward()
    out = torch.zeros(1, device=DEV)
    dg = torch.empty(npix, C, device=DEV, dtype=torch.bfloat16)
    K.cosine_loss(g.detach().to(DEV, torch.bfloat16), t.to(DEV, torch.bfloat16), 0.25, 0.7, out, dg)
    close(out[0], cos.detach() * 0.25, 1e-4, "cosine sum C=%d" % C)
    tight(dg, g.grad, "cosine grad C=%d" % C, abs_=2e-4 * float(g.grad.abs().max()))


def test_act_backward_bf16_vectorised():
    y, g = rnd(2, 9, 5, 64, seed=1).bfloat16().float(), rnd(2, 9, 5, 64, seed=2).bfloat16().float()
    d = torch.empty(2, 9, 5, 64, device=DEV, dtype=torch.bfloat16)
    K.act_backward(g.to(DEV, torch.bfloat16), y.to(DEV, torch.bfloat16), d, ACT_LRELU, 0.2, 0.5)
    tight(d, 0.5 * g * torch.where(y > 0, 1.0, 0.2), "lrelu bwd bf16 x8")
    K.act_backward(g.to(DEV, torch.bfloat16), None, d, ACT_NONE, 0.0, 2.0)
    tight(d, 2.0 * g, "scale-only bf16 x8")


def test_pack_weights():
    flat = rnd(9 * 8 * 16 + 16 * 4, seed=1).to(DEV)
    tab = torch.tensor([[0, 0, 9, 8, 16, 8, 16], [9 * 8 * 16, 9 * 8 * 16, 1, 16, 4, 16, 4]], dtype=torch.int64,
                       device=DEV)
    for dtype in (torch.float32, torch.bfloat16):
        dst = torch.empty(flat.numel(), device=DEV, dtype=dtype)
        K.pack_weights(flat, dst, tab, 2, True)
        ref0 = flat[:9 * 8 * 16].reshape(9, 8, 16).permute(0, 2, 1).reshape(-1).to(dtype)
        ref1 = flat[9 * 8 * 16:].reshape(1, 16, 4).permute(0, 2, 1).reshape(-1).to(dtype)
        assert torch.equal(dst[:9 * 8 * 16], ref0) and torch.equal(dst[9 * 8 * 16:], ref1)
        K.pack_weights(flat, dst, tab, 2, False)
        assert torch.equal(dst, flat.to(dtype))
        # both layouts in one launch (the per-step refresh), with padded extents: 5 -> 8 reduction channels, 3 -> 8 outputs
        tabp = torch.tensor([[0, 0, 9, 8, 16, 8, 16], [9 * 8 * 16, 9 * 8 * 16, 4, 5, 3, 8, 8]], dtype=torch.int64, device=DEV)
        dT = torch.full((9 * 8 * 16 + 4 * 8 * 8,), 7.0, device=DEV, dtype=dtype)
        dN = torch.full_like(dT, 7.0)
        K.pack_weights_both(flat, dT, dN, tabp, 2)
        assert torch.equal(dT[:9 * 8 * 16], ref0) and torch.equal(dN[:9 * 8 * 16], flat[:9 * 8 * 16].to(dtype))
        w = flat[9 * 8 * 16:9 * 8 * 16 + 60].reshape(4, 5, 3)
        refT = torch.zeros(4, 3, 8, device=DEV)
        refT[:, :, :5] = w.permute(0, 2, 1)
        refN = torch.zeros(4, 8, 8, device=DEV)
        refN[:, :5, :3] = w
        assert torch.equal(dT[9 * 8 * 16:9 * 8 * 16 + 96], refT.reshape(-1).to(dtype))
        assert torch.equal(dN[9 * 8 * 16:], refN.reshape(-1).to(dtype))


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 64, 3, 1), (1, 9, 11, 32, 128, 4, 2), (3, 7, 5, 128, 64, 3, 1),
                                  (40, 32, 32, 64, 64, 3, 1), (2, 16, 16, 256, 8, 1, 1),
                                  # incremental addressing: several images per 64-pixel step, ragged carries, rows wider
                                  # than a step, strided output with ragged extents
                                  (21, 4, 4, 64, 64, 3, 1), (3, 6, 20, 64, 32, 3, 1), (2, 3, 70, 32, 64, 3, 1),
                                  (3, 5, 12, 64, 64, 4, 2), (5, 2, 2, 64, 64, 3, 1), (2, 10, 6, 8, 64, 3, 1)])
def test_wgrad_bf16_mfma_path(case):
    """bf16-MFMA weight gradient (transposed LDS staging) vs autograd on the same bf16-rounded operands."""
    N, H, W, Cin, Cout, k, s = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    Ho, pt = K.same_pad(H, k, s)
    Wo, pl = K.same_pad(W, k, s)
    gy = rnd(N, Ho, Wo, Cout, seed=5).bfloat16()
    xr = x.float().requires_grad_()
    w = torch.zeros(k, k, Cin, Cout, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    O.conv2(xr, w, b, s).backward(gy.float())
    d = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, TG_BF16, TG_BF16)
    dw = torch.zeros(k, k, Cin, Cout, device=DEV)
    db = torch.zeros(Cout, device=DEV)
    K.conv_wgrad(d, x.to(DEV), gy.to(DEV), dw, db)
    close(dw, w.grad, 3e-4, "bf16 wgrad %s" % (case,))
    close(db, b.grad, 3e-4, "bf16 bgrad %s" % (case,))


@pytest.mark.parametrize("case", [(5, 6, 8, 8, 64, 64, 3, 1), (1, 2, 16, 16, 64, 64, 3, 1), (3, 2, 16, 16, 64, 64, 4, 2)])
def test_wgrad_grouped_matches_per_layer(case):
    """tg_conv_wgrad_grouped == G independent tg_conv_wgrad calls (one launch for bf16 3x3 s1 layers, per-layer
    launches for any other geometry): the res-block layers of generator_F share one geometry (lib/frvsr.py:50-57)."""
    G, N, H, W, Cin, Cout, k, s = case
    Ho, pt = K.same_pad(H, k, s)
    Wo, pl = K.same_pad(W, k, s)
    d = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, TG_BF16, TG_BF16)
    xs = [rnd(N, H, W, Cin, seed=10 + g).to(DEV, torch.bfloat16) for g in range(G)]
    ys = [rnd(N, Ho, Wo, Cout, seed=50 + g).to(DEV, torch.bfloat16) for g in range(G)]
    ref_w = [torch.zeros(k, k, Cin, Cout, device=DEV) for _ in range(G)]
    ref_b = [torch.zeros(Cout, device=DEV) for _ in range(G)]
    for g in range(G):
        K.conv_wgrad(d, xs[g], ys[g], ref_w[g], ref_b[g])
    got_w = [torch.full((k, k, Cin, Cout), 0.5, device=DEV) for _ in range(G)]      # accumulates: start from 0.5
    got_b = [torch.zeros(Cout, device=DEV) for _ in range(G)]
    K.conv_wgrad_grouped(d, xs, ys, got_w, got_b)
    for g in range(G):
        close(got_w[g] - 0.5, ref_w[g], 2e-4, "grouped dW layer %d %s" % (g, case))
        close(got_b[g], ref_b[g], 2e-4, "grouped dbias layer %d %s" % (g, case))
    assert not torch.equal(got_w[0], got_w[-1]) or G == 1


def test_wgrad_bf16_padded_channel_stride():
    """First-layer case: 51 logical input channels stored with stride 56 (zero pad) -> dW is [3,3,51,64]."""
    N, H, W = 2, 8, 8
    x51 = rnd(N, H, W, 51, seed=1).bfloat16()
    x56 = torch.cat((x51, torch.zeros(N, H, W, 5, dtype=torch.bfloat16)), -1).contiguous()
    gy = rnd(N, H, W, 64, seed=5).bfloat16()
    xr = x51.float().requires_grad_()
    w = torch.zeros(3, 3, 51, 64, requires_grad=True)
    O.conv2(xr, w, None, 1).backward(gy.float())
    d = K.conv_desc(N, H, W, 51, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    dw = torch.zeros(3, 3, 51, 64, device=DEV)
    K.conv_wgrad(d, x56.to(DEV), gy.to(DEV), dw, None, ldx=56, ldy=64)
    close(dw, w.grad, 3e-4, "padded-stride bf16 wgrad")


# ---------------------------------------------------------------------------------------------------------
# round 2: kernel-level parity of the loss / packing / schedule entry points (previously only covered by the
# end-to-end step), per-element tolerance (tests/util.py)
# ---------------------------------------------------------------------------------------------------------
from util import assert_close_per_elem  # noqa: E402


def _triplet_setup(B=2, h=8, T=7, seed=3, zero_flow=False):
    """Frame-major sequences + LR flows as the engine holds them (lib/Teco.py:180-214)."""
    H = 4 * h
    frames = rnd(T, B, H, H, 3, seed=seed)
    lr = (rnd(T, B, h, h, 3, seed=seed + 1) + 1) * 0.5
    flow = rnd(T - 1, B, h, h, 2, seed=seed + 2, scale=0.0 if zero_flow else 1.5)
    t_size = 3 * (T // 3)
    nt = t_size // 3
    idx_pre = list(range(0, t_size, 3))
    idx_nxt = list(range(T - 1))[-2:-1 - t_size:-3]
    return frames, lr, flow, t_size, nt, idx_pre, idx_nxt


def _oracle_d_input(frames, lr, flow, t_size, nt, idx_pre, idx_nxt, B, h, off, merge):
    """oracle/teco.py d_input on batch-major tensors -> [tb,Ho,Ho,9|27] with tb index = b*nt + k."""
    H = 4 * h
    fr = frames[:t_size].transpose(0, 1)                                  # [B,t_size,H,H,3]
    gen_flow = O.upscale_four(flow.reshape(-1, h, h, 2) * 4.0).reshape(flow.shape[0], B, H, H, 2).transpose(0, 1)
    v_pre, v_nxt = gen_flow[:, idx_pre], gen_flow[:, idx_nxt]
    T_vel = torch.stack((v_pre, torch.zeros_like(v_pre), v_nxt), 2).reshape(B * t_size, H, H, 2)
    tb = B * nt
    flat = fr.reshape(B * t_size, H, H, 3)
    warped = O.crop_pad_dt(O.pack_triplets(O.dense_image_warp(flat, T_vel), tb), off)
    if not merge:
        return warped[:, off:H - off, off:H - off] if off else warped
    t_input = O.pack_triplets(lr[:t_size].transpose(0, 1).reshape(B * t_size, h, h, 3), tb)
    hi = O.resize_bilinear_legacy(t_input, H, H)
    return torch.cat((O.pack_triplets(flat, tb), warped, hi), -1)


def _to_engine_tb(x, B, nt):
    """oracle tb index b*nt+k -> engine tb index k*B+b."""
    return x.reshape(B, nt, *x.shape[1:]).transpose(0, 1).reshape(B * nt, *x.shape[1:])


@pytest.mark.parametrize("merge", [True, False])
def test_pack_d_input_zero_flow_is_bit_exact(merge):
    """Zero flow: the warp is the identity, so the before/warped blocks are pure index shuffles (channel = c*3+t,
    lib/Teco.py:227-229) and the crop/pad border is exact zeros -> bit-exact against oracle.ops.pack_triplets/crop_pad_dt."""
    B, h, off = 2, 8, 4
    frames, lr, flow, t_size, nt, ip, inx = _triplet_setup(B, h, zero_flow=True)
    ref = _to_engine_tb(_oracle_d_input(frames, lr, flow, t_size, nt, ip, inx, B, h, off, merge), B, nt)
    Ho = 4 * h if merge else 4 * h - 2 * off
    Cpad = 32 if merge else 16
    out = torch.full((B * nt, Ho, Ho, Cpad), 7.0, device=DEV)
    K.pack_d_input_forward(frames.to(DEV), lr.to(DEV), flow.to(DEV), flow.to(DEV), ip, inx, out, B, h, h, off, merge)
    C = 27 if merge else 9
    got = out.cpu()
    nshuf = 18 if merge else 9                                            # before | warped blocks: exact
    assert torch.equal(got[..., :nshuf], ref[..., :nshuf]), "index shuffle / crop-pad not bit-exact"
    assert torch.equal(got[..., C:], torch.zeros_like(got[..., C:])), "channel padding not zero"
    if merge:
        assert_close_per_elem(got[..., 18:27], ref[..., 18:27], 1e-5, what="bilinear LR context")


@pytest.mark.parametrize("merge", [True, False])
def test_pack_d_input_forward_backward_vs_oracle(merge):
    B, h, off = 2, 8, 4
    frames, lr, flow, t_size, nt, ip, inx = _triplet_setup(B, h, seed=11)
    fr = frames.clone().requires_grad_()
    ref = _to_engine_tb(_oracle_d_input(fr, lr, flow, t_size, nt, ip, inx, B, h, off, merge), B, nt)
    C = 27 if merge else 9
    Ho = ref.shape[1]
    Cpad = 32 if merge else 16
    out = torch.empty(B * nt, Ho, Ho, Cpad, device=DEV)
    K.pack_d_input_forward(frames.to(DEV), lr.to(DEV), flow.to(DEV), flow.to(DEV), ip, inx, out, B, h, h, off, merge)
    # the query position goes through upscale_four(4*flow) in a different association order than the oracle's 16 phase
    # blends (~1e-6 px), so near-zero pixels are compared against 1 % of the maximum, not 0.1 %
    assert_close_per_elem(out[..., :C], ref, 1e-3, floor=1e-2, what="pack_d fwd")
    g = rnd(B * nt, Ho, Ho, Cpad, seed=5)
    g[..., C:] = 0
    (ref * g[..., :C]).sum().backward()
    d_frames = torch.zeros_like(frames, device=DEV)
    K.pack_d_input_backward(g.to(DEV), frames.to(DEV), flow.to(DEV), flow.to(DEV), ip, inx, d_frames, B, h, h, off, merge)
    assert_close_per_elem(d_frames, fr.grad, 1e-3, floor=1e-2, what="pack_d bwd")


def test_pingpong_loss_and_gradient():
    """lib/Teco.py:362-370: mean |gen[k] - gen[T-1-k]| over k < RNN_N-1; gradient accumulates into d_gen."""
    T0, B, H = 4, 2, 8
    T = 2 * T0 - 1
    gen = rnd(T, B, H, H, 3, seed=21).requires_grad_()
    first, last_rev = gen[0:T0 - 1], gen[list(range(T))[-1:-T0:-1]]
    pp = (first - last_rev).abs().mean()
    (0.5 * pp).backward()
    seed_grad = rnd(T, B, H, H, 3, seed=22)
    d_gen = seed_grad.clone().to(DEV)
    loss = torch.zeros(1, device=DEV)
    cnt = float((T0 - 1) * gen[0].numel())
    K.pingpong(gen.detach().to(DEV), d_gen, T, T0 - 1, 1.0 / cnt, 0.5 / cnt, loss)
    assert_close_per_elem(loss, pp.detach().reshape(1), 1e-5, what="pingpong loss")
    assert_close_per_elem(d_gen.cpu() - seed_grad, gen.grad, 1e-4, what="pingpong grad")


def test_gan_losses_values_and_gradients():
    """lib/Teco.py:374-399 on sigmoid outputs incl. values near 0 and 1 (EPS = 1e-12)."""
    n, eps, w = 300, 1e-12, 0.01
    g = torch.Generator().manual_seed(31)
    real = torch.rand(n, generator=g).clamp(1e-4, 1 - 1e-4).requires_grad_()
    fake = torch.rand(n, generator=g).clamp(1e-4, 1 - 1e-4).requires_grad_()
    t_adv = (-torch.log(fake + eps)).mean()
    t_dis = (-(torch.log(1 - fake + eps) + torch.log(real + eps))).mean()
    t_bal = torch.log(real + eps).mean() + t_adv
    out = torch.zeros(5, device=DEV)
    d_r, d_f, d_g = (torch.empty(n, device=DEV) for _ in range(3))
    K.gan_losses(real.detach().to(DEV), fake.detach().to(DEV), eps, w, out, d_r, d_f, d_g)
    ref = torch.stack((t_adv, t_dis, t_bal, real.mean(), fake.mean())).detach()
    assert_close_per_elem(out, ref, 1e-5, what="gan loss scalars")
    gr, gf = torch.autograd.grad(t_dis, (real, fake), retain_graph=True)
    (gg,) = torch.autograd.grad(w * t_adv, (fake,))
    assert_close_per_elem(d_r, gr, 1e-5, what="d t_discrim / d real")
    assert_close_per_elem(d_f, gf, 1e-5, what="d t_discrim / d fake")
    assert_close_per_elem(d_g, gg, 1e-5, what="d adv / d fake")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_l1_layer_loss(dtype):
    """lib/Teco.py:291-302: mean over pixels of sum_c |r - f|; d_f = -grad_scale * sign(r - f)."""
    r, f = rnd(3, 6, 6, 64, seed=41).to(dtype).float(), rnd(3, 6, 6, 64, seed=42).to(dtype).float()
    npix = 3 * 6 * 6
    loss = torch.zeros(1, device=DEV)
    d = torch.empty(3, 6, 6, 64, device=DEV, dtype=dtype)
    K.l1_loss(r.to(DEV, dtype), f.to(DEV, dtype), 1.0 / npix, 0.25 / npix, loss, d)
    ref = (r - f).abs().sum(3).mean()
    assert_close_per_elem(loss, ref.reshape(1), 1e-5, what="l1 loss")
    gref = -(0.25 / npix) * torch.sign(r - f)
    assert_close_per_elem(d.float(), gref.to(dtype).float(), 1e-6, what="l1 grad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vgg_preprocess_forward_backward(dtype):
    """lib/Teco.py:9-10: deprocess * 255 - VGG_MEAN (RGB 123.68, 116.78, 103.94), zero channel padding; d_x += 127.5 d_out."""
    x = rnd(2, 9, 7, 3, seed=51)
    out = torch.full((2, 9, 7, 8), 3.0, device=DEV, dtype=dtype)
    K.vgg_preprocess_forward(x.to(DEV), out)
    ref = O.vgg_preprocess(x)
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert_close_per_elem(out[..., :3].float(), ref, tol, what="vgg preprocess")
    assert torch.equal(out[..., 3:].float().cpu(), torch.zeros(2, 9, 7, 5))
    g = rnd(2, 9, 7, 8, seed=52).to(dtype)
    acc0 = rnd(2, 9, 7, 3, seed=53)
    d_x = acc0.clone().to(DEV)
    K.vgg_preprocess_backward(g.to(DEV), d_x)
    assert_close_per_elem(d_x.cpu() - acc0, 127.5 * g[..., :3].float(), 1e-4, floor=1e-2, what="vgg preprocess bwd")


def test_schedule_step_lr_decay_ema_and_gate():
    """lib/Teco.py:95-99 (exponential_decay, staircase), :415-417 (EMA 0.99 from 0, no debias), :493-494 (gate on the OLD
    average), per-optimiser Adam bias correction; D learning rate x0.3 when Dt_mergeDs is off (:423-424)."""
    import math
    st = torch.zeros(8 + 2 * 3, dtype=torch.float64)
    st[0], st[2], st[3], st[4], st[5], st[6], st[7] = 9.0, 1e-3, 4.0, 0.5, 1.0, 0.4, 0.3
    state = st.to(DEV)
    hyper = torch.zeros(3, 8, device=DEV)
    tb, t_counts = 0.0, [0, 0, 0]
    for step, bal in enumerate((50.0, 10.0, -3.0)):
        K.schedule_step(state, hyper, 3, 0, torch.tensor([bal], device=DEV), 0.9, 0.999, 1e-8)
        gstep = 9 + step
        lr = O.exponential_decay(1e-3, gstep, 4, 0.5, True)
        gate = tb < 0.4
        tb = O.ema_tf(tb, bal)
        hs = hyper.cpu()
        for k in range(3):
            on = gate if k == 0 else True
            t_counts[k] += 1 if on else 0
            lrk = lr * 0.3 if k == 0 else lr
            t = t_counts[k]
            lr_t = lrk * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) if t > 0 else float("nan")
            assert hs[k, 4].item() == (1.0 if on else 0.0), (step, k)
            if on:
                # 2e-5: the kernel takes beta1/beta2 as float32 (0.999f), as TF's float32 beta-power variables do
                assert abs(hs[k, 0].item() - lr_t) <= 2e-5 * lr_t, (step, k, hs[k, 0].item(), lr_t)
            assert abs(hs[k, 5].item() - lrk) <= 1e-6 * lrk
        s = state.cpu()
        assert s[0].item() == gstep + 1 and abs(s[1].item() - tb) < 1e-9
    assert t_counts[0] < 3, "the gate never closed in this scenario"


def test_concat2_pad_lincomb_affine_seq_gather():
    a, b = rnd(2, 5, 7, 3, seed=61), rnd(2, 5, 7, 3, seed=62)
    out = torch.full((2, 5, 7, 8), 9.0, device=DEV)
    K.concat2_pad(a.to(DEV), b.to(DEV), out)
    ref = torch.cat((a, b, torch.zeros(2, 5, 7, 2)), -1)
    assert torch.equal(out.cpu(), ref), "concat2_pad is a pure copy: bit-exact"
    o2 = torch.empty(2, 5, 7, 3, device=DEV)
    K.lincomb(a.to(DEV), b.to(DEV), o2, 0.25, -1.5)
    assert_close_per_elem(o2, 0.25 * a - 1.5 * b, 1e-5, what="lincomb")
    K.lincomb(a.to(DEV), None, o2, 2.0, 0.0, accumulate=True)
    assert_close_per_elem(o2, 2.25 * a - 1.5 * b, 1e-5, what="lincomb accumulate")
    K.affine(a.to(DEV), o2, 0.5, 0.5)
    assert torch.equal(o2.cpu(), a * 0.5 + 0.5), "deprocess (x+1)/2 == x*0.5+0.5 bit for bit"
    # ping-pong order + [B,T] -> [T,B] (lib/Teco.py:80-85): bit-exact gather
    src = rnd(3, 4, 6, 6, 3, seed=63)
    idx = [0, 1, 2, 3, 2, 1, 0]
    dst = torch.empty(7, 3, 6, 6, 3, device=DEV)
    K.seq_gather(src.to(DEV), dst, idx)
    assert torch.equal(dst.cpu(), src.transpose(0, 1)[idx].contiguous())
    # odd crop size: a frame of 5*5*3 = 75 floats is not a whole number of 16-byte vectors (scalar form)
    src = rnd(2, 3, 5, 5, 3, seed=64)
    idx = [0, 1, 2, 1, 0]
    dst = torch.empty(5, 2, 5, 5, 3, device=DEV)
    K.seq_gather(src.to(DEV), dst, idx)
    assert torch.equal(dst.cpu(), src.transpose(0, 1)[idx].contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_backward_with_tap_gradient(dtype):
    """VGG taps sit right before a pool (lib/ops.py:319-327, lib/Teco.py:176): d_in = (route(d_out) + d_tap) * relu'(x)."""
    x = rnd(2, 9, 8, 16, seed=71).to(dtype).float()
    xr = torch.relu(x).requires_grad_()                    # the pooled tensor is a ReLU output
    pooled = O.maxpool(xr)
    g = rnd(*pooled.shape, seed=72).to(dtype).float()
    d_tap = rnd(*x.shape, seed=73).to(dtype).float()
    (gx,) = torch.autograd.grad((pooled * g).sum() + (xr * d_tap).sum(), (xr,))
    ref = gx * (xr.detach() > 0).float()
    out = torch.empty(*x.shape, device=DEV, dtype=dtype)
    K.maxpool2_backward(xr.detach().to(DEV, dtype), g.to(DEV, dtype), out, ACT_RELU, 0.0, add=d_tap.to(DEV, dtype))
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert_close_per_elem(out.float(), ref, tol, floor=1e-2, what="maxpool bwd + tap")


def test_launch_profiler_counts_and_flops():
    """tg_prof_*: every instrumented launch is booked with its algorithmic FLOPs; disabled -> nothing recorded."""
    x = rnd(1, 16, 16, 64, seed=81).to(DEV, torch.bfloat16)
    w = rnd(9, 64, 64, seed=82, scale=0.05).to(DEV, torch.bfloat16)
    out = torch.empty_like(x)
    d = K.conv_desc(1, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU)
    K.conv_forward(d, x, w, None, None, None, out)
    torch.cuda.synchronize()
    K.prof_collect()
    K.prof_enable(True)
    for _ in range(3):
        K.conv_forward(d, x, w, None, None, None, out)
    K.prof_enable(False)
    K.conv_forward(d, x, w, None, None, None, out)
    ents = K.prof_collect()
    assert len(ents) == 1 and ents[0]["calls"] == 3, ents
    assert ents[0]["name"].startswith("conv3x3_tile<bf16,bf16")
    assert ents[0]["flops"] == 3 * 2.0 * 16 * 16 * 64 * 9 * 64
    assert 0.5 < ents[0]["total_us"] / 3 < 200.0, ents
    assert K.prof_collect() == []


# ---- conv3x3_ws.hip: weights-in-registers / LDS-DMA kernel for one-chunk layers (>= 256 tiles of 8x16 pixels) -----
WS_CASES = [
    # N, H, W, Cin, Cout, flip(bwd_data form), res, aux(mask), act
    (1, 270, 250, 64, 64, False, False, False, ACT_RELU),     # ragged right/bottom edges, persistent tile loop
    (2, 128, 128, 64, 64, False, True, False, ACT_NONE),      # residual epilogue (res-block conv_2)
    (1, 270, 480, 56, 64, False, False, False, ACT_RELU),     # Cin = 56 (generator input conv): zero-filled chunk
    (3, 128, 128, 64, 128, False, False, False, ACT_LRELU),   # two channel tiles (grid.y = 2)
    (2, 128, 128, 64, 64, True, True, True, ACT_NONE),        # input-gradient form: mirrored taps + residual + ReLU mask
    (1, 200, 170, 32, 64, True, False, True, ACT_NONE),       # Cin = 32, LeakyReLU mask
    (1, 1080, 1920, 16, 64, False, False, False, ACT_RELU),   # > 2^16 tiles per image row block; Cin = 16
]


@pytest.mark.parametrize("case", WS_CASES)
@pytest.mark.parametrize("coexist", [0, 1])
def test_conv3x3_weights_in_registers_kernel(case, coexist):
    N, H, W, Cin, Cout, flip, has_res, has_aux, act = case
    if coexist and H * W > 300 * 300:
        pytest.skip("one size is enough for the residency-capped launch")
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.1).bfloat16()               # HWIO, as conv2 sees it
    b = None if flip else rnd(Cout, seed=3)
    res = rnd(N, H, W, Cout, seed=4).bfloat16() if has_res else None
    aux = rnd(N, H, W, Cout, seed=5).bfloat16() if has_aux else None
    alpha = 0.2 if act == ACT_LRELU else 0.0
    mask_act = ACT_LRELU if (has_aux and Cin == 32) else (ACT_RELU if has_aux else ACT_NONE)
    mask_alpha = 0.2 if mask_act == ACT_LRELU else 0.0
    # reference: mirrored taps == conv with the spatially flipped kernel
    wr = w.float().flip(0, 1) if flip else w.float()
    ref = O.conv2(x.float(), wr, b, 1)
    if act == ACT_RELU:
        ref = torch.relu(ref)
    elif act == ACT_LRELU:
        ref = torch.where(ref > 0, ref, ref * alpha)
    if has_res:
        ref = ref + res.float()
    if has_aux:
        ref = ref * torch.where(aux.float() > 0, torch.ones(()), torch.full((), 1.0 if mask_act == ACT_NONE else mask_alpha))
    wt = w.permute(0, 1, 3, 2).reshape(9, Cout, Cin).contiguous().to(DEV)     # [tap][Cout][Cin]
    out = torch.full((N, H, W, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1 if flip else 0, TG_BF16, TG_BF16, act, alpha, mask_act,
                    mask_alpha, flags=coexist)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_forward(d, x.to(DEV), wt, None if b is None else b.to(DEV), None if res is None else res.to(DEV),
                   None if aux is None else aux.to(DEV), out)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"].startswith("conv3x3_ws"), "the weights-in-registers kernel was not selected: %s" % ents
    # bf16 output rounding: 2^-8 relative per element, against values that reach a few units
    close(out, ref, 8e-3, "conv3x3_ws %s" % (case,))
    err = (out.float().cpu() - ref).abs()
    assert (err <= 8e-3 * ref.abs() + 2e-2).all(), "per-element: max %g" % err.max().item()


# ---- vectorised HBM-bound kernels of the inference step ---------------------------------------------------------------

@pytest.mark.parametrize("shape", [(1, 270, 480), (2, 128, 136), (1, 250, 130)])
@pytest.mark.parametrize("has_res,act", [(False, ACT_RELU), (True, ACT_NONE)])
def test_conv3x3_fragment_order_weights_entry_is_bit_identical(shape, has_res, act):
    """tg_conv3x3_c64_frag (the inference step's res-block convs: weight operand in fragment order, two workgroups per CU)
    against tg_conv_forward on the same kernel with the row-major operand: same MFMA order -> bit-identical."""
    N, H, W = shape
    x = rnd(N, H, W, 64, seed=1).bfloat16().to(DEV)
    wt = rnd(9, 64, 64, seed=2, scale=0.1).bfloat16().to(DEV)            # [tap][Cout][Cin]
    b = rnd(64, seed=3).to(DEV)
    res = rnd(N, H, W, 64, seed=4).bfloat16().to(DEV) if has_res else None
    ref = torch.full((N, H, W, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    out = torch.full_like(ref, 5.0)
    d = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, act, 0.0)
    K.conv_forward(d, x, wt, b, res, None, ref)
    K.conv3x3_c64_frag(x, K.frag_order(wt), b, res, out, act)
    assert torch.equal(out, ref)
    with pytest.raises(Exception):
        K.conv3x3_c64_frag(x[:, :32, :32].contiguous(), K.frag_order(wt), b, None, out[:, :32, :32].contiguous(), act)

@pytest.mark.parametrize("shape", [(1, 270, 480), (2, 128, 136), (1, 250, 131), (3, 6, 14), (1, 5, 3)])
def test_resblock_throughput_kernel_matches_oracle_and_the_two_launch_path(shape):
    """tg_resblock_c64_thr (round 5: one residual block of generator_F, lib/frvsr.py:50-57, as one launch at inference resolution, the
    intermediate in LDS) against the oracle's conv2 chain on the bf16-rounded operands with the intermediate rounded to bf16 (what
    both paths store), and against two tg_conv_forward launches: same products, fp32 accumulation, one rounding per conv."""
    N, H, W = shape
    x = rnd(N, H, W, 64, seed=1).bfloat16()
    w1 = rnd(3, 3, 64, 64, seed=2, scale=0.08).bfloat16()
    w2 = rnd(3, 3, 64, 64, seed=3, scale=0.08).bfloat16()
    b1, b2 = rnd(64, seed=4), rnd(64, seed=5)
    mid = torch.relu(O.conv2(x.float(), w1.float(), b1, 1)).bfloat16().float()
    ref = x.float() + O.conv2(mid, w2.float(), b2, 1)
    wt1 = w1.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV)
    wt2 = w2.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV)
    out = torch.full((N, H, W, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    K.prof_collect()
    K.prof_enable(True)
    K.resblock_c64_thr(x.to(DEV), K.frag_order(wt1), b1.to(DEV), K.frag_order(wt2), b2.to(DEV), out)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"] == "resblock_thr", ents
    # the intermediate is rounded to bf16 in both; a different summation order moves a few intermediate values by one bf16 step
    # (2^-8 relative), which the second conv spreads over its 576-term sums: well below the result's own rounding step
    err = (out.float().cpu() - ref).abs()
    assert (err <= 4e-3 * ref.abs() + 6e-3).all(), "%s: max err %g at |ref| %g" % (shape, err.max().item(), ref.abs().flatten()[err.argmax()].item())
    d = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU, 0.0)
    r = K.conv_forward(d, x.to(DEV), wt1, b1.to(DEV), None, None, torch.empty_like(out))
    d2 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_NONE, 0.0)
    two = K.conv_forward(d2, r, wt2, b2.to(DEV), x.to(DEV), None, torch.empty_like(out))
    close(out, two.float().cpu(), 8e-3, "one-launch block vs two launches %s" % (shape,))
    with pytest.raises(Exception):
        K.resblock_c64_thr(out, K.frag_order(wt1), None, K.frag_order(wt2), None, out)        # in place is refused


@pytest.mark.parametrize("B,h,w", [(1, 5, 9), (2, 33, 47), (1, 270, 480)])
def test_warp_s2d_forward_bf16_vectorised_rows(B, h, w):
    """The LDS-assembled 16-byte-row form (bf16, Cpad 56) incl. pixel counts that are not multiples of 4 / of the grid and
    the optional warped-frame output; against the oracle, and bit-exact against the scalar fallback (Cpad 59 = odd row size)."""
    pre = rnd(B, 4 * h, 4 * w, 3, seed=1)
    flow = rnd(B, h, w, 2, seed=2, scale=3.0)
    lr = (rnd(B, h, w, 3, seed=3) + 1) * 0.5
    ref = oracle_gen_input(pre, flow, lr, 0.5, 0.5, 56)
    out = torch.full((B, h, w, 56), 9.0, device=DEV, dtype=torch.bfloat16)
    warped = torch.empty(B, 4 * h, 4 * w, 3, device=DEV)
    K.warp_s2d_forward(pre.to(DEV), flow.to(DEV), lr.to(DEV), out, 0.5, 0.5, warped=warped)
    close(out, ref, 4e-3, "warp_s2d bf16 rows")
    # query coordinates reach 4w: fp32 carries them to ~4w * 6e-8 px, and random frames change by O(1) per pixel
    close(warped, O.dense_image_warp(pre, O.upscale_four(flow * 4.0)), max(2e-5, 4 * w * 3e-7), "warped frame")
    out59 = torch.empty(B, h, w, 59, device=DEV, dtype=torch.bfloat16)                   # 118-byte rows: scalar kernel
    K.warp_s2d_forward(pre.to(DEV), flow.to(DEV), lr.to(DEV), out59, 0.5, 0.5)
    assert torch.equal(out59[..., :51].cpu(), out[..., :51].cpu()), "vector and scalar kernels disagree"
    assert torch.equal(out[..., 51:].float().cpu(), torch.zeros(B, h, w, 5))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bicubic_quad_kernel_state_output(dtype):
    """Row-quad bicubic epilogue: frame in [-1,1] and, in the same pass, the deprocessed recurrent state (main.py:207);
    `out` may be omitted.  Odd sizes, replicate padding at all four borders."""
    B, h, w = 2, 7, 9
    gen_in = rnd(B, h, w, 56, seed=1).to(dtype)
    conv_out = rnd(B, 4 * h, 4 * w, 3, seed=2)
    ref = O.preprocess(conv_out + O.bicubic_four(gen_in[..., :3].float()))
    out = torch.empty(B, 4 * h, 4 * w, 3, device=DEV)
    state = torch.empty_like(out)
    K.bicubic_add_preprocess(conv_out.to(DEV), gen_in.to(DEV), out, state)
    close(out, ref, 2e-6, "bicubic quad out")
    assert torch.equal(state.cpu(), out.cpu() * 0.5 + 0.5), "state must be deprocess(out) bit for bit"
    state2 = torch.empty_like(out)
    K.bicubic_add_preprocess(conv_out.to(DEV), gen_in.to(DEV), None, state2)
    assert torch.equal(state2.cpu(), state.cpu())


# ---- narrow images packed side by side in one tile row (conv3x3.hip PACK = 2 / 4) -------------------------------------
PACK_CASES = [
    # N, H, W, Cin, Cout, flip, res, aux, act
    # (image counts: the packed tiles are only selected for chip-filling layers, pixels x channel tiles >= 16384)
    (130, 8, 8, 128, 128, False, False, False, ACT_RELU),     # VGG conv5-like geometry: two 8x8 images per tile row
    (257, 8, 8, 128, 64, False, True, False, ACT_NONE),       # odd image count (last group half empty), residual
    (257, 4, 4, 256, 256, False, False, False, ACT_LRELU),    # four 4x4 images per row, ragged last group
    (343, 8, 6, 128, 64, True, False, True, ACT_NONE),        # W = 6 < 8, input-gradient form with ReLU mask
    (66, 16, 8, 64, 128, False, False, False, ACT_RELU),      # H = 16: two tiles per image column
    (1823, 3, 3, 128, 64, True, True, True, ACT_NONE),        # 3x3 images, H <= 4, ragged last group
]


@pytest.mark.parametrize("case", PACK_CASES)
def test_conv3x3_packed_narrow_images(case):
    N, H, W, Cin, Cout, flip, has_res, has_aux, act = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.1).bfloat16()
    b = None if flip else rnd(Cout, seed=3)
    res = rnd(N, H, W, Cout, seed=4).bfloat16() if has_res else None
    aux = rnd(N, H, W, Cout, seed=5).bfloat16() if has_aux else None
    alpha = 0.2 if act == ACT_LRELU else 0.0
    ref = O.conv2(x.float(), w.float().flip(0, 1) if flip else w.float(), b, 1)
    if act == ACT_RELU:
        ref = torch.relu(ref)
    elif act == ACT_LRELU:
        ref = torch.where(ref > 0, ref, ref * alpha)
    if has_res:
        ref = ref + res.float()
    if has_aux:
        ref = ref * (aux.float() > 0).float()
    wt = w.permute(0, 1, 3, 2).reshape(9, Cout, Cin).contiguous().to(DEV)
    out = torch.full((N, H, W, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1 if flip else 0, TG_BF16, TG_BF16, act, alpha,
                    ACT_RELU if has_aux else ACT_NONE, 0.0)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_forward(d, x.to(DEV), wt, None if b is None else b.to(DEV), None if res is None else res.to(DEV),
                   None if aux is None else aux.to(DEV), out)
    K.prof_enable(False)
    ents = K.prof_collect()
    # (round 6: the tile kernel's packed instantiation exists with the LDS-staged epilogue only, and a residual operand takes the
    #  fp32 register epilogue -- one rounding -- so the 3x3-image case with a residual runs unpacked)
    assert ents and ("pack" in ents[0]["name"] or (has_res and ents[0]["name"].startswith("conv3x3_tile"))), \
        "the packed-tile instantiation was not selected: %s" % ents
    err = (out.float().cpu() - ref).abs()
    assert (err <= 8e-3 * ref.abs() + 3e-2).all(), "%s: max err %g" % (case, err.max().item())


# ---- conv3x3_dma.hip: double-buffered LDS-DMA stages for the wide layers (Cin a multiple of 32, > 64) ------------------
DMA_CASES = [
    # N, H, W, Cin, Cout, flip, res, aux, act
    (6, 32, 32, 256, 256, False, False, False, ACT_RELU),     # VGG conv3_x geometry: 4 tiles per image, 8 chunks
    (40, 16, 16, 512, 512, False, False, False, ACT_RELU),    # VGG conv4_x: one tile per image, 16 chunks, grid.y = 8
    (3, 64, 64, 128, 128, True, False, True, ACT_NONE),       # conv2_2 input gradient: mirrored taps + ReLU mask
    (35, 40, 27, 96, 64, False, True, False, ACT_LRELU),      # ragged right / bottom edges, Cin = 96 (3 chunks), residual
    (2, 128, 128, 128, 64, True, True, True, ACT_NONE),       # many tiles per workgroup (persistent loop), res + mask
    (97, 16, 16, 128, 128, False, False, False, ACT_NONE),    # FNet level-2 geometry; tile count not a multiple of the grid
    (76, 32, 32, 256, 256, False, False, False, ACT_RELU),    # VGG conv3_x at the full batch
    (57, 16, 16, 512, 512, False, False, False, ACT_RELU),    # odd tile count: the last unit's second tile does not exist
    (57, 40, 27, 96, 128, False, True, False, ACT_LRELU),     # ragged edges + residual, pairs straddle image boundaries
    (20, 64, 64, 128, 128, True, False, True, ACT_NONE),      # input-gradient form with ReLU mask
    # packed tiles (round 3): 4 whole 8 x 8 images / 16 whole 4 x 4 images per 16 x 16 tile, each with its own zero border
    (76, 8, 8, 512, 512, False, False, False, ACT_RELU),      # VGG conv5_x at the full batch
    (37, 8, 8, 128, 128, True, True, True, ACT_NONE),         # image count not a multiple of 4; input-gradient form, res + mask
    (70, 4, 4, 256, 256, False, True, False, ACT_LRELU),      # FNet's innermost level; count not a multiple of 16
    (65, 4, 4, 256, 256, True, False, True, ACT_NONE),
    # round 4: packed launches below 256 workgroups of 32 channels take 32-channel blocks (J = 1); these are the training step's
    (32, 8, 8, 512, 512, False, False, False, ACT_RELU),      # VGG conv5_x of the 8-frame chunk: 8 tiles x 16 blocks
    (44, 8, 8, 512, 512, True, False, True, ACT_NONE),        # ... input gradient with ReLU mask, 11 tiles
    (530, 4, 4, 256, 256, False, True, False, ACT_LRELU),     # 4 x 4 images in 64-channel blocks (34 tiles x 8 > 256)
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_conv3x3_wide_layer_dma_kernel(case):
    N, H, W, Cin, Cout, flip, has_res, has_aux, act = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.05).bfloat16()
    b = None if flip else rnd(Cout, seed=3)
    res = rnd(N, H, W, Cout, seed=4).bfloat16() if has_res else None
    aux = rnd(N, H, W, Cout, seed=5).bfloat16() if has_aux else None
    alpha = 0.2 if act == ACT_LRELU else 0.0
    ref = O.conv2(x.float(), w.float().flip(0, 1) if flip else w.float(), b, 1)
    if act == ACT_RELU:
        ref = torch.relu(ref)
    elif act == ACT_LRELU:
        ref = torch.where(ref > 0, ref, ref * alpha)
    if has_res:
        ref = ref + res.float()
    if has_aux:
        ref = ref * (aux.float() > 0).float()
    wt = w.permute(0, 1, 3, 2).reshape(9, Cout, Cin).contiguous().to(DEV)
    out = torch.full((N, H, W, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1 if flip else 0, TG_BF16, TG_BF16, act, alpha,
                    ACT_RELU if has_aux else ACT_NONE, 0.0)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_forward(d, x.to(DEV), wt, None if b is None else b.to(DEV), None if res is None else res.to(DEV),
                   None if aux is None else aux.to(DEV), out)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"].startswith("conv3x3_dma"), "the wide-layer DMA kernel was not selected: %s" % ents
    tight(out, ref, "%s" % (case,), abs_=1e-3)           # (sums of 1152 .. 4608 products of O(1) size: fp32 summation-order noise)


# ---- conv3x3_wr.hip (round 5): wide frozen layers, weight fragments streamed into registers ------------------------------------
WR_CASES = [
    # N, H, W, Cin, Cout, flip, res, aux, act, tile_rows[, ksplit]
    (6, 32, 32, 256, 256, False, False, False, ACT_RELU, 16),    # VGG conv3_x geometry: 4 tiles per image, 8 stages
    (6, 32, 32, 256, 256, False, False, False, ACT_RELU, 8),     # ... in 8-row tiles
    (28, 16, 16, 512, 512, False, False, False, ACT_RELU, 0),    # VGG conv4_x of the early chunk: tile height chosen by the launch
    (48, 16, 16, 512, 512, True, False, True, ACT_NONE, 0),      # ... input gradient of the late chunk with the ReLU mask
    (3, 64, 64, 128, 128, True, False, True, ACT_NONE, 16),      # conv2_2 input gradient: mirrored taps + ReLU mask
    (5, 64, 64, 128, 64, True, False, False, ACT_NONE, 0),       # conv2_1 input gradient (128 -> 64: one channel block)
    (35, 40, 27, 96, 64, False, True, False, ACT_LRELU, 16),     # ragged right / bottom edges, Cin = 96 (3 stages), residual
    (35, 40, 27, 96, 128, False, True, True, ACT_LRELU, 8),      # ... 8-row tiles, residual + mask; unit count not a multiple of 8
    (1, 128, 128, 128, 64, True, True, True, ACT_NONE, 0),       # one image, many tiles
    (76, 32, 32, 128, 256, False, False, False, ACT_RELU, 0),    # conv3_1 at the full batch
    # packed tiles: two whole 8 x 8 images per 8 x 16 tile, each with its own zero border (VGG conv5_x)
    (37, 8, 8, 128, 128, False, False, False, ACT_RELU, 0, 1),   # odd image count: the last tile's second image does not exist
    (48, 8, 8, 512, 512, False, False, False, ACT_RELU, 0, 0),   # conv5_x of the late chunk: K split chosen by the launch (2)
    (28, 8, 8, 512, 512, True, False, True, ACT_NONE, 0, 0),     # ... input gradient of the early chunk (4)
    (5, 8, 8, 256, 64, False, False, True, ACT_LRELU, 0, 2),     # forced K splits
    (9, 8, 8, 256, 128, True, False, False, ACT_NONE, 0, 4),
    (28, 16, 16, 512, 512, False, True, True, ACT_RELU, 8, 2),   # K split on plain 8-row tiles, residual + mask
    (3, 24, 40, 256, 64, False, False, False, ACT_NONE, 8, 4),
]


@pytest.mark.parametrize("case", WR_CASES)
def test_conv3x3_wide_frag_kernel_is_bit_identical_to_conv_forward(case):
    """tg_conv3x3_wide_frag (weights global -> registers from the fragment-order copy, halo by LDS-DMA) against the oracle's conv2
    (lib/ops.py:47-56) AND bit for bit against tg_conv_forward on the same operands: same MFMA, same operand roles, same
    accumulation order (stage, kw, kh)."""
    N, H, W, Cin, Cout, flip, has_res, has_aux, act, th = case[:10]
    ks = case[10] if len(case) > 10 else 1
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.05).bfloat16()
    b = None if flip else rnd(Cout, seed=3)
    res = rnd(N, H, W, Cout, seed=4).bfloat16() if has_res else None
    aux = rnd(N, H, W, Cout, seed=5).bfloat16() if has_aux else None
    alpha = 0.2 if act == ACT_LRELU else 0.0
    ref = O.conv2(x.float(), w.float().flip(0, 1) if flip else w.float(), b, 1)
    if act == ACT_RELU:
        ref = torch.relu(ref)
    elif act == ACT_LRELU:
        ref = torch.where(ref > 0, ref, ref * alpha)
    if has_res:
        ref = ref + res.float()
    if has_aux:
        ref = ref * (aux.float() > 0).float()
    wt = w.permute(0, 1, 3, 2).reshape(9, Cout, Cin).contiguous().to(DEV)          # [tap][Cout][Cin], the operand of tg_conv_forward
    wf = K.pack_wide_frag(wt, torch.empty_like(wt), Cout, Cin, flip)
    d = K.conv_desc(N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1 if flip else 0, TG_BF16, TG_BF16, act, alpha,
                    ACT_RELU if has_aux else ACT_NONE, 0.0)
    assert K.conv3x3_wide_frag_ok(d)
    args = (None if b is None else b.to(DEV), None if res is None else res.to(DEV), None if aux is None else aux.to(DEV))
    old = K.conv_forward(d, x.to(DEV), wt, *args, torch.full((N, H, W, Cout), 7.0, device=DEV, dtype=torch.bfloat16))
    K.prof_collect()
    K.prof_enable(True)
    out = K.conv3x3_wide_frag(d, x.to(DEV), wf, *args, torch.full((N, H, W, Cout), 7.0, device=DEV, dtype=torch.bfloat16),
                              tile_rows=th, ksplit=ks)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"].startswith("conv3x3_wr"), ents
    assert ("pack2" in ents[0]["name"]) == (H == 8 and W == 8), ents
    tight(out, ref, "%s" % (case,), abs_=1e-3)           # (sums of 1152 .. 4608 products of O(1) size: fp32 summation-order noise)
    if ks != 1:
        # K split: partial sums over input-channel ranges, added in a fixed order -- same products, another summation order than
        # tg_conv_forward's single sum: equal up to a bf16 rounding step of the result, and bit-reproducible from run to run
        d2 = (out.float() - old.float()).abs()
        assert (d2 <= 8e-3 * old.float().abs() + 1e-3).all(), "%s: differs from tg_conv_forward by %g" % (case, d2.max().item())
        again = K.conv3x3_wide_frag(d, x.to(DEV), wf, *args, torch.empty_like(out), tile_rows=th, ksplit=ks)
        assert torch.equal(out.view(torch.int16), again.view(torch.int16)), "K-split result not reproducible"
        return
    assert torch.equal(out.view(torch.int16), old.view(torch.int16)), \
        "%s: %d of %d elements differ from tg_conv_forward, max %g" % (case, (out != old).sum().item(), out.numel(),
                                                                       (out.float() - old.float()).abs().max().item())


def test_wide_frag_rejects_what_it_does_not_cover():
    from tecogan_amd._lib import TecoHipError
    x = torch.zeros(1, 16, 16, 48, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(9 * 64 * 48, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(1, 16, 16, 48, 16, 16, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    with pytest.raises(TecoHipError):
        K.conv3x3_wide_frag(d, x, w, None, None, None, torch.zeros(1, 16, 16, 64, device=DEV, dtype=torch.bfloat16))
    d = K.conv_desc(1, 16, 16, 128, 8, 8, 64, 4, 4, 2, 1, 1, 0, TG_BF16, TG_BF16)
    with pytest.raises(TecoHipError):
        K.conv3x3_wide_frag(d, torch.zeros(1, 16, 16, 128, device=DEV, dtype=torch.bfloat16), w, None, None, None,
                            torch.zeros(1, 8, 8, 64, device=DEV, dtype=torch.bfloat16))


# ---- conv4x4s2.hip (round 5): the discriminator's 4x4 stride-2 convs and their input gradients ---------------------------------
K4_CASES = [
    # N, H, W (input of the forward conv), Cin, Cout
    (24, 128, 128, 64, 64),     # disblock_1 at configs[2]: 768 tiles
    (3, 64, 64, 64, 64),        # disblock_3
    (5, 32, 32, 64, 128),       # disblock_5: two channel blocks
    (7, 16, 16, 128, 256),      # disblock_7: 8 x 8 outputs (half-empty tile columns), four stages
    (2, 20, 44, 64, 64),        # ragged: 10 x 22 outputs, partial tiles in both directions
    (1, 2, 2, 32, 64),          # one output pixel
]


def test_conv4x4s2_fused_bn_statistics_with_large_mean_small_variance_channels():
    """the conv epilogue takes the batch-norm statistics as single-pass moments of its fp32 accumulators (E[x^2] -
    mean^2, clamped at 0), where tf.nn.moments -- and the reduction kernels this path replaces -- are two-pass.  Cancellation
    grows with |mean| / std: channels with mean 8 and std ~0.1 (ratio 80, far beyond what D's pre-BN activations show: |mean| <
    std at initialisation and after training steps) must still give the variance to 1 % and the same normalised tensor as the
    two-pass path on the stored tensor.  (fp32: eps * mean^2 = 4e-6 absolute against a variance of 1e-2.)"""
    N, H, W, Cin, Cout = 8, 32, 32, 64, 64
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(4, 4, Cin, Cout, seed=2, scale=0.01).bfloat16()
    b = torch.full((Cout,), 8.0)
    b[::2] = -8.0
    wt = w.permute(0, 1, 3, 2).reshape(16, Cout, Cin).contiguous().to(DEV)
    wf = K.pack_taps_frag(wt, torch.empty_like(wt), 16, Cout, Cin)
    pre = (O.conv2(x.float(), w.float(), None, 2) + b).reshape(-1, Cout).double()
    d = K.conv_desc(N, H, W, Cin, H // 2, W // 2, Cout, 4, 4, 2, 1, 1, 0, TG_BF16, TG_BF16, ACT_NONE, 0.0)
    out = torch.empty(N, H // 2, W // 2, Cout, device=DEV, dtype=torch.bfloat16)
    rep = torch.zeros(K.BN_STAT_REPLICAS, 2, Cout, device=DEV)
    K.conv4x4s2_frag(d, x.to(DEV), wf, b.to(DEV), None, None, out, bn_stats=rep)
    beta = rnd(Cout, seed=9).to(DEV)
    y_fused = K.bn_lrelu_forward(out, torch.empty_like(out), beta, 1e-3, 0.2, rep, None, prezeroed=2)
    mean, var = pre.mean(0), pre.var(0, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 40.0                       # the regime the test is about
    assert float(((rep[0][0].double().cpu() - mean).abs() / mean.abs()).max()) < 1e-5
    rel = ((rep[0][1].double().cpu() - var).abs() / var).max().item()
    assert rel < 1e-2, "fused single-pass variance off by %.2e relative at |mean| / std = 80" % rel
    # (the normalised tensor itself is not compared here: at mean 8 the stored bf16 activation has a rounding step of 2^-4 = 0.6 std,
    #  which bounds what any statistic can reproduce; the regular cases above compare it with the two-pass path)
    assert torch.isfinite(y_fused.float()).all()


@pytest.mark.parametrize("case", K4_CASES)
def test_conv4x4s2_forward_matches_oracle(case):
    """conv2(net, 4, C, 2) of discriminator_F (lib/Teco.py:52-66; slim.conv2d k4 s2 SAME, lib/ops.py:47-56) on the fragment-order
    operand against the oracle's conv2 on the bf16-rounded tensors, with and without bias + LeakyReLU + residual."""
    N, H, W, Cin, Cout = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(4, 4, Cin, Cout, seed=2, scale=0.05).bfloat16()
    b = rnd(Cout, seed=3)
    res = rnd(N, H // 2, W // 2, Cout, seed=4).bfloat16()
    wt = w.permute(0, 1, 3, 2).reshape(16, Cout, Cin).contiguous().to(DEV)
    wf = K.pack_taps_frag(wt, torch.empty_like(wt), 16, Cout, Cin)
    pre = O.conv2(x.float(), w.float(), None, 2)
    for variant in (0, 1):
        d = K.conv_desc(N, H, W, Cin, H // 2, W // 2, Cout, 4, 4, 2, 1, 1, 0, TG_BF16, TG_BF16, ACT_LRELU if variant else ACT_NONE, 0.2)
        assert K.conv4x4s2_frag_ok(d)
        out = torch.full((N, H // 2, W // 2, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
        K.prof_collect()
        K.prof_enable(True)
        rep = torch.zeros(K.BN_STAT_REPLICAS, 2, Cout, device=DEV) if variant == 0 else None
        K.conv4x4s2_frag(d, x.to(DEV), wf, b.to(DEV) if variant else None, res.to(DEV) if variant else None, None, out, bn_stats=rep)
        K.prof_enable(False)
        if rep is not None:
            # the epilogue's batch statistics (mean, second moment over N Ho Wo; partial sets summed) and the batch norm built on
            # them against the two-reduction path on the stored bf16 tensor
            flat = pre.reshape(-1, Cout).double()
            close(rep.sum(0)[0], flat.mean(0).float(), 2e-5, "fused BN mean %s" % (case,))
            close(rep.sum(0)[1], (flat * flat).mean(0).float(), 2e-5, "fused BN second moment %s" % (case,))
            beta = rnd(Cout, seed=9).to(DEV)
            y_ref, st_ref = torch.empty_like(out), torch.zeros(2, Cout, device=DEV)
            K.bn_lrelu_forward(out, y_ref, beta, 1e-3, 0.2, st_ref, None, prezeroed=True)
            y_fused = K.bn_lrelu_forward(out, torch.empty_like(out), beta, 1e-3, 0.2, rep, None, prezeroed=2)
            stats = rep[0]
            close(stats[0], st_ref[0], 1e-3, "fused BN mean vs reduction kernels %s" % (case,))
            close(stats[1], st_ref[1], 2e-3, "fused BN variance vs reduction kernels %s" % (case,))
            if N * (H // 2) * (W // 2) >= 64:       # (with a handful of positions the variance is ~0 and rsqrt(eps) = 32 amplifies the
                #                                         bf16 rounding of the stored tensor, which only the reduction kernels see)
                close(y_fused, y_ref.float().cpu(), 2e-2, "batch norm on fused statistics %s" % (case,))
        ents = K.prof_collect()
        assert ents and ents[0]["name"].startswith("conv4x4s2_fwd"), ents
        ref = pre
        if variant:
            ref = O.lrelu(pre + b, 0.2) + res.float()
        # exact products, fp32 accumulation, ONE rounding to bf16 at the end: half a bf16 step of the result (2^-8 relative) plus the
        # fp32 summation-order noise of a 512..2048-term sum
        err = (out.float().cpu() - ref).abs()
        assert (err <= 4e-3 * ref.abs() + 2e-4).all(), "%s variant %d: max err %g" % (case, variant, err.max().item())
        # (tg_conv_forward's generic kernel for these descriptors only meets its own, wider tolerance of 1e-2 of the tensor maximum:
        # it is no reference for this bound -- profiles/r05f_pytest_k4.log)
        old = K.conv_forward(d, x.to(DEV), wt, b.to(DEV) if variant else None, res.to(DEV) if variant else None, None,
                             torch.empty_like(out))
        tight(old, ref, "tg_conv_forward (generic implicit-GEMM kernel) %s variant %d" % (case, variant))


@pytest.mark.parametrize("case", K4_CASES)
def test_conv4x4s2_input_gradient_matches_autograd(case):
    """Input gradient of the same convs (tf.gradients through lib/Teco.py:52-66): four output phases as 2x2-tap stride-1
    convolutions over one halo of dY, against torch autograd of the oracle's conv2; with the residual-gradient operand and the
    LeakyReLU mask of the layer below (the form the discriminator's backward pass uses at its first block)."""
    N, H, W, Cin, Cout = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16().float().requires_grad_()
    w = rnd(4, 4, Cin, Cout, seed=2, scale=0.05).bfloat16()
    gy = rnd(N, H // 2, W // 2, Cout, seed=5).bfloat16()
    O.conv2(x, w.float(), None, 2).backward(gy.float())
    res = rnd(N, H, W, Cin, seed=6).bfloat16()
    aux = rnd(N, H, W, Cin, seed=7).bfloat16()
    wn = w.reshape(16, Cin, Cout).contiguous().to(DEV)                  # HWIO as stored = [tap][Cout' = Cin][Cin' = Cout]
    wf = K.pack_taps_frag(wn, torch.empty_like(wn), 16, Cin, Cout)
    for variant in (0, 1):
        d = K.conv_desc(N, H // 2, W // 2, Cout, H, W, Cin, 4, 4, 2, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0,
                        ACT_LRELU if variant else ACT_NONE, 0.2)
        assert K.conv4x4s2_frag_ok(d) == (Cout % 32 == 0 and Cin % 64 == 0)
        if not K.conv4x4s2_frag_ok(d):
            return
        dx = torch.full((N, H, W, Cin), 7.0, device=DEV, dtype=torch.bfloat16)
        K.prof_collect()
        K.prof_enable(True)
        K.conv4x4s2_frag(d, gy.to(DEV), wf, None, res.to(DEV) if variant else None, aux.to(DEV) if variant else None, dx)
        K.prof_enable(False)
        ents = K.prof_collect()
        assert ents and ents[0]["name"].startswith("conv4x4s2_bwd"), ents
        ref = x.grad
        if variant:
            ref = (ref + res.float()) * torch.where(aux.float() > 0, 1.0, 0.2)
        err = (dx.float().cpu() - ref).abs()
        assert (err <= 4e-3 * ref.abs() + 2e-4).all(), "%s variant %d: max err %g" % (case, variant, err.max().item())
        old = K.conv_forward(d, gy.to(DEV), wn, None, res.to(DEV) if variant else None, aux.to(DEV) if variant else None,
                             torch.empty_like(dx))
        tight(old, ref, "tg_conv_forward (generic implicit-GEMM kernel), input gradient %s variant %d" % (case, variant))


@pytest.mark.parametrize("case", [(1, 270, 480, 64, 64), (2, 128, 128, 32, 64), (1, 133, 245, 64, 128)])
def test_deconv3x3s2_weights_in_registers_kernel(case):
    """slim.conv2d_transpose k3 s2 SAME (lib/ops.py:35-44) + bias + ReLU in the throughput regime: the four output phases as
    stride-1 sub-convolutions with the weights in registers (conv3x3_ws.hip), against the oracle's TF-aligned conv2_tran."""
    N, H, W, Cin, Cout = case
    x = rnd(N, H, W, Cin, seed=1).bfloat16()
    w = rnd(3, 3, Cout, Cin, seed=2, scale=0.1).bfloat16()               # TF layout [kh,kw,Cout,Cin]
    b = rnd(Cout, seed=3)
    ref = torch.relu(O.conv2_tran(x.float(), w.float(), b, 2))
    out = torch.full((N, 2 * H, 2 * W, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(N, H, W, Cin, 2 * H, 2 * W, Cout, 3, 3, 2, 0, 0, 1, TG_BF16, TG_BF16, ACT_RELU)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_forward(d, x.to(DEV), w.reshape(9, Cout, Cin).contiguous().to(DEV), b.to(DEV), None, None, out)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"] == "deconv3x3s2_ws", ents
    err = (out.float().cpu() - ref).abs()
    assert (err <= 8e-3 * ref.abs() + 2e-2).all(), "%s: max err %g" % (case, err.max().item())


# ---- csrc/conv_wgrad_tr.hip: weight gradients with transpose reads -------------------------------------------------------
@pytest.mark.skipif(os.environ.get("TG_WGRAD_TR") == "0", reason="TG_WGRAD_TR=0 switches the transpose-read kernel off")
@pytest.mark.parametrize("case", [(1, 2, 8), (3, 5, 32), (32, 76, 32)])
def test_wgrad_transpose_read_kernel_matches_autograd(case):
    """dW / dbias of 3x3 s1 SAME 64 -> 64 convs on 32-pixel-wide images (the generator trunk, lib/frvsr.py:50-57) against
    torch autograd on the bf16-rounded operands; several layers per launch, image counts that do and do not divide by the split."""
    G, N, H = case
    d = K.conv_desc(N, H, 32, 64, H, 32, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    xs = [rnd(N, H, 32, 64, seed=10 + g).bfloat16() for g in range(G)]
    ys = [rnd(N, H, 32, 64, seed=50 + g).bfloat16() for g in range(G)]
    got_w = [torch.full((3, 3, 64, 64), 0.5, device=DEV) for _ in range(G)]
    got_b = [torch.zeros(64, device=DEV) for _ in range(G)]
    K.prof_collect()
    K.prof_enable(True)
    K.conv_wgrad_grouped(d, [x.to(DEV) for x in xs], [y.to(DEV) for y in ys], got_w, got_b)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"] == "conv_wgrad_tr", ents
    for g in (0, G - 1):
        xr = xs[g].float().requires_grad_()
        w = torch.zeros(3, 3, 64, 64, requires_grad=True)
        b = torch.zeros(64, requires_grad=True)
        O.conv2(xr, w, b, 1).backward(ys[g].float())
        close(got_w[g] - 0.5, w.grad, 3e-4, "transpose-read dW layer %d %s" % (g, case))
        close(got_b[g], b.grad, 3e-4, "transpose-read dbias layer %d %s" % (g, case))


@pytest.mark.skipif(os.environ.get("TG_WGRAD_TR") == "0", reason="TG_WGRAD_TR=0 switches the transpose-read kernel off")
@pytest.mark.parametrize("case", [(2, 5, 32, 51, 56), (32, 76, 32, 51, 56), (3, 4, 16, 9, 16)])
def test_wgrad_transpose_read_kernel_with_a_narrower_last_group(case):
    """Round 5: generator_F's input conv (51 channels in a 56-channel pixel, lib/frvsr.py:47-49) as a narrower last group of the
    trunk's grouped transpose-read launch (tg_conv_wgrad_grouped_plus): its dW has 51 rows per tap, the pixel's 16-byte chunks past
    its 112 bytes read zeros.  Against torch autograd on the bf16-rounded operands; the trunk groups beside it must be unchanged."""
    G, N, H, A, ldx = case
    d = K.conv_desc(N, H, 32, 64, H, 32, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    xs = [rnd(N, H, 32, 64, seed=10 + g).bfloat16() for g in range(G)]
    ys = [rnd(N, H, 32, 64, seed=50 + g).bfloat16() for g in range(G)]
    xe = torch.zeros(N, H, 32, ldx)
    xe[..., :A] = rnd(N, H, 32, A, seed=7)
    xe, ye = xe.bfloat16(), rnd(N, H, 32, 64, seed=8).bfloat16()
    got_w = [torch.full((3, 3, 64, 64), 0.5, device=DEV) for _ in range(G)]
    got_b = [torch.zeros(64, device=DEV) for _ in range(G)]
    we, be = torch.full((3, 3, A, 64), 0.25, device=DEV), torch.zeros(64, device=DEV)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_wgrad_grouped_plus(d, [x.to(DEV) for x in xs], [y.to(DEV) for y in ys], got_w, got_b, (xe.to(DEV), ldx, A, ye.to(DEV), we, be))
    K.prof_enable(False)
    ents = K.prof_collect()
    assert len(ents) == 1 and ents[0]["name"] == "conv_wgrad_tr" and ents[0]["calls"] == 1, ents
    xr = xe[..., :A].float().requires_grad_()
    w = torch.zeros(3, 3, A, 64, requires_grad=True)
    b = torch.zeros(64, requires_grad=True)
    O.conv2(xr, w, b, 1).backward(ye.float())
    close(we - 0.25, w.grad, 3e-4, "narrow group dW %s" % (case,))
    close(be, b.grad, 3e-4, "narrow group dbias %s" % (case,))
    for g in (0, G - 1):
        xr = xs[g].float().requires_grad_()
        w = torch.zeros(3, 3, 64, 64, requires_grad=True)
        b = torch.zeros(64, requires_grad=True)
        O.conv2(xr, w, b, 1).backward(ys[g].float())
        close(got_w[g] - 0.5, w.grad, 3e-4, "trunk group %d beside the narrow one %s" % (g, case))
        close(got_b[g], b.grad, 3e-4, "trunk dbias %d %s" % (g, case))


@pytest.mark.skipif(os.environ.get("TG_WGRAD_TR") == "0", reason="TG_WGRAD_TR=0 switches the transpose-read kernel off")
@pytest.mark.parametrize("case", [(2, 8, 64, 64), (3, 24, 96, 64), (5, 16, 128, 3), (2, 128, 128, 3), (1, 8, 32, 8)])
def test_wgrad_transpose_read_kernel_wide_images_and_output_conv(case):
    """Round 3: the transpose-read weight-gradient kernel on images of any width that is a multiple of 32 (tiles of 8 x 32
    pixels, single-layer entry point) and its 64 -> (<= 8) instantiation for the generator's output conv (lib/frvsr.py:80):
    gradient tensor channel-padded to 8, the padding channels hold GARBAGE here and must not reach dW."""
    N, H, W, Co = case
    ldy = 64 if Co == 64 else 8
    x = rnd(N, H, W, 64, seed=3).bfloat16()
    y = rnd(N, H, W, ldy, seed=4).bfloat16()
    d = K.conv_desc(N, H, W, 64, H, W, Co, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16)
    dw = torch.full((3, 3, 64, Co), 0.25, device=DEV)
    db = torch.zeros(Co, device=DEV)
    K.prof_collect()
    K.prof_enable(True)
    K.conv_wgrad(d, x.to(DEV), y.to(DEV), dw, db, ldx=64, ldy=ldy)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert ents and ents[0]["name"] == ("conv_wgrad_tr" if Co == 64 else "conv_wgrad_tr_out"), ents
    xr = x.float().requires_grad_()
    w = torch.zeros(3, 3, 64, Co, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    O.conv2(xr, w, b, 1).backward(y.float()[..., :Co])
    close(dw - 0.25, w.grad, 3e-4, "transpose-read dW %s" % (case,))
    close(db, b.grad, 3e-4, "transpose-read dbias %s" % (case,))



def test_wgrad_multi_geometry_grouped_launch_matches_autograd():
    """tg_conv_wgrad_multi: weight / bias gradients of layers of DIFFERENT geometry in one launch (FNet's 14 convs,
    lib/frvsr.py:4-41 under tf.gradients): image sizes 32 / 16 / 8 / 4, channel counts 8(6) .. 256, padded channel strides,
    two layers sharing one geometry -- each against torch autograd on the bf16-rounded operands."""
    cases = [(5, 32, 32, 6, 8, 32), (5, 32, 32, 32, 32, 32), (5, 16, 16, 32, 32, 64), (5, 16, 16, 64, 64, 64), (5, 8, 8, 64, 64, 128),
             (5, 4, 4, 128, 128, 256), (5, 4, 4, 256, 256, 256), (5, 4, 4, 256, 256, 256), (5, 32, 32, 32, 32, 2)]
    descs, xs, ys, dws, dbs, lxs, lys, refs = [], [], [], [], [], [], [], []
    for i, (N, H, W, Ci, Cp, Co) in enumerate(cases):
        x = torch.zeros(N, H, W, Cp)
        x[..., :Ci] = rnd(N, H, W, Ci, seed=10 + i)
        x = x.bfloat16()
        Cop = (Co + 7) // 8 * 8
        y = torch.zeros(N, H, W, Cop)
        y[..., :Co] = rnd(N, H, W, Co, seed=40 + i)
        y = y.bfloat16()
        descs.append(K.conv_desc(N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16))
        xs.append(x.to(DEV))
        ys.append(y.to(DEV))
        dws.append(torch.full((3, 3, Ci, Co), 0.5, device=DEV))
        dbs.append(torch.zeros(Co, device=DEV))
        lxs.append(Cp)
        lys.append(Cop)
        xr = x.float()[..., :Ci].requires_grad_()
        w = torch.zeros(3, 3, Ci, Co, requires_grad=True)
        b = torch.zeros(Co, requires_grad=True)
        O.conv2(xr, w, b, 1).backward(y.float()[..., :Co])
        refs.append((w.grad, b.grad))
    K.prof_collect()
    K.prof_enable(True)
    K.conv_wgrad_multi(descs, xs, ys, dws, dbs, lxs, lys)
    K.prof_enable(False)
    ents = K.prof_collect()
    assert len(ents) == 1 and ents[0]["name"] == "conv_wgrad_row3_bf16_multi" and ents[0]["calls"] == 1, ents
    for i, (gw, gb) in enumerate(refs):
        close(dws[i] - 0.5, gw, 3e-4, "multi-geometry dW layer %d %s" % (i, cases[i]))
        close(dbs[i], gb, 3e-4, "multi-geometry dbias layer %d %s" % (i, cases[i]))


def test_wgrad_multi_strided_layers_share_one_launch_discriminator_geometry():
    """tg_conv_wgrad_multi on the discriminator's own-gradient pass (lib/Teco.py:35-39,52-66 under tf.gradients, both passes of a
    step as one batch): the four 4x4 stride-2 SAME convs 64->64, 64->64, 64->128, 128->256 go out as ONE launch of the per-tap
    bf16 kernel (conv_wgrad_bf16_multi), the 3x3 input conv (27 channels padded to 32) and the 1x1 dense layer (1 output
    channel) as their own launches -- each dW / dbias against torch autograd on the bf16-rounded operands."""
    N = 6
    cases = [(1, 1, 8, 8, 256, 256, 1), (4, 2, 16, 16, 128, 128, 256), (4, 2, 32, 32, 64, 64, 128), (4, 2, 64, 64, 64, 64, 64),
             (4, 2, 128, 128, 64, 64, 64), (3, 1, 128, 128, 27, 32, 64)]
    descs, xs, ys, dws, dbs, lxs, lys, refs = [], [], [], [], [], [], [], []
    for i, (k, s, H, W, Ci, Cp, Co) in enumerate(cases):
        x = torch.zeros(N, H, W, Cp)
        x[..., :Ci] = rnd(N, H, W, Ci, seed=10 + i)
        x = x.bfloat16()
        Ho, pt = K.same_pad(H, k, s)
        Wo, pl = K.same_pad(W, k, s)
        y = rnd(N, Ho, Wo, Co, seed=40 + i).bfloat16()
        has_b = k != 4                                                   # D's strided convs have no bias (batch-norm follows)
        descs.append(K.conv_desc(N, H, W, Ci, Ho, Wo, Co, k, k, s, pt, pl, 0, TG_BF16, TG_BF16))
        xs.append(x.to(DEV))
        ys.append(y.to(DEV))
        dws.append(torch.full((k, k, Ci, Co), 0.5, device=DEV))
        dbs.append(torch.zeros(Co, device=DEV) if has_b else None)
        lxs.append(Cp)
        lys.append(Co)
        xr = x.float()[..., :Ci].requires_grad_()
        w = torch.zeros(k, k, Ci, Co, requires_grad=True)
        b = torch.zeros(Co, requires_grad=True)
        (O.conv2(xr, w, b, s) * y.float()).sum().backward()
        refs.append((w.grad, b.grad if has_b else None))
    K.prof_collect()
    K.prof_enable(True)
    K.conv_wgrad_multi(descs, xs, ys, dws, dbs, lxs, lys)
    K.prof_enable(False)
    ents = {e["name"]: e["calls"] for e in K.prof_collect()}
    assert ents.get("conv_wgrad_bf16_multi") == 1 and "conv_wgrad_bf16<2>" not in ents, ents
    for i, (gw, gb) in enumerate(refs):
        close(dws[i] - 0.5, gw, 3e-4, "strided multi-layer dW layer %d %s" % (i, cases[i]))
        if gb is not None:
            close(dbs[i], gb, 3e-4, "strided multi-layer dbias layer %d %s" % (i, cases[i]))
    # a single strided layer still takes the one-layer launch, and the two agree
    dw1 = torch.zeros(4, 4, 64, 64, device=DEV)
    K.conv_wgrad(descs[4], xs[4], ys[4], dw1, None, ldx=64, ldy=64)
    close(dw1, refs[4][0], 3e-4, "single strided layer")


# ---- csrc/resblock_lat.hip: one launch per residual block of the training recurrence ------------------------------------------
RB_SHAPES = [(4, 32, 32), (2, 8, 8), (1, 5, 7), (3, 6, 6), (1, 13, 9), (1, 4, 4), (2, 3, 2)]


def _rb_setup(shape, seed=0):
    N, H, W = shape
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    x = bf(rnd(N, H, W, 64, seed=seed + 1))
    w1, w2 = bf(rnd(3, 3, 64, 64, seed=seed + 2, scale=0.1)), bf(rnd(3, 3, 64, 64, seed=seed + 3, scale=0.1))
    b1, b2 = rnd(64, seed=seed + 4, scale=0.3), rnd(64, seed=seed + 5, scale=0.3)
    return x, w1, w2, b1, b2


def _dev_bf(t):
    return t.to(DEV, torch.bfloat16).contiguous()


@pytest.mark.parametrize("frag", [False, True])
@pytest.mark.parametrize("shape", RB_SHAPES)
def test_resblock_one_launch_forward_is_bit_identical_to_two_launches_and_matches_oracle(shape, frag):
    """lib/frvsr.py:50-57: out = x + conv_2(relu(conv_1(x))).  One launch (tg_resblock) against (a) the two tg_conv_forward
    launches it replaces -- bit for bit, intermediate included -- and (b) the oracle on the bf16-rounded operands."""
    N, H, W = shape
    x, w1, w2, b1, b2 = _rb_setup(shape)
    wt = lambda w: w.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16)      # noqa: E731
    xd, w1t, w2t, b1d, b2d = _dev_bf(x), wt(w1), wt(w2), b1.to(DEV), b2.to(DEV)
    d1 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU)
    d2 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_NONE)
    r_ref, a_ref = torch.empty_like(xd), torch.empty_like(xd)
    K.conv_forward(d1, xd, w1t, b1d, None, None, r_ref)
    K.conv_forward(d2, r_ref, w2t, b2d, xd, None, a_ref)
    r, a = torch.full_like(xd, 7.0), torch.full_like(xd, 7.0)
    w1k, w2k = (K.frag_order(w1t), K.frag_order(w2t)) if frag else (w1t, w2t)       # fragment-order copies of the same operands
    K.resblock(0, xd, w1k, b1d, w2k, b2d, None, None, r, a, w_frag=frag)
    torch.cuda.synchronize()
    assert torch.equal(r.view(torch.int16), r_ref.view(torch.int16)), "intermediate differs from the two-launch path"
    assert torch.equal(a.view(torch.int16), a_ref.view(torch.int16)), "block output differs from the two-launch path"
    a2 = torch.full_like(xd, 7.0)
    K.resblock(0, xd, w1k, b1d, w2k, b2d, None, None, None, a2, w_frag=frag)   # stateless form: no intermediate written
    assert torch.equal(a2.view(torch.int16), a_ref.view(torch.int16))
    # per element against the oracle, each conv from the operand the kernel itself read (the second one from the stored r)
    tight(r, torch.relu(O.conv2(x, w1, b1, 1)), "resblock intermediate %s" % (shape,))
    tight(a, x + O.conv2(r.float().cpu(), w2, b2, 1), "resblock output %s" % (shape,))


@pytest.mark.parametrize("frag", [False, True])
@pytest.mark.parametrize("mask2", [False, True])
@pytest.mark.parametrize("shape", RB_SHAPES)
def test_resblock_one_launch_input_gradient_is_bit_identical_to_two_launches_and_matches_autograd(shape, mask2, frag):
    """tf.gradients through the block (lib/Teco.py:441-449): d r = bwd(conv_2)(g) * relu'(r), d x = (g + bwd(conv_1)(d r))
    [* relu'(a0) for the first block, whose input is the input stage's ReLU output]."""
    N, H, W = shape
    x, w1, w2, b1, b2 = _rb_setup(shape, seed=10)
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    g = bf(rnd(N, H, W, 64, seed=20))
    r_saved = bf(torch.relu(O.conv2(x, w1, b1, 1)))
    a0 = bf(rnd(N, H, W, 64, seed=21))
    wn = lambda w: w.reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16)   # HWIO as stored = [tap][in][out]      # noqa: E731
    gd, rd, a0d, w1n, w2n = _dev_bf(g), _dev_bf(r_saved), _dev_bf(a0), wn(w1), wn(w2)
    aux2 = a0d if mask2 else None
    dA = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    dB = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU if mask2 else ACT_NONE, 0.0)
    dr_ref, dx_ref = torch.empty_like(gd), torch.empty_like(gd)
    K.conv_forward(dA, gd, w2n, None, None, rd, dr_ref)
    K.conv_forward(dB, dr_ref, w1n, None, gd, aux2, dx_ref)
    dr, dx = torch.full_like(gd, 7.0), torch.full_like(gd, 7.0)
    w2k, w1k = (K.frag_order(w2n), K.frag_order(w1n)) if frag else (w2n, w1n)
    K.resblock(1, gd, w2k, None, w1k, None, rd, aux2, dr, dx, w_frag=frag)
    torch.cuda.synchronize()
    assert torch.equal(dr.view(torch.int16), dr_ref.view(torch.int16)), "d r differs from the two-launch path"
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), "d x differs from the two-launch path"
    # autograd on the bf16-rounded operands (the intermediate gradient rounded to bf16 where the kernels round it)
    rr = r_saved.clone().requires_grad_()
    O.conv2(rr, w2, None, 1).backward(g)
    dr_o = bf(rr.grad * (r_saved > 0).float())
    xx = torch.zeros(N, H, W, 64, requires_grad=True)
    O.conv2(xx, w1, None, 1).backward(dr_o)
    dx_o = g + xx.grad
    if mask2:
        dx_o = dx_o * (a0 > 0).float()
    tight(dr, rr.grad * (r_saved > 0).float(), "resblock d r %s" % (shape,))
    xk = torch.zeros(N, H, W, 64, requires_grad=True)
    O.conv2(xk, w1, None, 1).backward(dr.float().cpu())                       # from the kernel's own stored d r
    tight(dx, (g + xk.grad) * ((a0 > 0).float() if mask2 else 1.0), "resblock d x %s" % (shape,))
    close(dr, dr_o, 1e-2, "resblock d r %s" % (shape,))
    close(dx, dx_o, 1e-2, "resblock d x %s" % (shape,))


# ---- csrc/resblock_chain.hip: the whole residual trunk of a frame as ONE persistent launch -------------------------------------
RC_SHAPES = [(4, 32, 32), (2, 8, 8), (1, 5, 7), (3, 6, 6), (1, 13, 9), (1, 4, 4), (2, 3, 2), (1, 64, 60)]


@pytest.mark.parametrize("nb", [16, 5, 2, 1])
@pytest.mark.parametrize("shape", RC_SHAPES)
def test_resblock_chain_forward_is_bit_identical_to_per_block_launches_and_matches_oracle(shape, nb):
    """lib/frvsr.py:66-70: `for i in range(1, num_resblock + 1): net = residual_block(net, ...)`.  ONE launch (tg_resblock_chain:
    neighbour hand-offs instead of kernel boundaries) against nb x tg_resblock -- every intermediate and every block output bit for
    bit, launched three times on the same scratch (the epochs advance; odd lengths change the slot parity) -- and the last output
    against the oracle on bf16-rounded operands, the activations rounded to bf16 where the kernels store them."""
    N, H, W = shape
    if not K.resblock_chain_ok(N, H, W):
        pytest.skip("more 4x4 tiles than compute units: the engine runs tg_resblock per block there")
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    x = bf(rnd(N, H, W, 64, seed=31))
    ws = [bf(rnd(3, 3, 64, 64, seed=40 + i, scale=0.05)) for i in range(2 * nb)]
    bs = [rnd(64, seed=80 + i, scale=0.2) for i in range(2 * nb)]
    wt = lambda w: K.frag_order(w.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16))      # noqa: E731
    wf, bd, xd = [wt(w) for w in ws], [b.to(DEV) for b in bs], _dev_bf(x)
    r_ref = [torch.empty_like(xd) for _ in range(nb)]
    a_ref = [torch.empty_like(xd) for _ in range(nb)]
    a = xd
    for i in range(nb):
        a = K.resblock(0, a, wf[2 * i], bd[2 * i], wf[2 * i + 1], bd[2 * i + 1], None, None, r_ref[i], a_ref[i], w_frag=True)
    scratch = K.resblock_chain_scratch(N, H, W, DEV)
    for rep in range(3):
        r = [torch.full_like(xd, 7.0) for _ in range(nb)]
        o = [torch.full_like(xd, 7.0) for _ in range(nb)]
        K.resblock_chain(0, xd, wf[0::2], bd[0::2], wf[1::2], bd[1::2], None, None, r, o, scratch)
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0, "a workgroup gave up waiting for a neighbour"
        assert int(scratch[0]) == (rep + 1) * nb and int(scratch[1]) == 0            # epoch base advanced, arrivals reset
        for i in range(nb):
            assert torch.equal(r[i].view(torch.int16), r_ref[i].view(torch.int16)), "intermediate of block %d (launch %d)" % (i, rep)
            assert torch.equal(o[i].view(torch.int16), a_ref[i].view(torch.int16)), "output of block %d (launch %d)" % (i, rep)
    K.resblock_chain(0, xd, wf[0::2], bd[0::2], wf[1::2], bd[1::2], None, None, None, o, scratch)       # stateless: no intermediates
    torch.cuda.synchronize()
    assert torch.equal(o[-1].view(torch.int16), a_ref[-1].view(torch.int16))
    # ... and every stage per element against the oracle, from the operand the kernel itself read (exact products, fp32
    # accumulation, one rounding: the tight bound), plus the end-to-end chain on the oracle's own intermediates
    a_o, a_k = x, x
    for i in range(nb):
        tight(r_ref[i], torch.relu(O.conv2(a_k, ws[2 * i], bs[2 * i], 1)), "block %d intermediate %s" % (i, shape))
        tight(a_ref[i], a_k + O.conv2(r_ref[i].float().cpu(), ws[2 * i + 1], bs[2 * i + 1], 1), "block %d output %s" % (i, shape))
        a_k = a_ref[i].float().cpu()
        r_o = torch.relu(O.conv2(a_o, ws[2 * i], bs[2 * i], 1)).bfloat16().float()
        a_o = bf(a_o + O.conv2(r_o, ws[2 * i + 1], bs[2 * i + 1], 1))
    close(o[-1], a_o, 2e-2, "trunk output %s x %d" % (shape, nb))


@pytest.mark.parametrize("nb", [16, 2])
@pytest.mark.parametrize("shape", [(4, 32, 32), (1, 5, 7), (3, 6, 6), (1, 13, 9), (2, 8, 8)])
def test_resblock_chain_with_the_input_conv_in_front_is_bit_identical(shape, nb):
    """lib/frvsr.py:60-70: net = relu(conv2(gen_inputs, 3, 64, 1)); then the residual blocks.  The input-stage conv inside the trunk's
    launch (tg_resblock_chain pre_x: computed on the 8x8 region the first block needs, no exchange) against its own tg_conv_forward
    launch followed by nb x tg_resblock: a[0], every intermediate and every block output bit for bit; a[0] per element against the
    oracle.  The generator input has 51 channels in 56-channel pixels; the fragment-order weight copy pads them to 64 with zeros."""
    N, H, W = shape
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    xin = torch.zeros(N, H, W, 56)
    xin[..., :51] = bf(rnd(N, H, W, 51, seed=7))
    w_in, b_in = bf(rnd(3, 3, 51, 64, seed=8, scale=0.08)), rnd(64, seed=9, scale=0.2)
    ws = [bf(rnd(3, 3, 64, 64, seed=40 + i, scale=0.05)) for i in range(2 * nb)]
    bs = [rnd(64, seed=80 + i, scale=0.2) for i in range(2 * nb)]
    wt = lambda w: K.frag_order(w.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16))      # noqa: E731
    wf, bd = [wt(w) for w in ws], [b.to(DEV) for b in bs]
    xd = _dev_bf(xin)
    w_rows = torch.zeros(9, 64, 56)
    w_rows[:, :, :51] = w_in.permute(0, 1, 3, 2).reshape(9, 64, 51)
    d0 = K.conv_desc(N, H, W, 56, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU)
    a0_ref = K.conv_forward(d0, xd, w_rows.to(DEV, torch.bfloat16), b_in.to(DEV), None, None, torch.empty(N, H, W, 64, device=DEV, dtype=torch.bfloat16))
    w64 = torch.zeros(9, 64, 64)
    w64[:, :, :51] = w_in.permute(0, 1, 3, 2).reshape(9, 64, 51)
    win_f = K.frag_order(w64.to(DEV, torch.bfloat16))
    r_ref = [torch.empty_like(a0_ref) for _ in range(nb)]
    a_ref = [torch.empty_like(a0_ref) for _ in range(nb)]
    a = a0_ref
    for i in range(nb):
        a = K.resblock(0, a, wf[2 * i], bd[2 * i], wf[2 * i + 1], bd[2 * i + 1], None, None, r_ref[i], a_ref[i], w_frag=True)
    scratch = K.resblock_chain_scratch(N, H, W, DEV)
    for rep in range(2):
        a0 = torch.full_like(a0_ref, 7.0)
        r = [torch.full_like(a0_ref, 7.0) for _ in range(nb)]
        o = [torch.full_like(a0_ref, 7.0) for _ in range(nb)]
        K.resblock_chain(0, None, wf[0::2], bd[0::2], wf[1::2], bd[1::2], None, None, r, o, scratch, pre=(xd, win_f, b_in.to(DEV), a0))
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0
        assert torch.equal(a0.view(torch.int16), a0_ref.view(torch.int16)), "input-stage output differs from tg_conv_forward's"
        for i in range(nb):
            assert torch.equal(r[i].view(torch.int16), r_ref[i].view(torch.int16)), "intermediate of block %d" % i
            assert torch.equal(o[i].view(torch.int16), a_ref[i].view(torch.int16)), "output of block %d" % i
    tight(a0, torch.relu(O.conv2(xin[..., :51], w_in, b_in, 1)), "input-stage conv %s" % (shape,))


@pytest.mark.parametrize("nb", [16, 3])
@pytest.mark.parametrize("shape", [(4, 32, 32), (1, 5, 7), (3, 6, 6), (1, 13, 9)])
def test_resblock_chain_input_gradient_is_bit_identical_to_per_block_launches(shape, nb):
    """tf.gradients through the trunk (lib/Teco.py:441-449), blocks in reverse order: d r_i = bwd(conv_2)(g) * relu'(r_i),
    g <- g + bwd(conv_1)(d r_i), the last one masked by the input stage's ReLU.  One launch against nb x tg_resblock(mode 1)."""
    N, H, W = shape
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    g0 = _dev_bf(bf(rnd(N, H, W, 64, seed=51)))
    wn = lambda w: K.frag_order(w.reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16))      # noqa: E731
    wf = [wn(bf(rnd(3, 3, 64, 64, seed=60 + i, scale=0.05))) for i in range(2 * nb)]           # [conv_2, conv_1] per block processed
    rs = [_dev_bf(bf(rnd(N, H, W, 64, seed=100 + i))) for i in range(nb)]
    a0 = _dev_bf(bf(rnd(N, H, W, 64, seed=99)))
    dr_ref = [torch.empty_like(g0) for _ in range(nb)]
    dx_ref = [torch.empty_like(g0) for _ in range(nb)]
    g = g0
    for i in range(nb):
        g = K.resblock(1, g, wf[2 * i], None, wf[2 * i + 1], None, rs[i], a0 if i == nb - 1 else None, dr_ref[i], dx_ref[i], w_frag=True)
    scratch = K.resblock_chain_scratch(N, H, W, DEV)
    for rep in range(2):
        dr = [torch.full_like(g0, 7.0) for _ in range(nb)]
        dx = [torch.full_like(g0, 7.0) for _ in range(nb)]
        K.resblock_chain(1, g0, wf[0::2], None, wf[1::2], None, rs, a0, dr, dx, scratch)
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0
        for i in range(nb):
            assert torch.equal(dr[i].view(torch.int16), dr_ref[i].view(torch.int16)), "d r of block %d" % i
            assert torch.equal(dx[i].view(torch.int16), dx_ref[i].view(torch.int16)), "d x of block %d" % i


def test_resblock_chain_rejects_what_it_does_not_cover():
    from tecogan_amd._lib import TecoHipError
    x = torch.zeros(8, 32, 32, 64, device=DEV, dtype=torch.bfloat16)               # 512 tiles: more than compute units
    w = torch.zeros(9 * 64 * 64, device=DEV, dtype=torch.bfloat16)
    assert not K.resblock_chain_ok(8, 32, 32)
    with pytest.raises(TecoHipError):
        K.resblock_chain(0, x, [w, w], None, [w, w], None, None, None, None, [torch.empty_like(x), torch.empty_like(x)],
                         K.resblock_chain_scratch(8, 32, 32, DEV))
    xf = torch.zeros(1, 4, 4, 64, device=DEV)
    with pytest.raises(TecoHipError):                                              # fp32: the per-conv launches, not this kernel
        K.resblock_chain(0, xf, [w], None, [w], None, None, None, None, [torch.empty_like(xf)], K.resblock_chain_scratch(1, 4, 4, DEV))


# ---- csrc/resblock_plane.hip: the residual trunk of an INFERENCE frame as ONE persistent launch, activations resident in LDS --------
RP_SHAPES = [(1, 16, 32), (1, 48, 96), (1, 40, 70), (2, 33, 64), (1, 17, 33), (3, 5, 7), (1, 270, 480)]


@pytest.mark.parametrize("nb", [16, 3, 1])
@pytest.mark.parametrize("shape", RP_SHAPES)
def test_resblock_plane_is_bit_identical_to_per_block_launches_and_matches_oracle(shape, nb):
    """lib/frvsr.py:50-57,66-70 in the stateless forward (main.py:195-216).  ONE launch (tg_resblock_plane: 16x32-pixel tiles kept in
    LDS across all blocks, the one-pixel ring from the neighbour workgroups after every conv) against nb x tg_resblock bit for bit --
    launched three times on the same scratch (the epochs advance), both weight-prefetch variants, once in place (out = x) -- and
    per block against the oracle at the tight bound (exact products, fp32 accumulation, one rounding) from the operand the kernels
    read.  Shapes: one tile, 3 x 3 tiles, partial tiles in both directions, two images, one pixel past a tile, smaller than a tile,
    and the 1080p frame of BASELINE configs[4] (255 tiles)."""
    N, H, W = shape
    if shape == (1, 270, 480) and nb == 3:
        pytest.skip("the full frame runs with 16 blocks and with one")
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    x = bf(rnd(N, H, W, 64, seed=31))
    ws = [bf(rnd(3, 3, 64, 64, seed=40 + i, scale=0.05)) for i in range(2 * nb)]
    bs = [rnd(64, seed=80 + i, scale=0.2) for i in range(2 * nb)]
    wt = lambda w: K.frag_order(w.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16))      # noqa: E731
    wf, bd, xd = [wt(w) for w in ws], [b.to(DEV) for b in bs], _dev_bf(x)
    r_ref = [torch.empty_like(xd) for _ in range(nb)]
    a_ref = [torch.empty_like(xd) for _ in range(nb)]
    a = xd
    for i in range(nb):
        a = K.resblock(0, a, wf[2 * i], bd[2 * i], wf[2 * i + 1], bd[2 * i + 1], None, None, r_ref[i], a_ref[i], w_frag=True)
    scratch = K.resblock_plane_scratch(N, H, W, DEV)
    for rep in range(3):
        o = K.resblock_plane(xd, wf[0::2], bd[0::2], wf[1::2], bd[1::2], torch.full_like(xd, 7.0), scratch, variant=rep & 1)
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0, "a workgroup gave up waiting for a neighbour"
        assert int(scratch[0]) == (rep + 1) * 2 * nb and int(scratch[1]) == 0          # epoch base advanced, arrivals reset
        assert torch.equal(o.view(torch.int16), a_ref[-1].view(torch.int16)), "launch %d" % rep
    xa = xd.clone()
    K.resblock_plane(xa, wf[0::2], bd[0::2], wf[1::2], bd[1::2], xa, scratch)           # in place: every workgroup stages its tile first
    torch.cuda.synchronize()
    assert torch.equal(xa.view(torch.int16), a_ref[-1].view(torch.int16)), "in place"
    if H * W <= 48 * 96:                                                                # (the oracle on the CPU: small shapes only)
        a_k = x
        for i in range(nb):
            tight(r_ref[i], torch.relu(O.conv2(a_k, ws[2 * i], bs[2 * i], 1)), "block %d intermediate %s" % (i, shape))
            tight(a_ref[i], a_k + O.conv2(r_ref[i].float().cpu(), ws[2 * i + 1], bs[2 * i + 1], 1), "block %d output %s" % (i, shape))
            a_k = a_ref[i].float().cpu()


@pytest.mark.parametrize("nb", [16, 2])
@pytest.mark.parametrize("shape", [(1, 48, 96), (1, 40, 70), (2, 33, 64), (1, 270, 480)])
def test_resblock_plane_with_the_input_conv_in_front(shape, nb):
    """lib/frvsr.py:60-70: net = relu(conv2(gen_inputs, 3, 64, 1)); then the residual blocks.  The input-stage conv inside the trunk's
    launch (tg_resblock_plane pre_x: one more conv pass on the tile, its ring by one more hand-off).  a[0] is read out exactly through
    a one-block launch with zero block weights (x + 0): held per element against the oracle (small shapes) and against the generic
    launch; the trunk behind it bit for bit against nb x tg_resblock started from that a[0].  The generator input has 51 channels in
    56-channel pixels; the fragment-order weight copy pads them to 64 with zeros."""
    N, H, W = shape
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    xin = torch.zeros(N, H, W, 56)
    xin[..., :51] = bf(rnd(N, H, W, 51, seed=7))
    w_in, b_in = bf(rnd(3, 3, 51, 64, seed=8, scale=0.08)), rnd(64, seed=9, scale=0.2)
    ws = [bf(rnd(3, 3, 64, 64, seed=40 + i, scale=0.05)) for i in range(2 * nb)]
    bs = [rnd(64, seed=80 + i, scale=0.2) for i in range(2 * nb)]
    wt = lambda w: K.frag_order(w.permute(0, 1, 3, 2).reshape(9, 64, 64).contiguous().to(DEV, torch.bfloat16))      # noqa: E731
    wf, bd = [wt(w) for w in ws], [b.to(DEV) for b in bs]
    w64 = torch.zeros(3, 3, 64, 64)
    w64[:, :, :51] = w_in
    pre = (_dev_bf(xin), wt(w64), b_in.to(DEV))
    scratch = K.resblock_plane_scratch(N, H, W, DEV)
    new = lambda: torch.full((N, H, W, 64), 7.0, device=DEV, dtype=torch.bfloat16)      # noqa: E731
    zw = torch.zeros(9 * 64 * 64, device=DEV, dtype=torch.bfloat16)
    a0 = K.resblock_plane(None, [zw], None, [zw], None, new(), scratch, pre=pre)        # a[0] + conv(relu(conv(a[0], 0)), 0) = a[0]
    torch.cuda.synchronize()
    assert int(scratch[2]) == 0 and int(scratch[0]) == 3                                # one hand-off more than 2 per block
    if H * W <= 48 * 96:
        tight(a0, torch.relu(O.conv2(xin[..., :51], w_in, b_in, 1)), "a[0] inside the trunk launch %s" % (shape,))
    w56 = torch.zeros(3, 3, 56, 64)
    w56[:, :, :51] = w_in
    d = K.conv_desc(N, H, W, 56, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU)
    g0 = K.conv_forward(d, pre[0], w56.permute(0, 1, 3, 2).reshape(9, 64, 56).contiguous().to(DEV, torch.bfloat16), pre[2], None, None, new())
    tight(a0, g0.float().cpu(), "a[0] against its own launch %s" % (shape,), rel=8e-3)   # (summation orders may differ: one bf16 step at most)
    a = a0
    for i in range(nb):
        a = K.resblock(0, a, wf[2 * i], bd[2 * i], wf[2 * i + 1], bd[2 * i + 1], None, None, None, torch.empty_like(a), w_frag=True)
    for rep in range(2):
        o = K.resblock_plane(None, wf[0::2], bd[0::2], wf[1::2], bd[1::2], new(), scratch, pre=pre)
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0 and int(scratch[0]) == 3 + (rep + 1) * (2 * nb + 1)
        assert torch.equal(o.view(torch.int16), a.view(torch.int16)), "launch %d %s" % (rep, shape)


def test_resblock_plane_beside_other_work_and_what_it_rejects():
    """The launch needs every workgroup resident: it is refused (TG_EINVAL) when there are more 16x32 tiles than compute units, and
    `resblock_plane_ok` keeps the engine on the per-block kernel below half a chip's worth of tiles.  Beside a GEMM on a second
    stream (a workgroup becomes resident late) the result is still bit-identical, nobody gives up."""
    assert K.resblock_plane_ok(1, 270, 480) and not K.resblock_plane_ok(1, 144, 180) and not K.resblock_plane_ok(2, 270, 480)
    w = K.frag_order(_dev_bf(rnd(9, 64, 64, seed=1, scale=0.05)))
    x = _dev_bf(rnd(2, 270, 480, 64, seed=2))
    from tecogan_amd._lib import TecoHipError
    with pytest.raises(TecoHipError, match="more tiles than compute units"):
        K.resblock_plane(x, [w], None, [w], None, torch.empty_like(x), K.resblock_plane_scratch(2, 270, 480, DEV))
    xf = rnd(1, 16, 32, 64, seed=2).to(DEV)
    with pytest.raises(TecoHipError, match="bf16"):
        K.resblock_plane(xf, [w], None, [w], None, torch.empty_like(xf), K.resblock_plane_scratch(1, 16, 32, DEV))
    x = _dev_bf(rnd(1, 270, 480, 64, seed=3))
    ws = [K.frag_order(_dev_bf(rnd(9, 64, 64, seed=10 + i, scale=0.05))) for i in range(8)]
    a = x
    for i in range(4):
        a = K.resblock(0, a, ws[2 * i], None, ws[2 * i + 1], None, None, None, None, torch.empty_like(a), w_frag=True)
    scratch = K.resblock_plane_scratch(1, 270, 480, DEV)
    side, big = torch.cuda.Stream(), torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    for rep in range(4):
        with torch.cuda.stream(side):
            big @ big
        o = K.resblock_plane(x, ws[0::2], None, ws[1::2], None, torch.full_like(x, 7.0), scratch)
        torch.cuda.synchronize()
        assert int(scratch[2]) == 0 and torch.equal(o.view(torch.int16), a.view(torch.int16)), "launch %d beside a GEMM" % rep
    # the sticky give-up counter is what the engines check (Generator.handoff_give_ups -> check_handoffs): a non-zero count raises
    from tecogan_amd.infer import InferenceEngine
    eng = InferenceEngine(1, 16, 32, DEV, torch.bfloat16, use_graph=False)
    eng.G._plane_scratch[(1, 16, 32)] = K.resblock_plane_scratch(1, 16, 32, DEV)
    eng.check_handoffs()
    eng.G._plane_scratch[(1, 16, 32)][2] = 3
    with pytest.raises(RuntimeError, match="3 workgroup"):
        eng.check_handoffs()



def test_resblock_rejects_what_it_does_not_cover():
    from tecogan_amd._lib import TecoHipError
    x = torch.zeros(1, 4, 4, 64, device=DEV)
    w = torch.zeros(9, 64, 64, device=DEV)
    with pytest.raises(TecoHipError):
        K.resblock(0, x, w, None, w, None, None, None, None, torch.empty_like(x))          # fp32: two launches, not this kernel


def test_pack_weights_frag_matches_the_host_permutation_of_both_operands():
    """tg_pack_weights_frag (one launch after the Adam updates) against kernels.frag_order applied to the two row-order compute
    copies tg_pack_weights_both writes: forward operand [tap][out][in] and input-gradient operand [tap][in][out]."""
    from collections import OrderedDict
    from tecogan_amd import params as P
    ps = P.ParamStore(OrderedDict(generator=P.generator_spec(2)), DEV, torch.bfloat16)
    ps.load(P.init_values(P.generator_spec(2), 5))
    # 2 residual blocks x 2 convs, the two transposed convs and (round 6) the input conv, whose 51 input channels are zero-padded to 64
    assert len(ps.frag) == 7 and sum('/resblock_' in n for n in ps.frag) == 4 and sum('/conv_tran' in n for n in ps.frag) == 2
    for name in ps.frag:
        for tr in (True, False):
            e = ps.entries[name]
            if "/input_stage/" in name:
                w = ps.view(name).detach().float()                                   # HWIO [3,3,51,64]
                rows = torch.zeros(9, 64, 64, device=DEV)
                if tr:
                    rows[:, :, :e["A"]] = w.reshape(9, e["A"], 64).permute(0, 2, 1)   # [tap][out][in]
                else:
                    rows[:, :e["A"], :] = w.reshape(9, e["A"], 64)                    # [tap][in][out]
                rows = rows.bfloat16()
            else:
                rows = ps.packed(name, tr).view(9, 64, 64)
            assert torch.equal(ps.packed_frag(name, tr).view(torch.int16), K.frag_order(rows).view(torch.int16)), (name, tr)
    assert not P.ParamStore(OrderedDict(generator=P.generator_spec(1)), DEV, torch.float32).frag        # bf16 compute copies only


# ---- csrc/hr_bwd_lat.hip: the BPTT's HR tail (frame gradient -> g_out -> g_t2 -> g_t1) as one launch -----------------------
@pytest.mark.parametrize("shape", [(4, 64, 64), (1, 8, 8), (2, 6, 10), (1, 4, 8), (3, 5, 7), (1, 12, 24)])
def test_hr_tail_backward_one_launch_matches_the_three_launches_and_autograd(shape):
    """lib/frvsr.py:73-87 under tf.gradients: frame gradient -> (.)*2 -> output conv (64 -> 3) -> relu'(t2) -> conv_tran2 (k3 s2)
    -> relu'(t1).  One launch (tg_hr_tail_backward) against the three it replaces (tg_concat2_pad, tg_conv_forward in
    input-gradient form on the 8-channel kernel, tg_conv_forward in gather form): g_out and g_t2 bit for bit, g_t1 to bf16
    rounding (another kernel's accumulation order), and all three against autograd on the bf16-rounded operands."""
    N, H2, W2 = shape
    Ho, Wo = 2 * H2, 2 * W2
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    d_frame = rnd(N, Ho, Wo, 3, seed=1, scale=0.01)
    w_out = bf(rnd(3, 3, 64, 3, seed=2, scale=0.1))                       # HWIO
    w_tr = bf(rnd(3, 3, 64, 64, seed=3, scale=0.1))                       # TF conv2d_transpose layout [kh,kw,Cout,Cin]
    t2, t1 = bf(rnd(N, Ho, Wo, 64, seed=4)), bf(rnd(N, H2, W2, 64, seed=5))
    dev16 = lambda t: t.to(DEV, torch.bfloat16).contiguous()             # noqa: E731
    # compute copies as ParamStore packs them: output conv natural copy [tap][in][out pad 8]; conv_tran2 [tap][in][out] = wT
    wo_n = torch.zeros(9, 64, 8)
    wo_n[:, :, :3] = w_out.reshape(9, 64, 3)
    wo_n = dev16(wo_n)
    wtr_t = dev16(w_tr.reshape(9, 64, 64).permute(0, 2, 1))              # [tap][cin][cout]
    dfd, t2d, t1d = d_frame.to(DEV), dev16(t2), dev16(t1)
    # the three launches
    g_out_ref = K.concat2_pad(dfd, None, torch.empty(N, Ho, Wo, 8, device=DEV, dtype=torch.bfloat16), scale=2.0)
    dA = K.conv_desc(N, Ho, Wo, 8, Ho, Wo, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    g_t2_ref = torch.empty(N, Ho, Wo, 64, device=DEV, dtype=torch.bfloat16)
    K.conv_forward(dA, g_out_ref, wo_n, None, None, t2d, g_t2_ref)
    dB = K.conv_desc(N, Ho, Wo, 64, H2, W2, 64, 3, 3, 2, 0, 0, 0, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    g_t1_ref = torch.empty(N, H2, W2, 64, device=DEV, dtype=torch.bfloat16)
    K.conv_forward(dB, g_t2_ref, wtr_t, None, None, t1d, g_t1_ref)
    # one launch
    g_out, g_t2, g_t1 = (torch.full_like(t, 7.0) for t in (g_out_ref, g_t2_ref, g_t1_ref))
    K.hr_tail_backward(dfd, 2.0, wo_n, t2d, K.frag_order(wtr_t), t1d, g_out, g_t2, g_t1)
    torch.cuda.synchronize()
    assert torch.equal(g_out.view(torch.int16), g_out_ref.view(torch.int16)), "g_out differs"
    assert torch.equal(g_t2.view(torch.int16), g_t2_ref.view(torch.int16)), "g_t2 differs from the 8-channel kernel's"
    close(g_t1, g_t1_ref.float(), 1e-2, "g_t1 vs the gather-form kernel %s" % (shape,))
    # autograd on the bf16-rounded operands
    x2 = torch.zeros(N, Ho, Wo, 64, requires_grad=True)
    O.conv2(x2, w_out, None, 1).backward(bf(2.0 * d_frame))
    gt2_o = bf(x2.grad * (t2 > 0).float())
    tight(g_t2, x2.grad * (t2 > 0).float(), "g_t2 vs autograd %s" % (shape,), abs_=2e-5)
    x1 = torch.zeros(N, H2, W2, 64, requires_grad=True)
    O.conv2_tran(x1, w_tr, None, 2).backward(gt2_o)
    xk = torch.zeros(N, H2, W2, 64, requires_grad=True)
    O.conv2_tran(xk, w_tr, None, 2).backward(g_t2.float().cpu())              # from the kernel's own stored g_t2
    tight(g_t1, xk.grad * (t1 > 0).float(), "g_t1 vs autograd %s" % (shape,), abs_=2e-5)
    tight(g_t1_ref, xk.grad * (t1 > 0).float(), "g_t1 of the gather-form kernel vs autograd %s" % (shape,), abs_=2e-5)


# ---- csrc/hr_fwd_lat.hip: the transposed convs of the training recurrence as latency-regime launches -------------------------
@pytest.mark.parametrize("shape", [(4, 32, 32), (4, 64, 64), (1, 5, 7), (2, 4, 8), (3, 9, 17), (1, 6, 10)])
def test_deconv_latency_kernel_matches_oracle_and_the_generic_kernel(shape):
    """relu(conv2d_transpose k3 s2 SAME (x) + b), lib/ops.py:35-44 / lib/frvsr.py:73-78: the phase-form latency kernel with
    fragment-order weights against the oracle (TF alignment: output 2i + k <- input i) and the generic transposed-mode engine."""
    N, H1, W1 = shape
    x = rnd(N, H1, W1, 64, seed=1).bfloat16()
    wt = rnd(3, 3, 64, 64, seed=2, scale=0.06).bfloat16()                # TF conv2d_transpose layout [kh,kw,Cout,Cin]
    bt = rnd(64, seed=3, scale=0.1)
    ref = torch.relu(O.conv2_tran(x.float(), wt.float(), bt, 2))
    w_rows = wt.reshape(9, 64, 64).contiguous().to(DEV)                   # [tap][Cout][Cin]: the operand of the transposed mode
    out = torch.full((N, 2 * H1, 2 * W1, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    K.deconv_lat_forward(x.to(DEV), K.frag_order(w_rows), bt.to(DEV), out)
    tight(out, ref, "deconv latency kernel %s" % (shape,))
    d = K.conv_desc(N, H1, W1, 64, 2 * H1, 2 * W1, 64, 3, 3, 2, 0, 0, 1, TG_BF16, TG_BF16, ACT_RELU)
    gen = torch.empty_like(out)
    K.conv_forward(d, x.to(DEV), w_rows, bt.to(DEV), None, None, gen)
    tight(gen, ref, "transposed-mode engine (generic kernel) %s" % (shape,))


@pytest.mark.parametrize("shape", [(4, 64, 64), (1, 14, 30), (2, 22, 36), (1, 4, 8), (3, 6, 10), (1, 540, 960)])
def test_hr_tail_training_kernel_matches_oracle(shape):
    """t1 -> t2 = relu(conv2d_transpose) (stored) -> conv 64 -> 3 -> + bicubic_four(LR) -> *2-1 (lib/frvsr.py:73-87) in one
    launch for the training recurrence: t2 to bf16 rounding, the frame as the inference kernel is held (the staged t2 is rounded
    to bf16 exactly where the three-launch path stores it as bf16)."""
    N, h2, w2 = shape
    h, w = h2 // 2, w2 // 2
    t1 = rnd(N, h2, w2, 64, seed=1).bfloat16()
    wt = rnd(3, 3, 64, 64, seed=2, scale=0.06).bfloat16()
    bt = rnd(64, seed=3, scale=0.1)
    wo = rnd(3, 3, 64, 3, seed=4, scale=0.06).bfloat16()                 # HWIO
    bo = rnd(3, seed=5, scale=0.1)
    gen_in = rnd(N, h, w, 56, seed=6).bfloat16()
    t2_ref = torch.relu(O.conv2_tran(t1.float(), wt.float(), bt, 2))
    w_tran = wt.reshape(9, 64, 64).contiguous().to(DEV)
    w_out = wo.permute(0, 1, 3, 2).reshape(9, 3, 64).contiguous().to(DEV)
    t2 = torch.full((N, 2 * h2, 2 * w2, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    out = torch.full((N, 2 * h2, 2 * w2, 3), 7.0, device=DEV)
    K.hr_tail_train(t1.to(DEV), K.frag_order(w_tran), bt.to(DEV), w_out, bo.to(DEV), gen_in.to(DEV), t2, out)
    tight(t2, t2_ref, "hr_tail_train t2 %s" % (shape,))
    ref = O.preprocess(O.conv2(t2.float().cpu(), wo.float(), bo, 1) + O.bicubic_four(gen_in[..., :3].float()))   # from the kernel's own t2
    err = (out.cpu() - ref).abs()
    assert (err <= 2e-3 * ref.abs() + 2e-3).all(), "hr_tail_train frame %s: max err %g" % (shape, err.max().item())


@pytest.mark.parametrize("shape", [(1, 22, 36), (2, 40, 72), (1, 256, 512)])
def test_hr_tail_outputs_are_optional_and_consistent(shape):
    """The inference frame's use of the training kernel: t2 not stored (NULL), `state` = (frame + 1) / 2 exactly, frame and state
    independent of which of the three outputs are requested."""
    N, h2, w2 = shape
    t1 = rnd(N, h2, w2, 64, seed=1).bfloat16().to(DEV)
    w2t = rnd(9, 64, 64, seed=2, scale=0.06).bfloat16().to(DEV)
    w3 = rnd(9, 3, 64, seed=4, scale=0.06).bfloat16().to(DEV)
    bt, bo = rnd(64, seed=3, scale=0.1).to(DEV), rnd(3, seed=5, scale=0.1).to(DEV)
    gen_in = rnd(N, h2 // 2, w2 // 2, 56, seed=6).bfloat16().to(DEV)
    f2 = K.frag_order(w2t)
    ref = torch.full((N, 2 * h2, 2 * w2, 3), 7.0, device=DEV)
    t2 = torch.full((N, 2 * h2, 2 * w2, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    K.hr_tail_train(t1, f2, bt, w3, bo, gen_in, t2, ref)
    fr, st = torch.full_like(ref, 7.0), torch.full_like(ref, 7.0)
    K.hr_tail_train(t1, f2, bt, w3, bo, gen_in, None, fr, st)
    assert torch.equal(fr, ref) and torch.equal(st, fr * 0.5 + 0.5)
    st2 = torch.full_like(ref, 7.0)
    K.hr_tail_train(t1, f2, bt, w3, bo, gen_in, None, None, st2)
    assert torch.equal(st2, st)


@pytest.mark.parametrize("shape", [(1, 540, 960), (1, 22, 36), (2, 40, 72), (1, 9, 17)])
def test_hr_tail_tile_forms_are_bit_identical(shape, monkeypatch):
    """The fused tail's two tile forms -- 4 x 8 input pixels per workgroup (latency regime: the training recurrence) and 8 x 16
    (throughput regime: the 1080p inference frame, chosen from four rounds of such tiles per compute unit on) -- run the same
    MFMAs in the same order per element: t2, frame and state bit for bit, at the 1080p shape and at shapes with partial tiles."""
    N, h2, w2 = shape
    h2, w2 = h2 - h2 % 2, w2 - w2 % 2
    t1 = rnd(N, h2, w2, 64, seed=1).bfloat16().to(DEV)
    w2t = rnd(9, 64, 64, seed=2, scale=0.06).bfloat16().to(DEV)
    w3 = rnd(9, 3, 64, seed=4, scale=0.06).bfloat16().to(DEV)
    bt, bo = rnd(64, seed=3, scale=0.1).to(DEV), rnd(3, seed=5, scale=0.1).to(DEV)
    gen_in = rnd(N, h2 // 2, w2 // 2, 56, seed=6).bfloat16().to(DEV)
    f2 = K.frag_order(w2t)
    res = {}
    for tile in ("4", "8"):
        monkeypatch.setenv("TG_HR_TAIL_TILE", tile)
        t2 = torch.full((N, 2 * h2, 2 * w2, 64), 7.0, device=DEV, dtype=torch.bfloat16)
        fr = torch.full((N, 2 * h2, 2 * w2, 3), 7.0, device=DEV)
        st = torch.full_like(fr, 7.0)
        K.hr_tail_train(t1, f2, bt, w3, bo, gen_in, t2, fr, st)
        torch.cuda.synchronize()
        res[tile] = (t2, fr, st)
    for a, b, what in zip(res["4"], res["8"], ("t2", "frame", "state")):
        assert torch.equal(a, b), "%s %s" % (what, shape)


@pytest.mark.parametrize("mask", [False, True])
@pytest.mark.parametrize("shape", [(4, 32, 32), (1, 5, 7), (2, 4, 8), (3, 9, 17), (4, 64, 64)])
def test_deconv_latency_input_gradient_matches_autograd_and_the_generic_kernel(shape, mask):
    """Input gradient of conv2d_transpose k3 s2 SAME (the stride-2 gather, lib/ops.py:35-44 under tf.gradients) on the latency
    kernel with fragment-order weights: against autograd through the oracle and against the gather-form engine."""
    N, H, W = shape
    bf = lambda t: t.bfloat16().float()                                   # noqa: E731
    wt = bf(rnd(3, 3, 64, 64, seed=2, scale=0.06))                        # TF layout [kh,kw,Cout,Cin]
    dy = bf(rnd(N, 2 * H, 2 * W, 64, seed=3))
    aux = bf(rnd(N, H, W, 64, seed=4))
    x = torch.zeros(N, H, W, 64, requires_grad=True)
    O.conv2_tran(x, wt, None, 2).backward(dy)
    want = x.grad * ((aux > 0).float() if mask else 1.0)
    w_t = wt.reshape(9, 64, 64).permute(0, 2, 1).contiguous().to(DEV, torch.bfloat16)        # [tap][cin][cout]
    dyd, auxd = dy.to(DEV, torch.bfloat16), aux.to(DEV, torch.bfloat16)
    dx = torch.full((N, H, W, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    K.deconv_lat_backward(dyd, K.frag_order(w_t), auxd if mask else None, dx)
    tight(dx, want, "deconv latency input gradient %s" % (shape,))
    d = K.conv_desc(N, 2 * H, 2 * W, 64, H, W, 64, 3, 3, 2, 0, 0, 0, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU if mask else ACT_NONE, 0.0)
    gen = torch.empty_like(dx)
    K.conv_forward(d, dyd, w_t, None, None, auxd if mask else None, gen)
    tight(gen, want, "gather-form engine (generic kernel), input gradient %s" % (shape,))

#!/usr/bin/env python
"""Lane-level CPU emulation of csrc/hr_tail.hip (LDS images, fragment addresses, MFMA operand / accumulator lane layouts,
staging writes, output-conv groups, bicubic gather, store masks) against the oracle ops -- the check that can be made
without a GPU.  float32 arithmetic; the staged block is rounded to bf16 where the kernel rounds it.
    python tools/emu_hr_tail.py [N h2 w2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oracle.ops as O  # noqa: E402

ROWB, SY, SX = 144, 14, 30
KB = np.array([[0., 1., 0., 0.], [-0.10546875, 0.87890625, 0.26171875, -0.03515625], [-0.09375, 0.59375, 0.59375, -0.09375],
               [-0.03515625, 0.26171875, 0.87890625, -0.10546875]], np.float32)


def bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16().float().numpy()


def mfma(acc, A, B):
    """v_mfma_f32_16x16x32: A[lane] = 8 K-values of row (lane & 15), K-group lane >> 4; B likewise for column (lane & 15);
    acc[lane][r] = D[row 4 (lane >> 4) + r][column lane & 15]."""
    Am = np.zeros((16, 32), np.float32)
    Bm = np.zeros((32, 16), np.float32)
    for lane in range(64):
        fr, fg = lane & 15, lane >> 4
        Am[fr, 8 * fg:8 * fg + 8] = A[lane]
        Bm[8 * fg:8 * fg + 8, fr] = B[lane]
    D = Am @ Bm
    for lane in range(64):
        fr, fg = lane & 15, lane >> 4
        acc[lane] += D[4 * fg:4 * fg + 4, fr]


def main():
    N, h2, w2 = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 10, 18)
    h, w, Ho, Wo = h2 // 2, w2 // 2, 2 * h2, 2 * w2
    g = torch.Generator().manual_seed(0)
    t1 = bf16(torch.randn(N, h2, w2, 64, generator=g).numpy())
    wt = bf16(torch.randn(3, 3, 64, 64, generator=g).numpy() * 0.06)        # [kh,kw,Cout,Cin]
    bt = (torch.randn(64, generator=g) * 0.1).numpy()
    wo = bf16(torch.randn(3, 3, 64, 3, generator=g).numpy() * 0.06)         # HWIO
    bo = (torch.randn(3, generator=g) * 0.1).numpy()
    lr = bf16(torch.rand(N, h, w, 4, generator=g).numpy())                  # Cpad = 4
    # oracle
    t2 = torch.relu(O.conv2_tran(torch.from_numpy(t1), torch.from_numpy(wt), torch.from_numpy(bt), 2)).bfloat16().float()
    c = O.conv2(t2, torch.from_numpy(wo), torch.from_numpy(bo), 1)
    ref = O.preprocess(c + O.bicubic_four(torch.from_numpy(lr[..., :3]))).numpy()
    W2 = wt.reshape(9, 64, 64)                                              # [tap][Cout][Cin]
    W3 = np.transpose(wo, (0, 1, 3, 2)).reshape(9, 3, 64)                   # [tap][Cout][Cin]
    out = np.full((N, Ho, Wo, 3), 7.0, np.float32)
    written = np.zeros((N, Ho, Wo), int)
    tiles_y, tiles_x = (Ho + 1 + SY - 1) // SY, (Wo + 1 + SX - 1) // SX
    lanes = np.arange(64)
    frow, fg = lanes & 15, lanes >> 4
    for tile in range(N * tiles_y * tiles_x):
        tx, tq = tile % tiles_x, tile // tiles_x
        ty, n = tq % tiles_y, tq // tiles_y
        # ---- halo tile by DMA: slot S -> pixel S / 9 (dy, dx), chunk S % 9
        y0, x0 = ty * 7 - 2, tx * 15 - 2
        halo = np.zeros((26 * 1024,), np.float32).reshape(-1)               # element = one bf16 (2 bytes) -> index byte / 2
        halo = np.zeros(26 * 512, np.float32)
        for S in range(10 * 18 * 9):
            pix, cch = S // 9, S % 9
            dy, dx = pix // 18, pix % 18
            ok = cch < 8 and 0 <= y0 + dy < h2 and 0 <= x0 + dx < w2
            if ok:
                halo[S * 8:S * 8 + 8] = t1[n, y0 + dy, x0 + dx, cch * 8:cch * 8 + 8]
        stage = np.zeros(16 * 32 * (ROWB // 2), np.float32)
        yb, xb = ty * SY - 2, tx * SX - 2
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            cbase = wn * 32
            Afrag = ((wm * 4) * 18 + frow) * ROWB + fg * 16                  # byte offsets per lane
            for py in range(2):
                for px in range(2):
                    acc = np.zeros((4, 2, 64, 4), np.float32)
                    for kk in range(2):
                        for dyi in range(1 if py else 2):
                            for dxi in range(1 if px else 2):
                                ky, kx = (1 if py else 2 * dyi), (1 if px else 2 * dxi)
                                for i in range(4):
                                    addr = Afrag + ((i + 1 - dyi) * 18 + 1 - dxi) * ROWB + kk * 64
                                    Bop = np.stack([halo[a // 2:a // 2 + 8] for a in addr])
                                    for j in range(2):
                                        Aop = np.stack([W2[ky * 3 + kx, cbase + j * 16 + fr, kk * 32 + g_ * 8:kk * 32 + g_ * 8 + 8]
                                                        for fr, g_ in zip(frow, fg)])
                                        mfma(acc[i, j], Aop, Bop)
                    for i in range(4):
                        yl = 2 * (wm * 4 + i) + py
                        for lane in range(64):
                            xl = 2 * frow[lane] + px
                            yo, xo = yb + yl, xb + xl
                            inside = 0 <= yo < Ho and 0 <= xo < Wo
                            for j in range(2):
                                ch0 = cbase + j * 16 + fg[lane] * 4
                                v = np.maximum(acc[i, j, lane] + bt[ch0:ch0 + 4], 0) if inside else np.zeros(4, np.float32)
                                off = ((yl * 32 + xl) * ROWB + ch0 * 2) // 2
                                stage[off:off + 4] = bf16(v)
        for wave in range(4):
            for gi in range(7):
                g_ = wave + 4 * gi
                yl, cb = 1 + (g_ >> 1), (15 if (g_ & 1) else 1)
                acc = np.zeros((64, 4), np.float32)
                Bfrag = ((yl - 1) * 32 + cb - 1 + frow) * ROWB + fg * 16
                for kh in range(3):
                    for kw in range(3):
                        for kk in range(2):
                            addr = Bfrag + (kh * 32 + kw) * ROWB + kk * 64
                            Bop = np.stack([stage[a // 2:a // 2 + 8] for a in addr])
                            Aop = np.stack([W3[kh * 3 + kw, fr, kk * 32 + gg * 8:kk * 32 + gg * 8 + 8] if fr < 3 else np.zeros(8, np.float32)
                                            for fr, gg in zip(frow, fg)])
                            mfma(acc, Aop, Bop)
                for lane in range(64):
                    fr, gg = frow[lane], fg[lane]
                    yo, xo = yb + yl, xb + cb + fr
                    mine = (cb == 1 or fr >= 2) and 0 <= yo < Ho and 0 <= xo < Wo
                    if gg != 0 or not mine:
                        continue
                    part = np.zeros(3, np.float32)
                    li, lj = yo >> 2, xo >> 2
                    for q in range(4):                                       # the four lane groups
                        ry = min(max(li + q - 1, 0), h - 1)
                        for k in range(4):
                            rx = min(max(lj + k - 1, 0), w - 1)
                            part += KB[yo & 3][q] * KB[xo & 3][k] * lr[n, ry, rx, :3]
                    out[n, yo, xo] = (acc[lane, :3] + bo + part) * 2 - 1
                    written[n, yo, xo] += 1
    assert (written == 1).all(), "coverage: min %d max %d" % (written.min(), written.max())
    err = np.abs(out - ref)
    print("hr_tail emulation [%d,%d,%d]: max |err| %.3e (ref max %.3f), every output pixel written exactly once" %
          (N, h2, w2, err.max(), np.abs(ref).max()))
    assert err.max() < 2e-3


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Lane-level CPU emulation of csrc/conv_wgrad_tr.hip (DMA slot images, transpose-read addressing under the documented
semantic of ds_read_b64_tr_b16, MFMA lane layouts, split-K) against torch autograd.
    python tools/emu_wgrad_tr.py [N H nsplit]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import oracle.ops as O  # noqa: E402

W, TH, PIX = 32, 8, 128
XSLOTS, XINST, YOFF = (TH + 2) * (W + 2) * 8, 43, 43 * 1024


def tr_read(lds, addr):
    """ds_read_b64_tr_b16 for one wave: addr[64] byte addresses; lane i of a 16-lane group receives M[4j + i/4][i%4], j = 0..3,
    with M[L] = the 4 consecutive 16-bit elements at lane L's address."""
    M = np.stack([lds[a // 2:a // 2 + 4] for a in addr])                    # [64][4]
    out = np.zeros((64, 4), np.float32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for j in range(4):
            out[lane, j] = M[16 * g + 4 * j + i // 4][i % 4]
    return out


def tr_frag(lds, addr):
    return np.concatenate([tr_read(lds, addr), tr_read(lds, addr + 4 * PIX)], axis=1)     # [64][8]


def mfma(acc, A, B):
    Am, Bm = np.zeros((16, 32), np.float32), np.zeros((32, 16), np.float32)
    for lane in range(64):
        fr, fg = lane & 15, lane >> 4
        Am[fr, 8 * fg:8 * fg + 8] = A[lane]
        Bm[8 * fg:8 * fg + 8, fr] = B[lane]
    D = Am @ Bm
    for lane in range(64):
        fr, fg = lane & 15, lane >> 4
        acc[lane] += D[4 * fg:4 * fg + 4, fr]


def main():
    N, H, nsplit = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 8, 2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, 64, generator=g).bfloat16().float()
    dy = torch.randn(N, H, W, 64, generator=g).bfloat16().float()
    xr = x.clone().requires_grad_()
    w = torch.zeros(3, 3, 64, 64, requires_grad=True)
    b = torch.zeros(64, requires_grad=True)
    O.conv2(xr, w, b, 1).backward(dy)
    xn, yn = x.numpy(), dy.numpy()
    dw, db = np.zeros((9, 64, 64), np.float32), np.zeros(64, np.float32)
    ntiles, tiles_y = N * (H // TH), H // TH
    lanes = np.arange(64)
    frow, fg = lanes & 15, lanes >> 4
    lp, lc = 8 * fg + (frow >> 2), 4 * (frow & 3)
    for split in range(nsplit):
        acc = np.zeros((4, 9, 2, 2, 64, 4), np.float32)                     # [wave][tap][i][j][lane][r]
        accb = np.zeros((4, 2, 64, 4), np.float32)
        for tile in range(split, ntiles, nsplit):
            n, y0 = tile // tiles_y, (tile % tiles_y) * TH
            lds = np.zeros((XINST + 32) * 512, np.float32)                  # element index = byte / 2
            for S in range(XINST * 64):                                     # X halo slots
                q, c = S >> 3, S & 7
                d_y, d_x = q // (W + 2), q % (W + 2)
                ok = S < XSLOTS and 0 <= y0 + d_y - 1 < H and 0 <= d_x - 1 < W
                if ok:
                    lds[S * 8:S * 8 + 8] = xn[n, y0 + d_y - 1, d_x - 1, c * 8:c * 8 + 8]
            lds[YOFF // 2:YOFF // 2 + TH * W * 64] = yn[n, y0:y0 + TH].reshape(-1)           # dY tile: contiguous copy
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                abase = lp * PIX + (32 * wm + lc) * 2
                bbase = YOFF + lp * PIX + (32 * wn + lc) * 2
                for yy in range(TH):
                    bq = [tr_frag(lds, bbase + yy * W * PIX + j * 32) for j in range(2)]
                    if wm == 0:
                        for j in range(2):
                            mfma(accb[wave, j], np.ones((64, 8), np.float32), bq[j])
                    for kh in range(3):
                        for kw in range(3):
                            aq = [tr_frag(lds, abase + ((yy + kh) * (W + 2) + kw) * PIX + i * 32) for i in range(2)]
                            for i in range(2):
                                for j in range(2):
                                    mfma(acc[wave, kh * 3 + kw, i, j], aq[i], bq[j])
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            for t in range(9):
                for i in range(2):
                    for j in range(2):
                        for lane in range(64):
                            for r in range(4):
                                ci, co = 32 * wm + 16 * i + 4 * fg[lane] + r, 32 * wn + 16 * j + frow[lane]
                                dw[t, ci, co] += acc[wave, t, i, j, lane, r]
            if wm == 0:
                for j in range(2):
                    for lane in range(16):                                  # fg == 0
                        db[32 * wn + 16 * j + lane] += accb[wave, j, lane, 0]
    ew = np.abs(dw.reshape(3, 3, 64, 64) - w.grad.numpy()).max() / np.abs(w.grad.numpy()).max()
    eb = np.abs(db - b.grad.numpy()).max() / np.abs(b.grad.numpy()).max()
    print("wgrad_tr emulation N=%d H=%d nsplit=%d: dW rel err %.2e, dbias rel err %.2e" % (N, H, nsplit, ew, eb))
    assert ew < 1e-5 and eb < 1e-5


if __name__ == "__main__":
    main()

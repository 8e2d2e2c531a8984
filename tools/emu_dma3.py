#!/usr/bin/env python
"""CPU checks of conv3x3_dma3_kernel (csrc/conv3x3_dma.hip, the three-weight-buffer variant): (1) every fragment read hits the
LDS slot the DMA filled with exactly that (pixel | weight row, channel group) -- swizzle, buffer offsets, immediates;
(2) the counted-vmcnt protocol: per wave, with in-order retirement, the halo and weight panel of a stage have landed at the
stage's wait, and no DMA targets a buffer that is read in the current or the next stage."""
HW, HALO, HB, WI = 18, 324, 21 * 1024, 36
WOFF0 = 2 * HB


def check_addressing():
    for hb in range(2):
        for wb in range(3):
            lds = {}
            for wave in range(8):
                for k in range(3):                                           # halo rounds
                    inst = wave + 8 * k
                    if k + 1 < 3 or inst < 21:
                        for lane in range(64):
                            S = inst * 64 + lane
                            q, ch = S >> 2, (S & 3) ^ (((S >> 4) & 1) << 1)
                            lds[(hb * HB + inst * 1024) // 16 + lane] = ("A", q, ch) if q < HALO else ("Z",)
                for k in range(5):                                           # weight rounds
                    inst = wave + 8 * k
                    if k + 1 < 5 or inst < WI:
                        for lane in range(64):
                            S = inst * 64 + lane
                            q, ch = S >> 2, (S & 3) ^ (((S >> 4) & 1) << 1)
                            lds[(WOFF0 + wb * WI * 1024 + inst * 1024) // 16 + lane] = ("W", q, ch)
            for wave in range(8):
                wm, wn = wave >> 1, wave & 1
                for lane in range(64):
                    frow, fg = lane & 15, lane >> 4
                    Q0 = wm * 4 * HW + frow
                    abase = [Q0 * 64 + ((fg ^ (((((Q0 & 7) + d) >> 2) & 1) << 1)) << 4) for d in range(8)]
                    bbase = WOFF0 + (wn * 32 + frow) * 64 + ((fg ^ (((frow >> 2) & 1) << 1)) << 4)
                    for kw in range(3):
                        for r in range(6):
                            K = r * HW + kw
                            a = hb * HB + abase[K & 7] + K * 64
                            assert lds.get(a // 16) == ("A", (wm * 4 + r) * HW + kw + frow, fg), (hb, wb, wave, lane, kw, r)
                        for kh in range(3):
                            for j in range(2):
                                a = wb * WI * 1024 + bbase + ((kh * 3 + kw) * 64 + j * 16) * 64
                                assert lds.get(a // 16) == ("W", (kh * 3 + kw) * 64 + wn * 32 + j * 16 + frow, fg)


def check_protocol(ntiles, nchunk, wave):
    nw = 5 if wave < 4 else 4
    nh = sum(1 for k in range(3) if k < 2 or wave + 8 * k < 21)
    stages = [(t, c) for t in range(ntiles) for c in range(nchunk)]
    q, landed = [], set()

    def issue(tag, n):
        q.extend([tag] * n)

    def wait_all_but(keep):
        while len(q) > keep:
            landed.add(q.pop(0))

    S = len(stages)
    issue(("H", 0), nh)
    issue(("W", 0), nw)
    young = 0
    if S > 1:
        issue(("W", 1), nw)
        young = nw
    prev_epi, hb, wb = False, 0, 0
    hbuf, wbuf = {0: 0}, {0: 0, 1: 1}
    for s in range(S):
        has1, has2 = s + 1 < S, s + 2 < S
        wait_all_but(young + (8 if prev_epi else 0))
        assert ("H", s) in landed and ("W", s) in landed and ("H", s) not in q and ("W", s) not in q
        assert hbuf[s] == hb and wbuf[s] == wb
        wb2, wb1 = (2 if wb == 0 else wb - 1), (0 if wb == 2 else wb + 1)
        if has1:
            issue(("H", s + 1), nh)
            hbuf[s + 1] = hb ^ 1
        if has2:
            issue(("W", s + 2), nw)
            wbuf[s + 2] = wb2
            assert wb2 != wb and wb2 != wb1
        young = nw if has2 else 0
        prev_epi = stages[s][1] == nchunk - 1
        if prev_epi:
            issue(("ST", s), 8)
        hb ^= 1
        wb = wb1


if __name__ == "__main__":
    check_addressing()
    for w in range(8):
        for T in (1, 2, 3, 5):
            for nc in (2, 3, 8, 16):
                check_protocol(T, nc, w)
    print("conv3x3_dma3: LDS addressing and counted-vmcnt protocol consistent")

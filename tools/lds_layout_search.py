"""Brute-force search for LDS layouts whose MFMA B-fragment reads (ds_read_b128: lane = pixel frow, 16-byte chunk fg) are
conflict-free under the gfx950 lane grouping of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32; 64 banks x 4 B,
/opt/skills/guides/MI355X_MICROARCH.md, LDS table) for EVERY tap of a 3x3 conv read from a haloed region -- the layouts
of csrc/resblock_lat.hip: byte pitch P per position, row pitch Ri positions, optional XOR swizzle of the chunk index, and the
lane -> pixel assignment of an MFMA pixel tile (4x4 block / two rows of 8).  Pure Python, no GPU."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[32 + i for i in g] for g in GROUPS]


def conflicts(addr_of_lane):
    """64 byte addresses (16-byte aligned) -> (worst n-way conflict over the four lane groups, sum of LDS cycles)."""
    worst = tot = 0
    for g in GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addr_of_lane[l] // 16) % 16, set()).add(addr_of_lane[l])
        w = max(len(v) for v in slots.values())
        worst, tot = max(worst, w), tot + w
    return worst, tot


def check(tiles, Ri, P, swz):
    """tiles: list of lists of 16 (ry,rx) output coords. swz(y,x)->xor"""
    worst=0; tot=0; n=0
    for tl in tiles:
        for ky in range(3):
            for kx in range(3):
                for kk in range(2):
                    addrs=[]
                    for lane in range(64):
                        frow, fg = lane&15, lane>>4
                        ry, rx = tl[frow]
                        y, x = ry+ky, rx+kx
                        pos = y*Ri + x
                        chunk = (kk*4+fg) ^ swz(y,x)
                        addrs.append(pos*P + chunk*16)
                    w,tt = conflicts(addrs)
                    worst=max(worst,w); tot+=tt; n+=4
    return worst, tot/n
def main():
    swzs = {"none": lambda y,x:0, "x&1": lambda y,x:x&1, "y&1": lambda y,x:y&1, "(x^y)&1": lambda y,x:(x^y)&1,
            "x&3": lambda y,x:x&3, "y&3":lambda y,x:y&3, "(y&1)*2": lambda y,x:(y&1)*2, "(y&3)*2": lambda y,x:(y&3)*2 & 7,
            "(y&1)*2+(x&1)": lambda y,x:(y&1)*2+(x&1), "(y&1)*4": lambda y,x:(y&1)*4, "(y&1)*4 ^ (x&1)": lambda y,x:((y&1)*4)^(x&1),
            "y&7": lambda y,x:y&7, "(y*2)&7 ^ (x&1)": lambda y,x: ((y*2)&7)^(x&1), "(y>>1&1)*2": lambda y,x:((y>>1)&1)*2}
    def t44(): return [[(f>>2, f&3) for f in range(16)]]
    def t28(rows): return [[(2*t+(f>>3), f&7) for f in range(16)] for t in range(rows//2)]
    def t116(rows): return [[(t, f) for f in range(16)] for t in range(rows)]
    shapes = {"4x4 (So=4)": t44(), "2x8 x3 (So=6)": t28(6), "2x8 x4 (So=8)": t28(8), "2x8x5 + (So=10, 2 tiles per row pair)": None}
    for name, tiles in shapes.items():
        if tiles is None: continue
        print("==", name)
        for Ri in range(6, 21):
            for P in (128,144,160,176,192):
                for sn, f in swzs.items():
                    w, avg = check(tiles, Ri, P, f)
                    if w==1: print("  Ri=%d P=%d swz=%s OK" % (Ri,P,sn))


if __name__ == "__main__":
    main()

"""tg_resblock_chain (csrc/resblock_chain.hip): the residual trunk of a frame as ONE persistent launch with neighbour hand-offs,
against nb x tg_resblock -- bit-identity (idle, under uneven load from a second stream, repeated launches: the epochs advance),
time per block in a graph chain for every variant, and (--trace, private -DTG_RC_TRACE build) the cycle stamps of the middle
workgroup: where a block's time goes and what the hand-off edge costs.
    python tools/mb_chain.py --build   (here, cross-compiles the trace library)      python tools/mb_chain.py [--trace]   (GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tecogan_amd import build as B  # noqa: E402

so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_rc.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    B.build(verbose=False)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "resblock_chain_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_RC_TRACE", "-c", os.path.join(csrc, "resblock_chain.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "resblock_chain.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
TRACE = "--trace" in sys.argv
import tecogan_amd._lib as L  # noqa: E402

if TRACE:
    L.LIB_PATH = so
import torch  # noqa: E402

from microbench import timeit  # noqa: E402
from tecogan_amd import kernels as K  # noqa: E402

torch.manual_seed(0)
DEV = "cuda"
N, H, W, NB = 4, 32, 32, 16
bf = torch.bfloat16
x = torch.randn(N, H, W, 64, device=DEV).to(bf)
wrow = [(torch.randn(9, 64, 64, device=DEV) * 0.03).to(bf) for _ in range(2 * NB)]
wf = [K.frag_order(w) for w in wrow]
bias = [torch.randn(64, device=DEV) * 0.1 for _ in range(2 * NB)]
aux = [torch.randn(N, H, W, 64, device=DEV).to(bf) for _ in range(NB + 1)]


def bufs():
    return [torch.full((N, H, W, 64), 7.0, device=DEV, dtype=bf) for _ in range(NB)]


def run_ref(mode, mid, out, nb=NB):
    a = x
    for i in range(nb):
        if mode == 0:
            a = K.resblock(0, a, wf[2 * i], bias[2 * i], wf[2 * i + 1], bias[2 * i + 1], None, None, mid[i], out[i], w_frag=True)
        else:
            a = K.resblock(1, a, wf[2 * i], None, wf[2 * i + 1], None, aux[i], aux[NB] if i == nb - 1 else None, mid[i], out[i], w_frag=True)


def chain_args(mode, mid, out, scratch, variant, nb=NB):
    if mode == 0:
        return K.ChainArgs(0, x, wf[0:2 * nb:2], bias[0:2 * nb:2], wf[1:2 * nb:2], bias[1:2 * nb:2], None, None, mid[:nb], out[:nb], scratch, variant)
    return K.ChainArgs(1, x, wf[0:2 * nb:2], None, wf[1:2 * nb:2], None, aux[:nb], aux[NB], mid[:nb], out[:nb], scratch, variant)


WORST = [0.0]


def same(a, b):
    """Bit-identical, or (the eight-wave kernel: two partial sums per conv) within 2 % of the tensor maximum after 16 blocks --
    the figure is printed; the per-element bound against the oracle is the tests' business."""
    ok = True
    for p, q in zip(a, b):
        if not torch.equal(p.view(torch.int16), q.view(torch.int16)):
            d = float((p.float() - q.float()).abs().max() / q.float().abs().max().clamp_min(1e-30))
            WORST[0] = max(WORST[0], d)
            ok = ok and d < 2e-2
    return ok


VARIANTS = [(14 << 1, "prefetch distance 14"), (0, "prefetch distance 28"), (63 << 1, "all loads in level 1")]
print("residual trunk [%d,%d,%d,64] x %d blocks: ONE persistent launch (tg_resblock_chain) against %d launches (tg_resblock)" % (N, H, W, NB, NB))
scratch = K.resblock_chain_scratch(N, H, W, DEV)
side = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=DEV, dtype=bf)
for mode, label in ((0, "forward"), (1, "input gradient")):
    mid_r, out_r = bufs(), bufs()
    run_ref(mode, mid_r, out_r)
    torch.cuda.synchronize()
    for variant, vlabel in VARIANTS:
        ok_idle = ok_load = True
        detail = []
        for rep in range(12):
            mid_c, out_c = bufs(), bufs()
            ca = chain_args(mode, mid_c, out_c, scratch, variant)
            if rep >= 3:                          # uneven load: a GEMM on a second stream occupies part of the chip meanwhile
                with torch.cuda.stream(side):
                    big @ big
            import time
            t0 = time.time()
            ca.launch()
            torch.cuda.synchronize()
            good = same(mid_c, mid_r) and same(out_c, out_r)
            detail.append("%s%d/%.0fms" % ("ok" if good else "BAD", int(scratch[2]), (time.time() - t0) * 1e3))
            if rep < 3:
                ok_idle &= good
            else:
                ok_load &= good
        nbs_ok = True
        for nb in (1, 2, 5):                      # shorter chains (odd lengths change the slot parity of the next launch)
            mid_c, out_c = bufs(), bufs()
            mid_s, out_s = bufs(), bufs()
            run_ref(mode, mid_s, out_s, nb)
            chain_args(mode, mid_c, out_c, scratch, variant, nb).launch()
            torch.cuda.synchronize()
            nbs_ok &= same(mid_c[:nb], mid_s[:nb]) and same(out_c[:nb], out_s[:nb])
        print("  %-14s %-30s equal to the per-block launches (worst relative difference %.1e; 0 = bit-identical): idle %s, beside a GEMM %s, "
              "1/2/5-block chains %s; give-ups %d, epoch %d  [%s]"
              % (label, vlabel, WORST[0], ok_idle, ok_load, nbs_ok, int(scratch[2]), int(scratch[0]), " ".join(detail)))
        WORST[0] = 0.0

if TRACE:
    VARIANTS += [(1 << 11, "two polls in flight"), (2 << 11, "one poll at a time, after 128"), (3 << 11, "one poll at a time, after 256"),
                 (256, "NO WEIGHT STREAM, distance 28"), (512, "NO WEIGHT LOADS ISSUED"), (512 | 1024, "... AND NO LEVEL-1 LDS READS")]
print("time per block, graph of 4 trunks x %d blocks (each trunk = one frame's chain):" % NB)
for mode, label in ((0, "forward"), (1, "input gradient")):
    mid, out = bufs(), bufs()

    def ref(mode=mode, mid=mid, out=out):
        run_ref(mode, mid, out)
    for _ in range(3):
        ref()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            ref()
    us = timeit(g.replay, 30, 5) / (4 * NB)
    print("  %-14s %-30s %.2f us per block" % (label, "%d launches (tg_resblock)" % NB, us))
    for variant, vlabel in VARIANTS:
        ca = chain_args(mode, mid, out, scratch, variant)
        for _ in range(3):
            ca.launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4):
                ca.launch()
        us = timeit(g.replay, 30, 5) / (4 * NB)
        torch.cuda.synchronize()
        print("  %-14s %-30s %.2f us per block   (give-ups %d)" % (label, vlabel, us, int(scratch[2])))
        if TRACE:
            lib = C.CDLL(so)
            lib.tg_debug_rc_trace.argtypes = [C.POINTER(C.c_ulonglong)]
            buf = (C.c_ulonglong * (8 * 16 * 8))()
            assert lib.tg_debug_rc_trace(buf) == 0
            t = list(buf)
            names = ["level-1 MFMAs", "epilogue 1 + barrier A", "level-2 MFMAs", "epilogue 2 + publish", "sweep (hand-off wait)", "barrier B"]
            for wv in (0, 3):
                rows = [t[(wv * 16 + k) * 8:(wv * 16 + k) * 8 + 7] for k in range(NB)]
                for k in (1, 7, 14):
                    r = rows[k]
                    st = t[(wv * 16 + k) * 8 + 7]
                    print("      wave %d block %2d: " % (wv, k) + "  ".join("%s %d" % (names[i], r[i + 1] - r[i]) for i in range(6)) +
                          "  | block %d cycles; sweep: %d polls checked" % (rows[k + 1][0] - r[0], st & 0xffffffff))
                mean = sum(rows[k + 1][0] - rows[k][0] for k in range(1, NB - 1)) / (NB - 2)
                sweep = sum(rows[k][5] - rows[k][4] for k in range(1, NB - 1)) / (NB - 2)
                print("      wave %d: mean block %.0f cycles, of which in the sweep %.0f" % (wv, mean, sweep))

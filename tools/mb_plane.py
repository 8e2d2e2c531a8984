"""tg_resblock_plane (csrc/resblock_plane.hip): the residual trunk of an inference frame as ONE persistent launch (activations
resident in LDS, neighbour hand-offs after every conv) against nb x tg_resblock (bit-identity: the same MFMA order) and against
nb x tg_resblock_c64_thr (the per-block launches it replaces: time per block in a graph).
    python tools/mb_plane.py --build   (here, cross-compiles the trace library)      python tools/mb_plane.py [--trace]   (GPU)"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tecogan_amd import build as B  # noqa: E402

so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_rp.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    B.build(verbose=False)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "resblock_plane_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_RP_TRACE", "-c", os.path.join(csrc, "resblock_plane.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "resblock_plane.hip"]
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
bf = torch.bfloat16
NB = 16
wrow = [(torch.randn(9, 64, 64, device=DEV) * 0.03).to(bf) for _ in range(2 * NB)]
wf = [K.frag_order(w) for w in wrow]
bias = [torch.randn(64, device=DEV) * 0.1 for _ in range(2 * NB)]


def ref(x, nb):
    a = x
    for i in range(nb):
        a = K.resblock(0, a, wf[2 * i], bias[2 * i], wf[2 * i + 1], bias[2 * i + 1], None, None, None, torch.empty_like(a), w_frag=True)
    return a


def plane(x, nb, scratch, out=None, variant=0):
    return K.resblock_plane(x, wf[0:2 * nb:2], bias[0:2 * nb:2], wf[1:2 * nb:2], bias[1:2 * nb:2],
                            torch.full_like(x, 7.0) if out is None else out, scratch, variant)


print("residual trunk, stateless forward: ONE persistent launch (tg_resblock_plane) against nb x tg_resblock (bit-identity)")
for (N, H, W) in ((1, 16, 32), (1, 48, 96), (1, 40, 70), (2, 33, 64), (1, 270, 480)):
    x = torch.randn(N, H, W, 64, device=DEV).to(bf)
    scratch = K.resblock_plane_scratch(N, H, W, DEV)
    for nb in (1, 2, 5, 16):
        r = ref(x, nb)
        res = []
        for variant in (0, 1):
            for rep in range(3):
                t0 = time.time()
                o = plane(x, nb, scratch, variant=variant)
                torch.cuda.synchronize()
                ok = torch.equal(o.view(torch.int16), r.view(torch.int16))
                d = float((o.float() - r.float()).abs().max())
                res.append("%s/%.0fms%s" % ("ok" if ok else "BAD", (time.time() - t0) * 1e3, "" if ok else "(max |d| %.3g)" % d))
        print("  [%d,%d,%d] %2d blocks: %s   give-ups %d, epoch %d" % (N, H, W, nb, " ".join(res), int(scratch[2]), int(scratch[0])))
    if (N, H, W) == (1, 48, 96):
        xa = x.clone()
        o = plane(xa, 16, scratch, out=xa)
        torch.cuda.synchronize()
        print("  in place (out = x): %s" % ("ok" if torch.equal(o.view(torch.int16), ref(x, 16).view(torch.int16)) else "BAD"))

N, H, W = 1, 270, 480
x = torch.randn(N, H, W, 64, device=DEV).to(bf)
scratch = K.resblock_plane_scratch(N, H, W, DEV)
print("time per block at [1,270,480,64] (19.1 GFLOP per block, MFMA floor 7.6 us), graph of 4 trunks x 16 blocks:")
bufs = [torch.empty_like(x), torch.empty_like(x)]


def thr():
    a = x
    for i in range(NB):
        a = K.resblock_c64_thr(a, wf[2 * i], bias[2 * i], wf[2 * i + 1], bias[2 * i + 1], bufs[i & 1])


for _ in range(3):
    thr()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(4):
        thr()
print("  %-44s %.2f us per block" % ("16 launches (tg_resblock_c64_thr)", timeit(g.replay, 30, 5) / (4 * NB)))
for variant, label in ((0, "weight prefetch distance 6 steps"), (1, "9 steps")):
    out = torch.empty_like(x)
    pa = K.PlaneArgs(x, wf[0:2 * NB:2], bias[0:2 * NB:2], wf[1:2 * NB:2], bias[1:2 * NB:2], out, scratch, variant)
    for _ in range(3):
        pa.launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            pa.launch()
    us = timeit(g.replay, 30, 5) / (4 * NB)
    torch.cuda.synchronize()
    print("  %-44s %.2f us per block   (give-ups %d)" % ("one launch (tg_resblock_plane), " + label, us, int(scratch[2])))
    if TRACE:
        lib = C.CDLL(so)
        lib.tg_debug_rp_trace.argtypes = [C.POINTER(C.c_ulonglong)]
        buf = (C.c_ulonglong * (8 * 16 * 16))()
        assert lib.tg_debug_rp_trace(buf) == 0
        t = list(buf)
        names = ["conv_1 MFMAs", "values", "publish", "LDS writes", "hand-off + barrier", "conv_2 MFMAs", "values", "publish", "LDS writes",
                 "hand-off + barrier"]
        for wv in (0, 3, 6):
            rows = [t[(wv * 16 + k) * 16:(wv * 16 + k) * 16 + 11] for k in range(NB)]
            for k in (1, 7, 14):
                r = rows[k]
                print("      wave %d block %2d: " % (wv, k) + "  ".join("%s %d" % (names[i], r[i + 1] - r[i]) for i in range(10)) +
                      "  | block %d cycles" % (rows[k + 1][0] - r[0]))
            print("      wave %d: mean block %.0f cycles" % (wv, sum(rows[k + 1][0] - rows[k][0] for k in range(1, NB - 1)) / (NB - 2)))

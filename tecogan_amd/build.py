"""Build libtecogan_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtecogan_hip.so")
SOURCES = ["conv_igemm.hip", "conv3x3.hip", "conv3x3_ws.hip", "conv3x3_dma.hip", "conv3x3_wr.hip", "conv4x4s2.hip", "resblock_lat.hip", "resblock_chain.hip", "resblock_plane.hip", "resblock_thr.hip", "hr_bwd_lat.hip", "hr_fwd_lat.hip", "conv_wgrad.hip", "conv_wgrad_tr.hip", "warp.hip", "pointwise.hip", "schedule.hip", "losses.hip", "runtime.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "handoff.h"), os.path.join(HERE, "..", "include", "tecogan_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

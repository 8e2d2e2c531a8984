#!/usr/bin/env python
"""Inference throughput of the recurrent step (BASELINE config 5: 480x270 -> 1920x1080)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd.infer import InferenceEngine

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=270); ap.add_argument("--w", type=int, default=480)
ap.add_argument("--frames", type=int, default=60); ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--nres", type=int, default=16)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--no-lookahead", action="store_true")
ap.add_argument("--window", type=int, default=0, help="lookahead window: FNet on the next K frame pairs as one batch (0: one-frame lookahead)")
ap.add_argument("--no-plane-in", action="store_true", help="the input-stage conv as its own launch instead of inside tg_resblock_plane")
ap.add_argument("--no-plane", action="store_true", help="the residual trunk as 16 tg_resblock_c64_thr launches instead of tg_resblock_plane")
a = ap.parse_args()
tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
eng = InferenceEngine(a.nres, a.h, a.w, "cuda", tdt, use_graph=not a.no_graph)
eng.G.resblock_plane = not a.no_plane
eng.G.plane_input_conv = not a.no_plane_in
frames = torch.rand(8, 1, a.h, a.w, 3, device="cuda")
nx = (lambda i: None) if (a.no_lookahead or a.window) else (lambda i: frames[(i + 1) % 8])
eng.window = max(a.window, 1)
up = (lambda i: [frames[(i + j) % 8] for j in range(1, a.window + 1)]) if a.window else (lambda i: None)
for i in range(a.warmup):
    eng.step(frames[i % 8], next_frame=nx(i), upcoming=up(i))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.frames):
    eng.step(frames[(a.warmup + i) % 8], next_frame=nx(a.warmup + i), upcoming=up(a.warmup + i))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
gflop = 2 * (184.2 + 16.2) * (a.h * a.w) / (270 * 480) * (1 if a.nres == 16 else 0.7)
print(json.dumps({"metric": "inference HR fps", "value": round(a.frames / dt, 2), "ms_per_frame": round(dt / a.frames * 1e3, 3),
                  "shape": "%dx%d->%dx%d" % (a.w, a.h, 4 * a.w, 4 * a.h), "dtype": a.dtype,
                  "approx_TFLOPs": round(gflop * a.frames / dt / 1e3, 2), "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))

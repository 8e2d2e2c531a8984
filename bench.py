#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: 4x SR training frames/s of the FRVSR/TecoGAN step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one full training step (forward, backward, [RCCL grad all-reduce], three TF-Adams) over one
synthetic batch; frames/s follows the reference's own accounting `batch_size * steps/s * frame_len`
(reference main.py:369,407-411).  Default workload = BASELINE.json configs[1] (FRVSR training, runGan.py 4,
B=4 x 10 frames of 32x32 LR per GPU, bf16 activations / fp32 master weights); `--config tecogan` runs
configs[2] (runGan.py 3).  Weak scaling: per-GPU batch is fixed, `value` is the whole-job aggregate.
Inputs are resident in HBM before the timed region.  rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "f32": 157.3}          # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--config", choices=["frvsr", "tecogan"], default="frvsr")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def make_flags(config):
    from tecogan_amd.flags import frvsr_flags, tecogan_flags
    return frvsr_flags() if config == "frvsr" else tecogan_flags()


def synthetic_batch(F, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 3, generator=g)
    y = torch.rand(F.batch_size, F.RNN_N, 4 * F.crop_size, 4 * F.crop_size, 3, generator=g) * 2 - 1
    return x.to(device), y.to(device)


def dominant_kernel_roofline(dtype, device):
    """Time the generator's 3x3 64->64 convolution (the res-block workhorse: 20 of the 24 convs of every generator_F call,
    and with mirrored taps their input gradients) at the training shape [4,32,32,64].  The launches are captured in a
    hipGraph (as the product step is) and the graph is replayed between two HIP events recorded on the launch stream, so the
    figure is GPU time per launch, not Python/ctypes dispatch time.  `traffic` = HBM-side bytes per launch from the
    committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by tools/pmc_summary.py), null if absent."""
    from tecogan_amd import kernels as K
    from tecogan_amd._lib import ACT_RELU
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    N, H, W, Cc = 4, 32, 32, 64
    x = torch.randn(N, H, W, Cc, device=device).to(tdt)
    w = (torch.randn(9, Cc, Cc, device=device) * 0.05).to(tdt)
    b = torch.zeros(Cc, device=device)
    out = torch.empty_like(x)
    d = K.conv_desc(N, H, W, Cc, H, W, Cc, 3, 3, 1, 1, 1, 0, K.dt(x), K.dt(out), ACT_RELU)
    per_graph, replays = 200, 10
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):
            K.conv_forward(d, x, w, b, None, None, out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            K.conv_forward(d, x, w, b, None, None, out)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (per_graph * replays)
    flops = 2.0 * N * H * W * Cc * 9 * Cc          # algorithmic: 2 * M * N * K = 302 MFLOP per launch
    ach = flops / (us * 1e-6) / 1e12
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            t = json.load(fh).get("conv3x3_tile_kernel@[4,32,32,64->64]_%s" % dtype)
        if t:
            traffic = t["fetch_bytes"] + t["write_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "mfma", "kernel": "conv3x3_tile_kernel 3x3 64->64 @[4,32,32,64] %s (generator res-block conv)" % dtype,
            "achieved": round(ach, 3), "peak": PEAK[dtype], "unit": "TFLOP/s", "frac": round(ach / PEAK[dtype], 5),
            "us_per_launch": round(us, 3), "flop_per_launch": flops,
            "algorithmic_bytes": N * H * W * Cc * 2 * (2 if dtype == "bf16" else 4) + 9 * Cc * Cc * (2 if dtype == "bf16" else 4),
            "traffic": traffic}


def cpu_baseline(config, seconds):
    """The CPU oracle (torch restatement of the reference TF1 path) timed on this box's host cores."""
    from oracle import teco as OT
    F = OT.frvsr_flags() if config == "frvsr" else OT.default_flags()
    gan = config != "frvsr"
    S = OT.State(F, seed=42, gan=gan)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 3, generator=g)
    y = torch.rand(F.batch_size, F.RNN_N, 4 * F.crop_size, 4 * F.crop_size, 3, generator=g) * 2 - 1
    frame_len = 2 * F.RNN_N - 1 if F.pingpang else F.RNN_N
    OT.train_step(S, x, y)                             # warm-up
    n, t0 = 0, time.time()
    while n < 5 and (time.time() - t0) < seconds:
        OT.train_step(S, x, y)
        n += 1
    dt = time.time() - t0
    return {"value": round(F.batch_size * frame_len * n / dt, 3), "unit": "frames/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full %s training steps (B=%d, %d frames) of the torch-CPU oracle after 1 warm-up" %
                      (n, config, F.batch_size, frame_len)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    if world > 1 and local >= ndev and os.environ.get("TG_DIST_BACKEND", "nccl") != "nccl":
        local = local % ndev                                     # plumbing test: several ranks share one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TG_DIST_BACKEND", "nccl")     # "nccl" == RCCL; gloo only for 1-GPU plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD
    from tecogan_amd.engine import TrainEngine
    F = make_flags(a.config)
    gan = a.config != "frvsr"
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    eng = TrainEngine(F, device, gan=gan, act_dtype=tdt, seed=42, process_group=pg, use_graph=not a.no_graph)
    x, y = synthetic_batch(F, 1234 + rank, device)
    eng.set_batch(x, y)
    for _ in range(a.warmup):
        eng.step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = tmax.item()
    frame_len = eng.T
    value = world * F.batch_size * frame_len * a.steps / dt
    if rank == 0:
        L = eng.losses()
        assert all(v == v for v in L.values()), "NaN in losses: %s" % L
        line = {"metric": "4x SR train frames/sec (G+D step)", "value": round(value, 2), "unit": "frames/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                "data": "synthetic (uniform LR/HR sequences, seeded xavier weights%s)" %
                        (", He-normal VGG-19 stand-in" if eng.use_vgg else ""),
                "config": {"workload": ("configs[1]: FRVSR training (runGan.py 4): " if a.config == "frvsr" else
                                        "configs[2]: TecoGAN training (runGan.py 3: G + Dst + VGG + ping-pong): ") +
                                       "B=%d x %d frames, %dx%d LR -> %dx%d HR per GPU, num_resblock=%d" %
                                       (F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 4 * F.crop_size, 4 * F.crop_size,
                                        F.num_resblock),
                           "global_batch": world * F.batch_size, "frames_per_step": world * F.batch_size * frame_len,
                           "parallelism": "dp%d" % world, "hipgraph": not a.no_graph},
                "losses": {k: round(v, 6) for k, v in L.items() if v != 0.0}}
        line["roofline"] = dominant_kernel_roofline(a.dtype, device)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.config, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

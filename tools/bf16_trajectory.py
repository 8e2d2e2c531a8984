#!/usr/bin/env python
"""Does the timed (bf16) mode TRAIN like the parity (fp32) mode?  VERDICT r4 item 6: the four-engine control of bench.py's
`loss_trajectory` (reference training loop lib/Teco.py:316-417,441-449: content / warp / discriminator losses under three Adam
optimisers and the D gate) over >= 2000 steps on >= 8 DISTINCT batches of panning clips (so that FNet sees real motion) and 3 seeds.

Per seed, from the same damped seeded weights and the same cyclic batch sequence:
    bf16      the timed mode                         f32       the parity mode (the reference trajectory)
    bf16_b    the timed mode once more (atomics)     f32_pert  the parity mode from weights rounded ONCE to bf16 (the control:
                                                               a single perturbation of the size bf16 applies everywhere)
Reported per loss: the relative deviation of the TAIL MEAN (last quarter of the run) from the fp32 run for bf16 and for the
control, and the largest relative deviation of the sampled curve.  The step is a chaotic map (a gated GAN): trajectories separate
whatever the perturbation; a drop-in mode is one whose deviation is of the order of the control's.
    python tools/bf16_trajectory.py [--steps 2000] [--batches 8] [--seeds 3] [--out profiles/r05_bf16_trajectory.txt]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def pan_batch(F, seed, device):
    """B clips of RNN_N frames: a pan (its own direction and speed per clip, up to 1.5 LR pixels per frame) over a smooth random
    field; HR targets in [-1, 1], LR inputs = antialiased bicubic / 4 in [0, 1] (the value ranges of lib/dataloader.py:96-154)."""
    g = torch.Generator().manual_seed(seed)
    B, T, hr = F.batch_size, F.RNN_N, 4 * F.crop_size
    xs, ys = [], []
    for b in range(B):
        vy, vx = [int(v) for v in torch.randint(-6, 7, (2,), generator=g)]
        span = 6 * T + 8
        field = torch.nn.functional.interpolate(torch.rand(1, 3, (hr + 2 * span) // 8, (hr + 2 * span) // 8, generator=g),
                                                size=(hr + 2 * span, hr + 2 * span), mode="bicubic", align_corners=False).clamp(0, 1)
        frames = torch.cat([field[:, :, span + vy * t:span + vy * t + hr, span + vx * t:span + vx * t + hr] for t in range(T)], 0)
        lr = torch.nn.functional.interpolate(frames, scale_factor=0.25, mode="bicubic", align_corners=False, antialias=True).clamp(0, 1)
        xs.append(lr.permute(0, 2, 3, 1))
        ys.append(frames.permute(0, 2, 3, 1) * 2 - 1)
    return torch.stack(xs).contiguous().to(device), torch.stack(ys).contiguous().to(device)


def run(mode, seed, batches, steps, every, device, perturb=False):
    from tecogan_amd.engine import TrainEngine
    from tecogan_amd.params import damp_values
    F = bench.make_flags("tecogan")
    e = TrainEngine(F, device, gan=True, act_dtype=torch.bfloat16 if mode == "bf16" else torch.float32, seed=42 + seed, use_graph=True)
    P = damp_values(e.ps.state_dict())
    if perturb:
        P = {k: v.bfloat16().float() for k, v in P.items()}
    e.ps.load(P)
    rows = []
    for it in range(steps):
        x, y = batches[it % len(batches)]
        yn = batches[(it + 1) % len(batches)][1]
        e.step(x, y, next_targets=yn if getattr(e, "lookahead", False) else None)
        if (it + 1) % every == 0:
            torch.cuda.synchronize()
            L = e.losses()
            rows.append([L.get("l2_content_loss", L.get("content_loss")), L.get("l2_warp_loss", L.get("warp_loss")), L.get("t_discrim_loss")])
    del e
    torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_bf16_trajectory.txt"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = ["content_loss", "warp_loss", "t_discrim_loss"]
    lines = ["# tools/bf16_trajectory.py: %d Adam steps of the TecoGAN configs[2] step over %d distinct panning-clip batches (cyclic), %d seeds;"
             % (a.steps, a.batches, a.seeds),
             "# losses sampled every %d steps; tail = the last quarter of the samples.  dev = |mean_tail(run) - mean_tail(f32)| / |mean_tail(f32)|;"
             % a.every,
             "# curve = max over samples of |run - f32| / |f32|.  control = fp32 from weights rounded once to bf16.",
             "%-5s %-15s %12s %12s %12s | %10s %10s %10s | %10s %10s" % ("seed", "loss", "tail f32", "tail bf16", "tail control", "dev bf16",
                                                                         "dev ctrl", "dev bf16_b", "curve bf16", "curve ctrl")]
    ratios = {n: [] for n in names}
    t0 = time.time()
    F = bench.make_flags("tecogan")
    for s in range(a.seeds):
        batches = [pan_batch(F, 1000 * s + k, dev) for k in range(a.batches)]
        tr = {"bf16": run("bf16", s, batches, a.steps, a.every, dev), "f32": run("f32", s, batches, a.steps, a.every, dev),
              "bf16_b": run("bf16", s, batches, a.steps, a.every, dev), "f32_pert": run("f32", s, batches, a.steps, a.every, dev, perturb=True)}
        for j, n in enumerate(names):
            col = {k: [r[j] for r in v] for k, v in tr.items()}
            q = max(1, len(col["f32"]) // 4)
            tail = {k: sum(v[-q:]) / q for k, v in col.items()}
            dv = {k: abs(tail[k] - tail["f32"]) / max(abs(tail["f32"]), 1e-12) for k in ("bf16", "f32_pert", "bf16_b")}
            cv = {k: max(abs(u - v) / max(abs(v), 1e-12) for u, v in zip(col[k], col["f32"])) for k in ("bf16", "f32_pert")}
            ratios[n].append((dv["bf16"], dv["f32_pert"]))
            lines.append("%-5d %-15s %12.5f %12.5f %12.5f | %10.2e %10.2e %10.2e | %10.2e %10.2e"
                         % (s, n, tail["f32"], tail["bf16"], tail["f32_pert"], dv["bf16"], dv["f32_pert"], dv["bf16_b"], cv["bf16"], cv["f32_pert"]))
        print("\n".join(lines[-3:]), flush=True)
    lines.append("# over the seeds: tail-mean deviation of bf16 / of the control (mean, max), and the ratio of the means")
    verdict_ok = True
    for n in names:
        b = [r[0] for r in ratios[n]]
        c = [r[1] for r in ratios[n]]
        ratio = (sum(b) / len(b)) / max(sum(c) / len(c), 1e-12)
        verdict_ok &= ratio <= 1.5 or max(b) <= 5e-3
        lines.append("%-15s bf16 mean %.2e max %.2e | control mean %.2e max %.2e | ratio of means %.2f" % (n, sum(b) / len(b), max(b), sum(c) / len(c), max(c), ratio))
    lines.append("# verdict: %s (criterion of VERDICT r4 item 6: bf16's tail-mean deviation within 1.5x the control's, or below 5e-3 outright)"
                 % ("bf16 deviates as a perturbation of its size does" if verdict_ok else "bf16 deviates MORE than the control: see DESIGN.md section 2"))
    lines.append("# wall time %.0f s" % (time.time() - t0))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-6:]))


if __name__ == "__main__":
    main()

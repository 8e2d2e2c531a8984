#!/usr/bin/env python
"""Host-side phases of main.py's training loop (engine step, running-average kernel, next batch) with the loader thread alive:
where does a loop iteration spend its host time?  python tools/mb_mainloop.py [--scenes DIR]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lib.dataloader import frvsr_gpu_data_loader
from tecogan_amd import kernels as K
from tecogan_amd.engine import TrainEngine
from tecogan_amd.flags import tecogan_flags

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--switch", type=float, default=1e-4)
ap.add_argument("--prefetch", type=int, default=2)
ap.add_argument("--threads", type=int, default=0)
a = ap.parse_args()
sys.setswitchinterval(a.switch)
kw = dict(input_video_dir=a.scenes, str_dir=1000, end_dir=1003, end_dir_val=1003, max_frm=13) if a.scenes else {}
F = tecogan_flags(**kw)
dev = torch.device("cuda:0")
if a.threads:
    torch.set_num_threads(a.threads)
rdata = frvsr_gpu_data_loader(F, device=dev, synthetic=not a.scenes)
if hasattr(rdata.loader, '_prefetch'):
    rdata.loader._prefetch = a.prefetch
eng = TrainEngine(F, dev, gan=True, act_dtype=torch.bfloat16)
avg = torch.zeros_like(eng.loss)
x, y = rdata.s_inputs, rdata.s_targets
for _ in range(5):
    eng.step(x, y); x, y = rdata.loader.next_batch()
torch.cuda.synchronize()
import tecogan_amd.engine as E
wait_t, launch_t = [0.0], [0.0]
def timed_replay(self=eng):
    main = torch.cuda.current_stream(); evs = {}
    for what, arg in E.plan_launch_order(self._segs, self.lazy_side):
        q0 = time.perf_counter()
        if what == "wait":
            for d in arg: evs[d].synchronize()
            wait_t[0] += time.perf_counter() - q0
        else:
            st = main if arg["skey"] == "M" else self.streams[arg["skey"]]
            for d in arg["deps"]: st.wait_event(evs[d])
            with torch.cuda.stream(st):
                arg["graph"].replay()
            arg["event"].record(st); evs[arg["name"]] = arg["event"]
            launch_t[0] += time.perf_counter() - q0
eng._replay = timed_replay
ph = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(a.steps):
    a0 = time.perf_counter(); eng.step(x, y)
    a1 = time.perf_counter(); K.lincomb(avg, eng.loss, avg, 0.99, 0.01)
    a2 = time.perf_counter(); x, y = rdata.loader.next_batch()
    a3 = time.perf_counter()
    ph[0] += a1 - a0; ph[1] += a2 - a1; ph[2] += a3 - a2
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %.2f ms per iteration (%.1f image/s x19); host phases ms: step %.2f  lincomb %.2f  next_batch %.2f" % (
    "png scenes" if a.scenes else "synthetic", dt / a.steps * 1e3, a.steps * 4 / dt, ph[0] / a.steps * 1e3,
    ph[1] / a.steps * 1e3, ph[2] / a.steps * 1e3))
print("inside step(): host waits on events %.2f ms, graph launches %.2f ms" % (wait_t[0] / a.steps * 1e3, launch_t[0] / a.steps * 1e3))
ld = rdata.loader
if hasattr(ld, "cache_hits"):
    print("decoded-frame cache: %d hits, %d misses" % (ld.cache_hits, ld.cache_misses))

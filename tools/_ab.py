import os, sys, time
os.environ["TG_SEG_STAMPS"] = "1"
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
eng = bench.new_engine("tecogan", "bf16", dev)
F = bench.make_flags("tecogan")
eng.set_batch(*bench.synthetic_batch(F, 1, dev))
kw = {"next_targets": True}
for i in range(30):
    eng.step(**kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(150):
    eng.step(**kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 150 * 1e3
t = eng.seg_stamps.cpu().tolist()
names = sorted(eng.seg_stamp_names.items(), key=lambda kv: t[2 * kv[1]])
print("%-22s %.3f ms/step | " % (os.environ.get("AB_TAG", ""), ms) + " ".join("%s %.2f" % (n, (t[2 * i + 1] - t[2 * i]) / 1e5) for n, i in names), flush=True)

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q -k "conv4x4s2 or bn_lrelu or discriminator or tecogan_step_matches or temporal or standin or world2 or latency_kernels or deterministic" --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 ) > $O/r05n_pytest.log 2>&1; cat $O/r05n_pytest.log
python - <<'PY' 2>&1 | grep -v "amdgpu.ids" | tee $O/r05n_engines.txt
import os, sys, time
os.environ["TG_SEG_STAMPS"] = "1"
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
for k in range(4):
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
    print("engine %d of the process: %.3f ms/step | " % (k + 1, ms) + " ".join("%s %.2f" % (n, (t[2 * i + 1] - t[2 * i]) / 1e5) for n, i in names), flush=True)
    del eng
    torch.cuda.empty_cache()
PY
( timeout 100 python tools/mb_k4.py --n 24 2>&1 | grep -v "^ROCm\|^HIP\|amdgpu.ids" ) | tee $O/r05n_mb_k4.txt

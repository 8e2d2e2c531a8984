cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -k "pack_d or conv4x4s2 or discriminator or tecogan_step_matches or dt_merge or temporal" --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 ) > $O/r05i_pytest.log 2>&1; cat $O/r05i_pytest.log
( for c in 7 8 9 7 8; do echo "TG_VGG_CUTS=$c"; TG_VGG_CUTS=$c timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | sed -n 3,18p; done ) > $O/r05i_cuts.txt 2>&1
python - <<'PY'
import re
txt=open("gpurun_out/r05i_cuts.txt").read().split("TG_VGG_CUTS=")[1:]
for blk in txt:
    lines=blk.strip().splitlines(); c=lines[0]
    seg={l.split()[0]:(float(l.split()[2]),float(l.split()[3]),float(l.split()[4])) for l in lines[1:] if len(l.split())==5 and l.split()[0]!="vggt"}
    t0=seg["head"][0]; t1=seg["update"][1]
    print("cut %s: step %.3f ms | "%(c,t1-t0)+" ".join("%s %.2f"%(k,v[2]) for k,v in seg.items()))
PY

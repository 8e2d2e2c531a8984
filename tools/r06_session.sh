#!/bin/bash
# Round-6 GPU session driver: `bash tools/r06_session.sh <tag> <what...>`; results under gpurun_out/r06<tag>_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
TAG=$1; shift
for what in "$@"; do
case $what in
  quicktests)
    ( time timeout 900 python -m pytest tests -m gpu -q -x --maxfail=5 -k "$QT" --durations=8 ) > $O/r06${TAG}_pytest_quick.log 2>&1
    grep -E "passed|failed" $O/r06${TAG}_pytest_quick.log | tail -3; grep -E "^FAILED|^ERROR|^\[" $O/r06${TAG}_pytest_quick.log | cut -c1-300 ;;
  suite)
    ( time timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=8 ) > $O/r06${TAG}_pytest_gpu.log 2>&1
    grep -E "passed|failed" $O/r06${TAG}_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r06${TAG}_pytest_gpu.log | cut -c1-300 ;;
  benchq)
    ( timeout 300 python bench.py --no-sub --no-cpu-baseline --steps 100 --warmup 10 ) > $O/r06${TAG}_benchq.json 2> $O/r06${TAG}_benchq.err
    python - <<PY
import json
d=json.loads(open("$O/r06${TAG}_benchq.json").read().strip().splitlines()[-1])
print("TecoGAN ms/step", d["ms_per_step"], "roofline", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
for f in d.get("roofline",{}).get("families",[]): print("   ", f)
PY
    tail -2 $O/r06${TAG}_benchq.err ;;
  frvsrq)
    ( timeout 300 python bench.py --config frvsr --no-sub --no-roofline --no-cpu-baseline --steps 200 --warmup 10 ) > $O/r06${TAG}_frvsrq.json 2> $O/r06${TAG}_frvsrq.err
    python -c "import json;d=json.loads(open('$O/r06${TAG}_frvsrq.json').read().strip().splitlines()[-1]);print('FRVSR ms/step', d['ms_per_step'])" ;;
  inferq)
    timeout 200 python tools/bench_infer.py 2>&1 | grep -v "^ROCm\|^HIP\|amdgpu.ids" | tail -3 | tee $O/r06${TAG}_inferq.txt ;;
  timeline)
    timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -24 > $O/r06${TAG}_seg_timeline.txt; cat $O/r06${TAG}_seg_timeline.txt ;;
  bench)
    ( time timeout 900 python bench.py ) > $O/r06${TAG}_bench.json 2> $O/r06${TAG}_bench.err; cut -c1-300 $O/r06${TAG}_bench.json; tail -c 2500 $O/r06${TAG}_bench.json; tail -3 $O/r06${TAG}_bench.err ;;
  *)
    echo "== $what"; ( eval "$what" ) 2>&1 | grep -v "^ROCm\|^HIP\|amdgpu.ids" | tail -60 | tee -a $O/r06${TAG}_misc.txt ;;
esac
done

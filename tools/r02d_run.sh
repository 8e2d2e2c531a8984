#!/bin/bash
# Round-2 GPU session D: fork tax of multi-stream hipGraphs, ws-kernel cycle trace, fixed tests, inference kernels A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 120 python tools/mb_forktax.py 2>&1 | tail -1 | tee $O/r02d_forktax.txt
MB_CHAIN=400 MB_BIG=12 timeout 120 python tools/mb_forktax.py 2>&1 | tail -1 | tee -a $O/r02d_forktax.txt
timeout 120 python tools/trace_ws.py 2>&1 | grep -v amdgpu.ids | tee $O/r02d_trace_ws.txt
for v in "" "TG_NO_WARP_VEC=1 TG_NO_BICUBIC_QUAD=1" "TG_NO_C3WS=1"; do echo "== infer $v" | tee -a $O/r02d_infer.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02d_infer.txt; done
( time TG_OVERLAP=0 timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=5 ) > $O/r02d_pytest_gpu.log 2>&1; tail -30 $O/r02d_pytest_gpu.log | cut -c1-300
for parts in 7 15; do
  echo "== tecogan TG_OVERLAP_PARTS=$parts" | tee -a $O/r02d_ab.txt; TG_OVERLAP_PARTS=$parts timeout 120 python bench.py --steps 40 --warmup 3 --no-sub --no-roofline --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02d_ab.txt
done

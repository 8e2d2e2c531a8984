#!/bin/bash
# Round-2 GPU session C: capture-pattern reproducer, GPU suite on the serial schedule, overlap pieces one by one,
# ws-kernel variants.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python tools/mb_capture.py 2>&1 | tee $O/r02c_capture.txt
( time TG_OVERLAP=0 timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=8 ) > $O/r02c_pytest_gpu.log 2>&1; tail -40 $O/r02c_pytest_gpu.log | cut -c1-300
B="python bench.py --steps 40 --warmup 3 --no-sub --no-roofline --no-cpu-baseline"
for parts in 0 1 2 4 8 16 3 7 15; do
  echo "== tecogan TG_OVERLAP_PARTS=$parts" | tee -a $O/r02c_ab.txt; TG_OVERLAP_PARTS=$parts timeout 120 $B 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02c_ab.txt
done
for v in "" "TG_C3WS_PERCU=1" "TG_NO_C3WS=1"; do echo "== microbench $v" | tee -a $O/r02c_microbench.txt; env $v timeout 100 python tools/microbench.py --only "conv3x3" 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $O/r02c_microbench.txt; done

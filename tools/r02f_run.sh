#!/bin/bash
# Round-2 GPU session F: coarser segment DAG, cost of segmentation alone, swapped-operand chain epilogue, ws prologue.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 120 python tools/trace_ws.py 2>&1 | grep -v amdgpu.ids | head -20 | tee $O/r02f_trace_ws.txt
timeout 100 python tools/microbench.py --only "conv3x3" 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/r02f_microbench.txt
( time timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=5 ) > $O/r02f_pytest_gpu.log 2>&1; tail -12 $O/r02f_pytest_gpu.log | cut -c1-300
TG_OVERLAP=0 timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "C3" 2>&1 | tail -5 | cut -c1-300 | tee $O/r02f_c3_serial.txt
B="python bench.py --steps 40 --warmup 3 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_OVERLAP_PARTS=0" "TG_OVERLAP_PARTS=0 TG_SEGMENTS=force" "TG_OVERLAP_PARTS=15" "TG_OVERLAP_PARTS=15 TG_C3_PRIO=1" "TG_OVERLAP_PARTS=7" "TG_OVERLAP_PARTS=5" "TG_OVERLAP_PARTS=10"; do
  echo "== tecogan $v" | tee -a $O/r02f_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('graph_segments'), d['config'].get('host_enqueue_ms_per_step'))" | tee -a $O/r02f_ab.txt
done
echo "== frvsr" | tee -a $O/r02f_ab.txt; timeout 120 $B --config frvsr 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02f_ab.txt
timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02f_ab.txt

#!/bin/bash
# Round-2 GPU session B: full GPU suite (ws kernel, overlap engine), ws-kernel microbench A/B, overlap / priority A/B on the
# headline step, full bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=8 ) > $O/r02b_pytest_gpu.log 2>&1; tail -40 $O/r02b_pytest_gpu.log
for v in "" "TG_NO_C3WS=1"; do echo "== microbench $v" | tee -a $O/r02b_microbench.txt; env $v timeout 200 python tools/microbench.py 2>&1 | grep -v "^$" | tee -a $O/r02b_microbench.txt; done
B="python bench.py --steps 60 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_OVERLAP=0" "TG_OVERLAP=1" "TG_OVERLAP=1 TG_C3_PRIO=1" "TG_OVERLAP=0 TG_NO_C3WS=1"; do
  echo "== tecogan $v" | tee -a $O/r02b_ab.txt; env $v timeout 200 $B 2>&1 | tail -1 | cut -c1-160 | tee -a $O/r02b_ab.txt
done
for v in "TG_OVERLAP=0" "TG_OVERLAP=1" "TG_OVERLAP=1 TG_C3_PRIO=1"; do
  echo "== frvsr $v" | tee -a $O/r02b_ab.txt; env $v timeout 200 $B --config frvsr 2>&1 | tail -1 | cut -c1-160 | tee -a $O/r02b_ab.txt
done
( time timeout 600 python bench.py ) > $O/r02b_bench.json 2> $O/r02b_bench.err; cut -c1-300 $O/r02b_bench.json; tail -3 $O/r02b_bench.err

#!/bin/bash
# Round-2 GPU session H: packed narrow-image tiles, chain launches with co-residency-friendly tiles, bicubic / residual fixes.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in "" "TG_NO_C3_PACK=1"; do echo "== microbench $v" | tee -a $O/r02h_microbench.txt; env $v timeout 100 python tools/microbench.py --only "conv3x3" 2>&1 | grep "vgg5\|fnet \[36,4\|fnet \[72,8\|inf  \[1,270" | tee -a $O/r02h_microbench.txt; done
( time timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=4 ) > $O/r02h_pytest_gpu.log 2>&1; tail -10 $O/r02h_pytest_gpu.log | cut -c1-300; grep "^\[C" $O/r02h_pytest_gpu.log
B="python bench.py --steps 60 --warmup 3 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_OVERLAP_PARTS=0" "TG_OVERLAP_PARTS=15 TG_CHAIN_COEXIST=0" "TG_OVERLAP_PARTS=15" "TG_OVERLAP_PARTS=5" "TG_OVERLAP_PARTS=13" "TG_OVERLAP_PARTS=15 TG_NO_C3_PACK=1"; do
  echo "== tecogan $v" | tee -a $O/r02h_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('graph_segments'))" | tee -a $O/r02h_ab.txt
done
echo "== frvsr" | tee -a $O/r02h_ab.txt; timeout 120 $B --config frvsr 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02h_ab.txt
timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02h_ab.txt

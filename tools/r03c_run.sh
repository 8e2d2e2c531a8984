#!/bin/bash
# Round-3 session C: the new backward schedule (late VGG pieces descending on S, FNet backward split, late wgrads): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x > $O/r03c_pytest_full.txt 2>&1; grep -E "passed|failed|Error|assert|^\[C" $O/r03c_pytest_full.txt | head -20 | tee $O/r03c_pytest.txt
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in 47 111 175 239 495; do
  echo "== tecogan TG_OVERLAP_PARTS=$v" | tee -a $O/r03c_ab.txt; TG_OVERLAP_PARTS=$v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03c_ab.txt
done
for c in 1 4; do
  echo "== tecogan parts 239 TG_VGG_LATE_CHUNK=$c" | tee -a $O/r03c_ab.txt; TG_VGG_LATE_CHUNK=$c timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03c_ab.txt
done
for v in 47 175 431; do
  echo "== frvsr TG_OVERLAP_PARTS=$v" | tee -a $O/r03c_ab.txt; TG_OVERLAP_PARTS=$v timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03c_ab.txt
done
timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -32 | tee $O/r03c_seg_timeline.txt

#!/bin/bash
# Round-2 GPU session R: early VGG pass in several pieces (TG_VGG_CUTS) A/B; default parts now 47.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "" "6,12,16" "5,10,14" "7,13" "6,11,15,17" "8,14"; do
  echo "== tecogan TG_VGG_CUTS=$v" | tee -a $O/r02r_ab.txt; TG_VGG_CUTS=$v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02r_ab.txt
done
TG_VGG_CUTS=6,12,16 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl" | head -20 | tee $O/r02r_seg_timeline.txt
TG_VGG_CUTS=6,12,16 timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "tecogan_step_fp32_parity or three_steps" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | cut -c1-250 | tee $O/r02r_pytest.txt

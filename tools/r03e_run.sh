#!/bin/bash
# Round-3 session E: late VGG as ONE full-tile piece on S beside D's backward passes on M, BPTT alone afterwards
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_OVERLAP_PARTS=47" "TG_OVERLAP_PARTS=111" "TG_OVERLAP_PARTS=111 TG_DOWN_ON_MAIN=0" "TG_OVERLAP_PARTS=103" "TG_OVERLAP_PARTS=239"; do
  echo "== tecogan $v" | tee -a $O/r03e_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03e_ab.txt
done
TG_OVERLAP_PARTS=111 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -20 | tee $O/r03e_seg_timeline.txt
TG_OVERLAP_PARTS=111 timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "three_steps or tecogan_step_fp32_parity or standin or Dt_merge" 2>&1 | grep -E "passed|failed|Error|assert" | tee $O/r03e_pytest.txt

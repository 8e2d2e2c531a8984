#!/bin/bash
# Round-2 GPU session Q: overlap pieces under just-in-time side launches: bit 16 (VGG middle frames behind `down`), bit 32
# (generator weight gradients beside FNet's backward pass); parity of the replayed steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in 15 31 47 63 15; do
  echo "== tecogan TG_OVERLAP_PARTS=$v" | tee -a $O/r02q_ab.txt; TG_OVERLAP_PARTS=$v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02q_ab.txt
done
for v in 0 32; do
  echo "== frvsr TG_OVERLAP_PARTS=$v" | tee -a $O/r02q_ab.txt; TG_OVERLAP_PARTS=$v timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02q_ab.txt
done
TG_OVERLAP_PARTS=63 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl" | head -16 | tee $O/r02q_seg_timeline.txt
TG_OVERLAP_PARTS=63 timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "tecogan_step_fp32_parity or three_steps or no_pingpong or frvsr_two_steps" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | cut -c1-250 | tee $O/r02q_pytest.txt

#!/bin/bash
# Round-2 GPU session Y: default VGG cut at 11 of 19 frames: parity of the TecoGAN step (toy + C3) and the step time.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 200 --warmup 10 --no-sub --no-roofline --no-cpu-baseline"
echo "== tecogan (default)" | tee -a $O/r02y_ab.txt; timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02y_ab.txt
timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -16 | tee $O/r02y_seg_timeline.txt
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "tecogan_step_fp32_parity or three_steps or Dt_mergeDs or world2" 2>&1 | grep -E "passed|failed|Error|assert|\[C3\]" | tail -6 | cut -c1-300 | tee $O/r02y_pytest.txt

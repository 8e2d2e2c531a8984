#!/bin/bash
# Round-2 GPU session P: just-in-time side-stream launches (TG_LAZY_SIDE) A/B; parity of the replayed step.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_LAZY_SIDE=1" "TG_LAZY_SIDE=0" "TG_LAZY_SIDE=1 TG_OVERLAP_PARTS=31" "TG_LAZY_SIDE=1" "TG_LAZY_SIDE=0"; do
  echo "== tecogan $v" | tee -a $O/r02p_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02p_ab.txt
done
timeout 200 python tools/seg_timeline.py --steps 40 2>&1 | tail -34 | tee $O/r02p_seg_timeline.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "tecogan_step_fp32_parity or three_steps or no_pingpong or world2 or captured" 2>&1 | tail -4 | cut -c1-250 | tee $O/r02p_pytest.txt

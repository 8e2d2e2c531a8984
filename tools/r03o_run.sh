#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k "test_tecogan_three_steps_graph_and_gate or test_captured_exchange_segments_carry_nodes_standin_world2" 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee $O/r03o_pytest.txt
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 150 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_SPLIT_HEAD=0" "TG_SPLIT_HEAD=1" "TG_SPLIT_HEAD=0" "TG_SPLIT_HEAD=1"; do
  echo "== tecogan $v" | tee -a $O/r03o_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03o_ab.txt
done

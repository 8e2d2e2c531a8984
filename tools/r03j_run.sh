#!/bin/bash
# Round-3 session J: generalized transpose-read wgrad (+ output-conv variant)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "transpose_read or wgrad" 2>&1 | grep -E "passed|failed|Error|assert" | head | tee $O/r03j_pytest.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k "test_frvsr_step_fp32_parity or test_tecogan_step_fp32_parity or test_frvsr_step_bf16_error_is_bounded" --deselect tests/test_train_gpu.py::test_frvsr_step_fp32_parity_at_baseline_config_C2 --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 2>&1 | grep -E "passed|failed|Error|assert" | head | tee -a $O/r03j_pytest.txt
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_OVERLAP_PARTS=111 TG_WGRAD_TR=0" "TG_OVERLAP_PARTS=111"; do
  echo "== tecogan $v" | tee -a $O/r03j_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03j_ab.txt
  echo "== frvsr $v" | tee -a $O/r03j_ab.txt; env $v timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03j_ab.txt
done
timeout 100 python tools/mb_wgrad.py 2>&1 | grep "^wgrad" | tee $O/r03j_mb_wgrad.txt
